#!/usr/bin/env python3
"""Random ragged row lists through bma_ragged_attention, saved for comparison between the two kernels behind it:

    BMA_RAGGED_LONG=0 python tools/fuzz_long_attn.py --out gpurun_out/fuzz_short.pt
    BMA_RAGGED_LONG=1 python tools/fuzz_long_attn.py --out gpurun_out/fuzz_long.pt
    python tools/fuzz_long_attn.py --compare gpurun_out/fuzz_short.pt gpurun_out/fuzz_long.pt

(The kernel choice is read once per process, hence two runs.)  Blocks of 96 to 500 tokens with random lengths, random
parent lengths (rows of the list in front of a candidate's own), prefixes of 0 to 70 keys, 128- and 256-wide heads in
groups of 1, 2, 3 and 4 query heads per key/value head, with and without a merged prefix partial, bf16 and fp16.
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def cases(n, seed):
    g = torch.Generator().manual_seed(seed)
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    for _ in range(n):
        Dh = (128, 256)[ri(0, 1)]
        Hk = ri(1, 3)
        rep = ri(1, 4)
        B2 = ri(1, 24)
        max_len = ri(96, 500 if Dh == 128 else 330)
        P = (0, ri(1, 70))[ri(0, 1)]
        merge = P == 0 and ri(0, 2) == 0
        dtype = (torch.bfloat16, torch.float16)[ri(0, 3) == 0]
        parent = ri(0, 40)                                    # rows 0 .. parent-1 of the list are the parent's
        lens = [ri(1, max_len) for _ in range(B2)]
        lens[ri(0, B2 - 1)] = max_len
        firsts = [ri(0, parent) for _ in range(B2)]
        yield dict(Dh=Dh, Hk=Hk, H=Hk * rep, B2=B2, max_len=max_len, P=P, merge=merge, dtype=dtype, parent=parent, lens=lens, firsts=firsts)


def run(c, seed, dev):
    from bimodalattack_amd import ops
    g = torch.Generator(device=dev).manual_seed(seed)
    starts, n = [], c["parent"]
    for ln in c["lens"]:
        starts.append(n)
        n += ln
    N, H, Hk, Dh, dt = n, c["H"], c["Hk"], c["Dh"], c["dtype"]
    q = torch.randn((N, H, Dh), generator=g, device=dev).to(dt)
    k, v = (torch.randn((N, Hk, Dh), generator=g, device=dev).to(dt) for _ in range(2))
    as4 = lambda t: t.unsqueeze(0).transpose(1, 2)
    cs, cf, cl = (torch.tensor(x, dtype=torch.int32, device=dev) for x in (starts, c["firsts"], c["lens"]))
    scale = Dh ** -0.5
    if c["merge"]:
        o1 = torch.randn((N, H, Dh), generator=g, device=dev).to(dt)
        lse1 = torch.randn((H, N), generator=g, device=dev) * 2
        return ops.ragged_attention(as4(q), as4(k), as4(v), None, None, cs, cf, cl, c["max_len"], scale, o1=o1, lse1=lse1)
    P = c["P"]
    pk, pv = (torch.randn((max(P, 1), Hk, Dh), generator=g, device=dev).to(dt)[:P] for _ in range(2))
    return ops.ragged_attention(as4(q), as4(k), as4(v), as4(pk) if P else None, as4(pv) if P else None, cs, cf, cl, c["max_len"], scale)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out")
    ap.add_argument("--compare", nargs=2)
    ap.add_argument("--n", type=int, default=60)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    if a.compare:
        x, y = (torch.load(p) for p in a.compare)
        worst = 0.0
        for i, (u, w) in enumerate(zip(x["out"], y["out"])):
            u, w = u[x["cases"][i]["parent"]:], w[x["cases"][i]["parent"]:]      # the parent's rows are keys only: nobody writes them
            d = float((u.float() - w.float()).abs().max())
            tol = 2.5e-2 if u.dtype == torch.bfloat16 else 4e-3
            worst = max(worst, d / tol)
            if not (d < tol) or not torch.isfinite(w.float()).all():
                print("MISMATCH case", i, x["cases"][i], "max abs diff", d)
                return 1
        print(f"{len(x['out'])} cases agree; worst difference {worst:.2f} of the tolerance; kernels: {x['mode']} vs {y['mode']}")
        return 0
    dev = "cuda:0"
    outs, cs = [], []
    for i, c in enumerate(cases(a.n, a.seed)):
        outs.append(run(c, 1000 + i, dev).cpu())
        cs.append({k: (str(v) if k == "dtype" else v) for k, v in c.items() if k not in ("lens", "firsts")})
    torch.cuda.synchronize()
    torch.save(dict(out=outs, cases=cs, mode=os.environ.get("BMA_RAGGED_LONG", "1")), a.out)
    print("saved", len(outs), "cases to", a.out)
    return 0


if __name__ == "__main__":
    sys.exit(main())
