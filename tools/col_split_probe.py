"""Column cut of a product whose LAST tile round is nearly empty: rows x N as one library call against (the first c tile columns
that fill whole rounds, the rest) into column slices of one output (fused.round_plan).  Medians of 5 x 24 launches.

    PYTHONPATH=. python tools/col_split_probe.py
"""
import statistics, sys, torch
from bimodalattack_amd import gemm_tuning
DEV = "cuda:0"
gemm_tuning.enable("auto", torch.device(DEV))
g = torch.Generator(device=DEV).manual_seed(0)
bf = torch.bfloat16
def bench(fn, iters=24):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3
for name, N, K, Ms in [("gate_up", 22016, 4096, (2112, 2176, 2240, 2304, 2432, 16896)), ("qkv", 12288, 4096, (2304, 4352))]:
    ws = [(torch.randn((N, K), generator=g, device=DEV) * 0.02).to(bf) for _ in range(4)]
    C = (N + 255) // 256
    for M in Ms:
        r = (M + 255) // 256
        k = r * C // 256
        c = k * 256 // r
        x = torch.randn((M, K), generator=g, device=DEV).to(bf)
        out = torch.empty((M, N), device=DEV, dtype=bf)
        cnt = [0]
        def whole():
            w = ws[cnt[0] & 3]; cnt[0] += 1
            torch.mm(x, w.t(), out=out)
        def split():
            w = ws[cnt[0] & 3]; cnt[0] += 1
            torch.mm(x, w[:c * 256].t(), out=out[:, :c * 256])
            torch.mm(x, w[c * 256:].t(), out=out[:, c * 256:])
        cnt[0] = 0; whole(); ref = out.clone(); cnt[0] = 0; split(); cnt[0] = 0
        same = bool((out.float() - ref.float()).abs().max() <= 2 ** -7 * ref.float().abs().max())
        res = {"whole": [], "cols": []}
        for _ in range(5):
            res["whole"].append(bench(whole)); res["cols"].append(bench(split))
        print(f"{name:8s} M={M} ({r} x {C} = {r * C} tiles = {r * C / 256:.2f} rounds; cut at {c} tile columns, {r * (C - c)} tiles behind it): "
              + "  ".join(f"{k_} {statistics.median(v):7.1f} us" for k_, v in res.items()) + f"  equal {same}", flush=True)
