"""Does a row split at a tile-round boundary pay for the products of the scoring forward (fused.round_cut)?  M rows as one
library call against (the last multiple of 4096 rows, the rest) and against two equal calls, into row slices of one output;
four weights in turn, medians of 5 x 24 launches (profiles/r6_round_split.txt).

    PYTHONPATH=. [SPLIT_M=4352,8704,16896] python tools/round_split_probe.py
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
for name, N, K in [("o_proj", 4096, 4096), ("down", 4096, 11008), ("qkv", 12288, 4096), ("gate_up", 22016, 4096)]:
    ws = [(torch.randn((N, K), generator=g, device=DEV) * 0.02).to(bf) for _ in range(4)]
    import os
    for M in [int(v) for v in os.environ.get('SPLIT_M', '16896,17152,17408').split(',')]:
        x = torch.randn((M, K), generator=g, device=DEV).to(bf)
        out = torch.empty((M, N), device=DEV, dtype=bf)
        cnt = [0]
        def whole():
            w = ws[cnt[0] & 3]; cnt[0] += 1
            torch.mm(x, w.t(), out=out)
        def split(cut):
            def f():
                w = ws[cnt[0] & 3]; cnt[0] += 1
                torch.mm(x[:cut], w.t(), out=out[:cut])
                torch.mm(x[cut:], w.t(), out=out[cut:])
            return f
        CUT = M // 4096 * 4096
        res = {"whole": [], "rounds+rest": [], "half": []}
        for r in range(5):
            res["whole"].append(bench(whole))
            res["rounds+rest"].append(bench(split(CUT)))
            res["half"].append(bench(split(M // 2)))
        line = f"{name:8s} M={M}: " + "  ".join(f"{k} {statistics.median(v):7.1f} us" for k, v in res.items())
        print(line, flush=True)
