"""What the split-K tail of bma_gemm_nt costs: every split product of the gradient pass timed whole and with the kernel
stopped right behind its partial stores (bma_gemm_nt_set_plan flag 4: wrong results, measurement only).

    python tools/gemm_tail_probe.py
"""
import os, sys, statistics, torch, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bimodalattack_amd import ops, gemm_tuning
from bimodalattack_amd.native import lib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from gemm_bench import graph_time
DEV = torch.device("cuda", 0)
ops.GEMM_NT_MIN_K_OVER_N = 0.0
ops.gemm_workspace(DEV); ops.gemm_workspace_for_graphs(DEV)
gen = torch.Generator(device=DEV).manual_seed(0)
for name, N, K in [("gate_up dX", 4096, 22016), ("qkv dX", 4096, 12288), ("down", 4096, 11008), ("o", 4096, 4096)]:
    ws = [(torch.randn((N, K), generator=gen, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(32)]
    x = torch.randn((1, 65, K), generator=gen, device=DEV).to(torch.bfloat16)
    fn = lambda: [ops.gemm_nt(x, w) for w in ws]
    out = []
    for fl in (3, 7):
        lib.bma_gemm_nt_set_plan(0, 0, 0, fl)
        out.append(statistics.median(graph_time(fn, 32) for _ in range(3)))
    lib.bma_gemm_nt_set_plan(0, 0, 0, -1)
    print(f"{name:11s} full {out[0]:6.1f} us   without the reduction tail {out[1]:6.1f} us   tail {out[0]-out[1]:5.1f} us", flush=True)
    del ws
