#!/usr/bin/env python3
"""Does a library GEMM of the scoring forward slow down under SUSTAINED load?  (Round 5: TunableOp's 60 ms timing loop
measures down_proj 17152 x 4096 x 11008 at 891 us = 0.69 of the 2.5 PFLOP/s peak; inside an attack step the same launch
takes 1065-1100 us = 0.58.)  Runs the product back to back for ~3 s under the shipped selection table and prints the
mean time of every window of 200 launches, then the same with 5 ms of idle between windows."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
from bimodalattack_amd import gemm_tuning  # noqa: E402

dev = torch.device("cuda", 0)
gemm_tuning.enable("auto", dev)
for name, M, N, K in (("down_proj", 17152, 4096, 11008), ("gate_up", 17152, 22016, 4096)):
    xs = [torch.randn((1, M, K), device=dev).to(torch.bfloat16) for _ in range(3)]
    ws = [(torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16) for _ in range(4)]
    for _ in range(5):
        torch.nn.functional.linear(xs[0], ws[0])
    torch.cuda.synchronize()
    time.sleep(1.0)
    flops = 2.0 * M * N * K
    for idle_ms in (0, 5):
        out = []
        for win in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(200):
                torch.nn.functional.linear(xs[i % 3], ws[i % 4])
            e1.record()
            torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 200
            out.append(us)
            if idle_ms:
                time.sleep(idle_ms / 1e3)
        print(f"{name} {M}x{N}x{K}, windows of 200 launches, {idle_ms} ms idle between windows: "
              + " ".join(f"{u:.0f}" for u in out) + f" us  (first {flops / out[0] / 1e6 / 2.5e3:.3f}, last {flops / out[-1] / 1e6 / 2.5e3:.3f} of 2.5 PF)", flush=True)
        time.sleep(2.0)
