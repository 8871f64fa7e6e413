#!/usr/bin/env python3
"""CLIP-L's 577-row tower products: the library (with and without its bias) against bma_gemm_mid
(no bias) at the same shapes, each over 24 different weights from one hipGraph -- would routing the tower through
bma_gemm_mid (given a bias epilogue) pay?

    python3 tools/tower_gemm_probe.py
"""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402


def timed(fn, n_w, rounds=20):
    dev = torch.device("cuda", 0)
    s = torch.cuda.Stream(dev)
    s.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(s):
        for i in range(n_w):
            fn(i)
    torch.cuda.current_stream(dev).wait_stream(s)
    torch.cuda.synchronize(dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=s):
        for i in range(n_w):
            fn(i)
    g.replay()
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds):
        g.replay()
    e1.record()
    torch.cuda.synchronize(dev)
    return 1e3 * e0.elapsed_time(e1) / (rounds * n_w)


def main():
    from bimodalattack_amd import ops
    from bimodalattack_amd import gemm_tuning
    dev = torch.device("cuda", 0)
    gemm_tuning.enable("table", dev)
    ops.gemm_workspace_for_graphs(dev)
    M, L = 577, 24
    g = torch.Generator(device=dev).manual_seed(3)
    shapes = [("qkv", 3072, 1024), ("qkv dX", 1024, 3072), ("out", 1024, 1024), ("fc1", 4096, 1024), ("fc1 dX", 1024, 4096),
              ("fc2", 1024, 4096), ("fc2 dX", 4096, 1024)]
    for name, N, K in shapes:
        ws = [(torch.randn((N, K), generator=g, device=dev) * 0.02).to(torch.bfloat16) for _ in range(L)]
        bs = [(torch.randn((N,), generator=g, device=dev) * 0.02).to(torch.bfloat16) for _ in range(L)]
        x = (torch.randn((1, M, K), generator=g, device=dev)).to(torch.bfloat16)
        lib_b = timed(lambda i: torch.nn.functional.linear(x, ws[i], bs[i]), L)
        lib_n = timed(lambda i: torch.nn.functional.linear(x, ws[i]), L)
        try:
            mid = timed(lambda i: ops.gemm_mid(x, ws[i]), L)
            err = float((ops.gemm_mid(x, ws[0]).float() - torch.nn.functional.linear(x, ws[0]).float()).abs().max())
        except Exception as e:
            mid, err = float("nan"), str(e)[:60]
        fl = 2.0 * M * N * K
        print(f"{name:8s} M={M} N={N:5d} K={K:5d}: library+bias {lib_b:6.1f} us   library {lib_n:6.1f} us ({fl / lib_n / 1e6:5.0f} TF/s)   "
              f"bma_gemm_mid {mid:6.1f} us ({fl / mid / 1e6:5.0f} TF/s)   x{lib_n / mid:4.2f}   max|diff| {err}", flush=True)


if __name__ == "__main__":
    main()
