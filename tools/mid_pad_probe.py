#!/usr/bin/env python3
"""Does bma_gemm_mid run faster when the rows of its operands are NOT a power of two apart?

    python tools/mid_pad_probe.py [--rows 644] [--layers 16]

A unit of K (64 columns) of a 224 x 256 tile is 480 rows x 128 B; with contiguous operands at K = 4096 the rows sit 8 KiB
apart, so everything a workgroup asks for in one unit has the same address bits 7..12 -- if L2 / fabric channels are picked
by those bits, a unit is served by one channel.  Here the same product is timed through the C ABI with the weight and / or
the activation stored with a padded leading dimension (K + pad elements).  `layers` different weights back to back from
one hipGraph, as tools/gemm_bench.py --mid.
"""
import argparse
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from bimodalattack_amd import ops  # noqa: E402
from bimodalattack_amd.native import check, lib  # noqa: E402

DEV = torch.device("cuda", 0)
SHAPES = [("gate_up", 22016, 4096), ("gate_up dX", 4096, 22016), ("qkv dX", 4096, 12288), ("down", 4096, 11008)]


def graph_time(fn, n):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=644)
    ap.add_argument("--layers", type=int, default=16)
    ap.add_argument("--pads", default="0,64,128,192,320")
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    ops.gemm_workspace(DEV)
    pair = ops.gemm_workspace_for_graphs(DEV)
    M = args.rows
    gen = torch.Generator(device=DEV).manual_seed(0)
    for name, N, K in [s for s in SHAPES if args.only is None or s[0] in args.only.split(",")]:
        flops = 2.0 * M * N * K
        ref = None
        for which in ("w", "x", "both"):
            for pad in [int(p) for p in args.pads.split(",")]:
                if pad == 0 and which != "w":
                    continue
                pw = pad if which in ("w", "both") else 0
                px = pad if which in ("x", "both") else 0
                g2 = torch.Generator(device=DEV).manual_seed(1)
                wbufs = [torch.empty((N, K + pw), device=DEV, dtype=torch.bfloat16) for _ in range(args.layers)]
                for wb in wbufs:
                    wb[:, :K] = (torch.randn((N, K), generator=g2, device=DEV) * 0.02).to(torch.bfloat16)
                xb = torch.empty((M, K + px), device=DEV, dtype=torch.bfloat16)
                xb[:, :K] = torch.randn((M, K), generator=g2, device=DEV).to(torch.bfloat16)
                ys = [torch.empty((M, N), device=DEV, dtype=torch.bfloat16) for _ in range(args.layers)]
                st = torch.cuda.current_stream

                def fn():
                    for wb, y in zip(wbufs, ys):
                        check("bma_gemm_mid", lib.bma_gemm_mid(xb.data_ptr(), K + px, wb.data_ptr(), K + pw, y.data_ptr(), N, M, N, K, 1,
                                                               pair[0].data_ptr(), pair[0].numel(), st().cuda_stream))
                t = statistics.median(graph_time(fn, args.layers) for _ in range(5))
                if ref is None:
                    ref = ys[0].clone()
                same = bool(torch.equal(ref, ys[0]))
                print(f"{name:11s} M={M} N={N:5d} K={K:5d}  pad {which:4s} +{pad:3d}: {t:7.1f} us  {flops / t / 1e6:6.0f} TF/s = "
                      f"{flops / t / 1e6 / 2500:4.2f} of peak   same bits as unpadded: {same}", flush=True)
                del wbufs, xb, ys
    del gen


if __name__ == "__main__":
    main()
