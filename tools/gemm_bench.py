#!/usr/bin/env python3
"""bma_gemm_nt against the library on the batch-1 shapes of the gradient pass, cold weights.

    python tools/gemm_bench.py [--rows 65,44] [--layers 32] [--rounds 5] [--json out.json]

Each shape is timed the way the pass meets it: `layers` different weight tensors of the shape (32 x 180 MB does not
fit the 256 MB Infinity Cache, so every launch streams its weight from HBM), launched back to back from ONE hipGraph
between two HIP events; library (torch.nn.functional.linear under the shipped TunableOp table) and kernel alternate
inside one process, `rounds` times, and the median is reported (cdna_hip_programming.md 5.4 rule 24).
"""
import argparse
import json
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import torch  # noqa: E402

from bimodalattack_amd import gemm_tuning, ops  # noqa: E402

DEV = torch.device("cuda", 0)
SHAPES = [("gate_up", 22016, 4096), ("gate_up dX", 4096, 22016), ("qkv", 12288, 4096), ("qkv dX", 4096, 12288),
          ("down", 4096, 11008), ("down dX", 11008, 4096), ("o_proj", 4096, 4096)]


def graph_time(fn, n):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        keep = fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    del keep
    return 1e3 * e0.elapsed_time(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", default="65,44")
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="comma list of shape names")
    ap.add_argument("--sweep", action="store_true", help="time the kernel under pinned decompositions (bma_gemm_nt_set_plan): "
                    "the planner's choice with each flag combination, round 3's 128-row slabs, and neighbours")
    args = ap.parse_args()
    ops.GEMM_NT_MIN_K_OVER_N = 0.0                 # time the kernel on every shape, routed or not
    gemm_tuning.enable("auto", DEV)
    ops.gemm_workspace(DEV)
    ops.gemm_workspace_for_graphs(DEV)
    out = {}
    gen = torch.Generator(device=DEV).manual_seed(0)
    for name, N, K in [s_ for s_ in SHAPES if args.only is None or s_[0] in args.only.split(",")]:
        ws = [(torch.randn((N, K), generator=gen, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(args.layers)]
        for M in [int(r) for r in args.rows.split(",")]:
            x = torch.randn((1, M, K), generator=gen, device=DEV).to(torch.bfloat16)
            lib_fn = lambda: [torch.nn.functional.linear(x, w) for w in ws]          # noqa: E731
            own_fn = lambda: [ops.gemm_nt(x, w) for w in ws]                          # noqa: E731
            tl, to = [], []
            for _ in range(args.rounds):
                tl.append(graph_time(lib_fn, len(ws)))
                to.append(graph_time(own_fn, len(ws)))
            nbytes = 2.0 * (M * K + N * K + M * N)
            l, o = statistics.median(tl), statistics.median(to)
            if args.sweep:
                import ctypes
                from bimodalattack_amd.native import lib
                plan = (ctypes.c_int * 8)()
                lib.bma_gemm_nt_plan(M, N, K, plan)
                _, _, ntw0, R0, _, S0, _, _ = list(plan)
                T = K // 64
                cands = [(0, 0, 0, f) for f in (0, 1, 2, 3)] + [(2, 128, 0, 3), (2, 128, 0, 0)]
                cands += [(ntw0, R0, s_, 3) for s_ in sorted({max(1, S0 // 2), min(16, S0 * 2)}) if s_ != S0 and s_ <= T]
                if ntw0 == 3:
                    cands += [(3, 192, 0, 3), (2, 0, 0, 3)]
                else:
                    cands += [(3, 0, 0, 3)]
                for ntw, R, S_, fl in cands:
                    lib.bma_gemm_nt_set_plan(ntw, R, S_, fl)
                    if lib.bma_gemm_nt_plan(M, N, K, plan) != 0 or lib.bma_gemm_nt_ws_bytes(M, N, K) > ops._GEMM_WS_BYTES \
                            or lib.bma_gemm_nt_tiles(M, N, K) > ops._GEMM_COUNTERS:
                        continue
                    tt = statistics.median(graph_time(own_fn, len(ws)) for _ in range(3))
                    pl = list(plan)
                    print(f"      pinned ntw={ntw} R={R} S={S_} flags={fl} -> ntw={pl[2]} R={pl[3]} slabs={pl[4]} S={pl[5]} xcd={pl[6]} nt={pl[7]}: "
                          f"{tt:7.1f} us ({nbytes / tt / 1e6:4.2f} TB/s)", flush=True)
                lib.bma_gemm_nt_set_plan(0, 0, 0, -1)
            out[f"{name} M={M} N={N} K={K}"] = dict(library_us=l, kernel_us=o, library_TBps=nbytes / l / 1e6, kernel_TBps=nbytes / o / 1e6,
                                                    kernel_frac_of_8TBps=nbytes / o / 1e6 / 8.0, speedup=l / o, kernel_min_us=min(to), library_min_us=min(tl))
            print(f"{name:11s} M={M:3d} N={N:5d} K={K:5d}: library {l:7.1f} us ({nbytes / l / 1e6:4.2f} TB/s)   bma_gemm_nt {o:7.1f} us "
                  f"({nbytes / o / 1e6:4.2f} TB/s = {nbytes / o / 1e6 / 8.0:4.2f} of 8)   x{l / o:4.2f}", flush=True)
        del ws
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
