#!/usr/bin/env python3
"""bma_gemm_nt / bma_gemm_mid against the library on the batch-1 shapes of the gradient pass, cold weights.

    python tools/gemm_bench.py [--rows 65,44] [--layers 32] [--rounds 5] [--json out.json]
    python tools/gemm_bench.py --mid [--rows 644,599] [--layers 16] [--sweep]      # the pass with the image in the prompt

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
    ap.add_argument("--chain", action="store_true", help="the pass's own chain of skinny products (per layer forward qkv -> gate/up -> down, "
                    "then backward through the transposed shapes in reverse) from one hipGraph, with and without the cross-product "
                    "weight prefetch of bma_gemm_nt_next; per-product figures of the same-shape chains as well")
    ap.add_argument("--mid", action="store_true", help="bma_gemm_mid at a few hundred rows (default --rows 644,599 --layers 16) "
                    "instead of bma_gemm_nt; with --sweep also pinned tile widths / K splits / split tail columns")
    args = ap.parse_args()
    if args.mid:
        return main_mid(args)
    if args.chain:
        return main_chain(args)
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


def main_chain(args):
    """A/B of the cross-product weight prefetch: (1) per product shape, `layers` different weights back to
    back, each launch given its successor; (2) the pass's own order -- per layer qkv, gate/up, down forward, then per layer in
    reverse down dX, gate/up dX, qkv dX -- as one hipGraph.  Kernel launches only (no norms / attention in between): an upper
    bound on what the hint can buy in the pass."""
    ops.GEMM_NT_MIN_K_OVER_N = 0.0
    ops.GEMM_NT_PREFETCH = False                    # successors are passed explicitly here
    gemm_tuning.enable("auto", DEV)
    ops.gemm_workspace(DEV)
    ops.gemm_workspace_for_graphs(DEV)
    gen = torch.Generator(device=DEV).manual_seed(0)
    rows = [int(r) for r in args.rows.split(",")]
    layers = args.layers
    names = ["qkv", "gate_up", "down", "down dX", "gate_up dX", "qkv dX"]
    shapes = {n: (N, K) for n, N, K in SHAPES}
    W = {n: [(torch.randn(shapes[n], generator=gen, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(layers)] for n in names}
    out = {}
    for M in rows:
        X = {K: torch.randn((1, M, K), generator=gen, device=DEV).to(torch.bfloat16) for K in {k for _, k in shapes.values()}}
        for n in names:
            ws = W[n]
            x = X[shapes[n][1]]
            plain = lambda: [ops.gemm_nt(x, w) for w in ws]                                              # noqa: E731
            hinted = lambda: [ops.gemm_nt(x, w, next_w=ws[i + 1] if i + 1 < len(ws) else None) for i, w in enumerate(ws)]   # noqa: E731
            tp, th = [], []
            for _ in range(args.rounds):
                tp.append(graph_time(plain, len(ws)))
                th.append(graph_time(hinted, len(ws)))
            a, b = statistics.median(tp), statistics.median(th)
            nbytes = 2.0 * (M * shapes[n][1] + shapes[n][0] * shapes[n][1] + M * shapes[n][0])
            out[f"{n} M={M}"] = dict(plain_us=a, prefetch_us=b, plain_TBps=nbytes / a / 1e6, prefetch_TBps=nbytes / b / 1e6)
            print(f"{n:11s} M={M:3d}: plain {a:6.1f} us ({nbytes / a / 1e6:4.2f} TB/s)   with next-weight prefetch {b:6.1f} us "
                  f"({nbytes / b / 1e6:4.2f} TB/s)   x{a / b:5.3f}", flush=True)
        seq = []
        for l in range(layers):
            seq += [("qkv", l), ("gate_up", l), ("down", l)]
        for l in reversed(range(layers)):
            seq += [("down dX", l), ("gate_up dX", l), ("qkv dX", l)]
        wl = [W[n][l] for n, l in seq]
        xs = [X[shapes[n][1]] for n, _ in seq]
        plain = lambda: [ops.gemm_nt(x, w) for x, w in zip(xs, wl)]                                       # noqa: E731
        hinted = lambda: [ops.gemm_nt(x, w, next_w=wl[i + 1] if i + 1 < len(wl) else None) for i, (x, w) in enumerate(zip(xs, wl))]   # noqa: E731
        tp, th = [], []
        for _ in range(args.rounds):
            tp.append(graph_time(plain, 1))
            th.append(graph_time(hinted, 1))
        a, b = statistics.median(tp), statistics.median(th)
        out[f"pass chain M={M}"] = dict(plain_us=a, prefetch_us=b, launches=len(seq))
        print(f"the pass's chain at {M} rows ({len(seq)} launches, {layers} layers): plain {a / 1e3:6.3f} ms   with prefetch {b / 1e3:6.3f} ms   "
              f"x{a / b:5.3f}  ({(a - b) / len(seq):+.2f} us per launch)", flush=True)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


def main_mid(args):
    import ctypes
    from bimodalattack_amd.native import lib
    rows = [int(r) for r in (args.rows if args.rows != "65,44" else "644,599").split(",")]
    layers = args.layers if args.layers != 32 else 16
    ops.GEMM_MID_MIN_K_OVER_N = 0.0                # time the kernel on every shape, routed or not
    gemm_tuning.enable("auto", DEV)
    ops.gemm_workspace(DEV)
    ops.gemm_workspace_for_graphs(DEV)
    out = {}
    gen = torch.Generator(device=DEV).manual_seed(0)
    plan = (ctypes.c_int * 8)()
    totals = {}
    for name, N, K in [s_ for s_ in SHAPES if args.only is None or s_[0] in args.only.split(",")]:
        ws = [(torch.randn((N, K), generator=gen, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(layers)]
        for M in rows:
            x = torch.randn((1, M, K), generator=gen, device=DEV).to(torch.bfloat16)
            lib_fn = lambda: [torch.nn.functional.linear(x, w) for w in ws]          # noqa: E731
            own_fn = lambda: [ops.gemm_mid(x, w) for w in ws]                         # noqa: E731
            tl, to = [], []
            for _ in range(args.rounds):
                tl.append(graph_time(lib_fn, len(ws)))
                to.append(graph_time(own_fn, len(ws)))
            l, o = statistics.median(tl), statistics.median(to)
            flops = 2.0 * M * N * K
            lib.bma_gemm_mid_plan(M, N, K, plan)
            pl = list(plan)
            # what every tile pulls L2 -> LDS: (224 + tile width) rows x K per tile, padding rows included
            l2_bytes = 2.0 * pl[1] * pl[3] * (224 + 64 * pl[2]) * K
            routed = K >= 2.5 * N or N >= 4.0 * K
            t_ = totals.setdefault(M, [0.0, 0.0])
            t_[0] += l
            t_[1] += o if routed else l
            out[f"{name} M={M} N={N} K={K}"] = dict(library_us=l, kernel_us=o, library_TFLOPs=flops / l / 1e6, kernel_TFLOPs=flops / o / 1e6,
                                                    kernel_frac_of_mfma_peak=flops / o / 1e6 / 2500.0, speedup=l / o, routed=routed,
                                                    plan=dict(tile_cols=64 * pl[2], col_tiles=pl[3], splits=pl[4], workgroups=pl[6], unsplit_tiles=pl[7]),
                                                    l2_to_lds_TBps=l2_bytes / o / 1e6)
            print(f"{name:11s} M={M:3d} N={N:5d} K={K:5d}: library {l:7.1f} us ({flops / l / 1e6:6.0f} TF/s)   bma_gemm_mid {o:7.1f} us "
                  f"({flops / o / 1e6:6.0f} TF/s = {flops / o / 1e6 / 2500:4.2f} of MFMA peak; tiles 224x{64 * pl[2]} x{pl[3] * pl[1]}, "
                  f"{pl[4]} splits, {pl[6]} workgroups; L2->LDS {l2_bytes / o / 1e6:4.1f} TB/s)   x{l / o:4.2f}{'' if routed else '   (not routed)'}", flush=True)
            if args.sweep:
                for nf, S_, tail in ((3, 1, 0), (4, 1, 0), (3, 3, 0), (4, 3, 0), (4, 4, 0), (4, 5, 0), (4, 6, 0), (4, 8, 1), (4, 4, 1), (4, 8, 2)):
                    lib.bma_gemm_mid_set_plan(nf, S_, tail, -1)
                    if lib.bma_gemm_mid_plan(M, N, K, plan) == 0 and lib.bma_gemm_mid_ws_bytes(M, N, K) <= ops._GEMM_WS_BYTES \
                            and not (tail and plan[7] > 256):
                        tt = statistics.median(graph_time(own_fn, len(ws)) for _ in range(3))
                        p2 = list(plan)
                        print(f"      pinned tiles 224x{64 * nf} splits={S_} split tail columns={tail} -> {p2[6]} workgroups, {p2[7]} unsplit tiles: "
                              f"{tt:7.1f} us ({flops / tt / 1e6:6.0f} TF/s)", flush=True)
                lib.bma_gemm_mid_set_plan(0, 0, -1, -1)
        del ws
    for M, (l, o) in totals.items():
        print(f"per layer at {M} rows (forward + input gradients, as routed): library {l:.1f} us -> {o:.1f} us", flush=True)
    if args.json:
        json.dump(out, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
