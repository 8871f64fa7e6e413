#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv) of
tools/kernel_bench.py into per-kernel HBM bytes per launch.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json

Corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.
"""
import csv
import json
import re
import sys
from collections import defaultdict


def fold(path, counter):
    """{(kernel<template>, total threads, run#): [counter values]} -- a "run" is a maximal
    stretch of consecutive dispatches of one kernel at one grid (kernel_bench.py launches each
    case warmup+iters times back to back), so two cases that share a symbol and a grid size
    are kept apart."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            m = re.search(r"\b(ce_rows_kernel|ce_dlogits_kernel|ce_fold_kernel|splice_kernel|mask_topk_kernel|"
                          r"linf_step_vec4|linf_step_scalar|sample_scatter_kernel|rand_positions_kernel|rmsnorm_kernel|"
                          r"swiglu_kernel|qknorm_rope2_kernel|rope2_kernel|rope_kernel|attn_merge_kernel|gather_rows_kernel|rmsnorm_short_kernel|ragged_attn_kernel|ragged_attn_long_kernel|"
                          r"topk_slice_kernel|topk_merge_kernel|prefix_attn_kernel|add_rmsnorm_kernel|splice_rows_kernel|gemm_nt_kernel|gemm_mid_kernel|gemm_mid_reduce_kernel|causal_fwd_kernel|causal_dq_kernel|causal_dkv_kernel|(?:Custom_)?Cijk_\w+)\b(<[^>]*>)?",
                          r["Kernel_Name"])
            if not m:
                continue
            rows.append((int(r["Dispatch_Id"]), m.group(1) + (m.group(2) or ""), r.get("Grid_Size", ""), float(r["Counter_Value"])))
    rows.sort()
    by = defaultdict(list)
    run_of = defaultdict(int)
    last = {}
    for disp, kern, grid, val in rows:
        k = (kern, grid)
        # a gap of more than 2 dispatch ids (the fold kernel sits between CE launches) starts a new run
        if k in last and disp - last[k] > 2:
            run_of[k] += 1
        last[k] = disp
        by[(kern, grid, run_of[k])].append(val)
    return by


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = fold(fetch, "FETCH_SIZE"), fold(write, "WRITE_SIZE")
    res = {}
    for key in sorted(set(f) | set(w)):
        kern, grid, run = key
        fv, wv = f.get(key, []), w.get(key, [])
        fetch_b = 2.0 * 1024.0 * (sum(fv) / len(fv)) if fv else None
        write_b = 1024.0 * (sum(wv) / len(wv)) if wv else None
        res[f"{kern}/threads{grid}/run{run}"] = dict(launches=max(len(fv), len(wv)), fetch_bytes_per_launch_corrected=fetch_b,
                                         write_bytes_per_launch=write_b,
                                         hbm_bytes_per_launch=(fetch_b or 0) + (write_b or 0))
    json.dump(dict(corrections="FETCH_SIZE KiB x2 (gfx950 half-count), WRITE_SIZE KiB x1", kernels=res), open(out, "w"), indent=1)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
