#!/usr/bin/env python3
"""Fold rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, csv) of
tools/kernel_bench.py into per-kernel HBM bytes per launch.

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> out.json

Corrections (MI355X_MICROARCH.md, HBM): counters are in KiB; on gfx950 FETCH_SIZE
reports exactly half of the bytes of a wide coalesced streaming read, so it is
doubled; WRITE_SIZE is exact for 16-byte-per-lane streaming stores.  Both factors are
CHECKED per pass on kernel_bench.py's calibration case (a 256 MiB copy: known bytes) and
the measured factors are reported beside the nominal ones.

Rows are cut into CASES at kernel_bench.py's markers (a float64 reduction in front of every
case): keys are "case<i>/<kernel><template>/threads<n>".  (Round 4 cut "runs" at gaps in the
dispatch ids; three bma_gemm_nt cases with one symbol and one grid ran back to back without
a gap, were averaged together, and two entries then read fewer bytes than their weights hold.)
"""
import csv
import json
import re
import sys
from collections import defaultdict


KERNELS = (r"\b(ce_rows_kernel|ce_dlogits_kernel|ce_fold_kernel|splice_kernel|mask_topk_kernel|"
           r"linf_step_vec4|linf_step_scalar|sample_scatter_kernel|rand_positions_kernel|rmsnorm_kernel|"
           r"swiglu_kernel|qknorm_rope2_kernel|rope2_kernel|rope_kernel|attn_merge_kernel|gather_rows_kernel|rmsnorm_short_kernel|ragged_attn_kernel|ragged_attn_long_kernel|"
           r"topk_slice_kernel|topk_merge_kernel|prefix_attn32_kernel|prefix_attn_kernel|add_rmsnorm_kernel|splice_rows_kernel|gemm_nt_kernel|gemm_mid_kernel|gemm_mid_reduce_kernel|causal_fwd_kernel|causal_dq_kernel|causal_dkv_kernel|(?:Custom_)?Cijk_\w+)\b(<[^>]*>)?")


def is_marker(name: str) -> bool:
    """kernel_bench.py's case marker: aten's reduction of a float64 tensor (no case launches one)."""
    return "reduce_kernel" in name and "double" in name


def fold(path, counter):
    """{(case index, kernel<template>, total threads): [counter values]}; the calibration copy of case 0 under the
    kernel name "calib_copy"."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r.get("Grid_Size", ""), float(r["Counter_Value"])))
    rows.sort()
    by = defaultdict(list)
    case = -1
    for disp, name, grid, val in rows:
        if is_marker(name):
            case += 1
            continue
        if case < 0:
            continue
        m = re.search(KERNELS, name)
        if m:
            by[(case, m.group(1) + (m.group(2) or ""), grid)].append(val)
        elif case == 0 and "copy" in name.lower() and val > 100 * 1024:        # the 256 MiB calibration copy (counter in KiB)
            by[(0, "calib_copy", grid)].append(val)
    return by


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = fold(fetch, "FETCH_SIZE"), fold(write, "WRITE_SIZE")
    # calibration: the 256 MiB copy read 256 MiB and wrote 256 MiB
    known = 256.0 * 1024.0                                                       # KiB
    cf = [v for k, vs in f.items() if k[1] == "calib_copy" for v in vs]
    cw = [v for k, vs in w.items() if k[1] == "calib_copy" for v in vs]
    fetch_factor = known / (sum(cf) / len(cf)) if cf else None
    write_factor = known / (sum(cw) / len(cw)) if cw else None
    res = {}
    for key in sorted(set(f) | set(w)):
        case, kern, grid = key
        if kern == "calib_copy":
            continue
        fv, wv = f.get(key, []), w.get(key, [])
        fetch_b = 2.0 * 1024.0 * (sum(fv) / len(fv)) if fv else None
        write_b = 1024.0 * (sum(wv) / len(wv)) if wv else None
        res[f"case{case}/{kern}/threads{grid}"] = dict(launches=max(len(fv), len(wv)), fetch_bytes_per_launch_corrected=fetch_b,
                                                       write_bytes_per_launch=write_b,
                                                       hbm_bytes_per_launch=(fetch_b or 0) + (write_b or 0))
    json.dump(dict(corrections="FETCH_SIZE KiB x2 (gfx950 half-count), WRITE_SIZE KiB x1",
                   calibration=dict(case="256 MiB device-to-device copy (kernel_bench.py calib/copy_256MiB)",
                                    fetch_factor_measured=fetch_factor, write_factor_measured=write_factor,
                                    fetch_factor_applied=2.0, write_factor_applied=1.0),
                   kernels=res), open(out, "w"), indent=1)
    print("calibration (256 MiB copy): FETCH_SIZE factor", fetch_factor, "WRITE_SIZE factor", write_factor)
    for k, v in res.items():
        print(k, v)


if __name__ == "__main__":
    main()
