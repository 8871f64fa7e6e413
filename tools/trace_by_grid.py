#!/usr/bin/env python3
"""Fold a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): the --stats summary averages all
launches of a symbol, and one symbol serves very different sizes here (the candidate forward and the
batch-1 gradient pass call the same kernels), so per-shape averages are what compares with bench.py's
live per-launch figures.

    python tools/trace_by_grid.py <kernel_trace.csv> <out.txt> [top] [--between-markers]

--between-markers: only the kernels between the first and the last marker kernel (a float64 reduction:
tools/grad_pass_profile.py brackets its measured region with two of them).
"""
import collections
import csv
import sys

argv = [a for a in sys.argv[1:] if not a.startswith("--")]
between = "--between-markers" in sys.argv
src, dst = argv[0], argv[1]
top = int(argv[2]) if len(argv) > 2 else 70
agg = collections.OrderedDict()
with open(src) as f:
    rows_all = list(csv.DictReader(f))
span = None
if between:
    marks = sorted(int(r["Start_Timestamp"]) for r in rows_all if "ReduceOp<double" in r["Kernel_Name"])
    if len(marks) >= 2:
        span = (marks[0], marks[-1])
if True:
    for r in rows_all:
        if span and not (span[0] < int(r["Start_Timestamp"]) < span[1]):
            continue
        if span and "ReduceOp<double" in r["Kernel_Name"]:
            continue
        key = (r["Kernel_Name"][:110], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"], r["Workgroup_Size_X"])
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        a = agg.setdefault(key, [0, 0.0, 1e30, 0.0])
        a[0] += 1
        a[1] += us
        a[2] = min(a[2], us)
        a[3] = max(a[3], us)
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
total = sum(v[1] for _, v in rows)
with open(dst, "w") as f:
    f.write(f"# {'between the marker kernels' if span else 'whole process'}, {sum(v[0] for _, v in rows)} dispatches, {total / 1e3:.1f} ms of kernel time; grid sizes in threads\n")
    f.write("#   total_ms  share  launches    avg_us    min_us    max_us  grid x workgroup  kernel\n")
    for (name, gx, gy, gz, wg), (n, us, lo, hi) in rows[:top]:
        f.write(f"{us / 1e3:10.2f} {100 * us / total:5.1f}% {n:9d} {us / n:9.1f} {lo:9.1f} {hi:9.1f}  ({gx},{gy},{gz})x{wg}  {name}\n")
