#!/usr/bin/env python3
"""Fold a rocprofv3 --kernel-trace CSV by (kernel, grid, workgroup): the --stats summary averages all
launches of a symbol, and one symbol serves very different sizes here (the candidate forward and the
batch-1 gradient pass call the same kernels), so per-shape averages are what compares with bench.py's
live per-launch figures.

    python tools/trace_by_grid.py <kernel_trace.csv> <out.txt> [top]
"""
import collections
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 70
agg = collections.OrderedDict()
with open(src) as f:
    for r in csv.DictReader(f):
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
    f.write(f"# whole process, {sum(v[0] for _, v in rows)} dispatches, {total / 1e3:.1f} ms of kernel time; grid sizes in threads\n")
    f.write("#   total_ms  share  launches    avg_us    min_us    max_us  grid x workgroup  kernel\n")
    for (name, gx, gy, gz, wg), (n, us, lo, hi) in rows[:top]:
        f.write(f"{us / 1e3:10.2f} {100 * us / total:5.1f}% {n:9d} {us / n:9.1f} {lo:9.1f} {hi:9.1f}  ({gx},{gy},{gz})x{wg}  {name}\n")
