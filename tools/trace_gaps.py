#!/usr/bin/env python3
"""Where the GPU waits for the host: idle gaps in a rocprofv3 --kernel-trace CSV, between the two marker kernels
bench.py launches around its timed region.

    python tools/trace_gaps.py <kernel_trace.csv> <out.txt> [min_gap_us=20] [top=40]

Prints the span, the busy time (union of kernel intervals), the idle time, a histogram of gaps, and the longest gaps
with the kernels on either side -- a gap after a `.item()`-style sync or a host-side planning step shows up here, a
launch-bound chain shows up as many 2-10 us gaps.
"""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
min_gap = float(sys.argv[3]) if len(sys.argv) > 3 else 20.0
top = int(sys.argv[4]) if len(sys.argv) > 4 else 40
with open(src) as f:
    rows = list(csv.DictReader(f))
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "ReduceOp<double" in r["Kernel_Name"])
lo, hi = (marks[0], marks[-1]) if len(marks) >= 2 else (0, 1 << 62)
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]) for r in rows
            if lo < int(r["Start_Timestamp"]) < hi and "ReduceOp<double" not in r["Kernel_Name"])
busy, gaps, end, prev = 0.0, [], ev[0][0], "(start)"
for s, e, name in ev:
    if s > end:
        gaps.append(((s - end) / 1e3, prev, name, (end - ev[0][0]) / 1e6))
        busy += (e - s) / 1e3
    else:
        busy += max(0, e - end) / 1e3
    if e > end:
        end, prev = e, name
span = (end - ev[0][0]) / 1e3
idle = sum(g[0] for g in gaps)
with open(dst, "w") as f:
    f.write(f"# {len(ev)} dispatches over {span / 1e3:.2f} ms: busy {busy / 1e3:.2f} ms, idle {idle / 1e3:.2f} ms ({100 * idle / span:.1f} %)\n")
    for a, b in ((0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 200), (200, 1000), (1000, 1e9)):
        sel = [g[0] for g in gaps if a <= g[0] < b]
        f.write(f"#   gaps {a:>5}-{b:<6g} us: {len(sel):6d}, {sum(sel) / 1e3:8.2f} ms\n")
    f.write("#   gap_us   at_ms  after -> before\n")
    for g in sorted(gaps, key=lambda g: -g[0])[:top]:
        if g[0] >= min_gap:
            f.write(f"{g[0]:10.1f} {g[3]:7.2f}  {g[1]}  ->  {g[2]}\n")
print(open(dst).read().split("#   gap_us")[0])
