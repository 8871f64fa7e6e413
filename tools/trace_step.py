#!/usr/bin/env python3
"""One step of a rocprofv3 --kernel-trace CSV as a timeline of gaps:

    python tools/trace_step.py <kernel_trace.csv> <out.txt> [anchor=sample_scatter_kernel] [which=6] [min_gap_us=15]

Cuts the trace between the `which`-th and the next launch of the anchor kernel (one per attack step) and lists every idle
gap of at least min_gap_us with the two kernels on either side, their queue ids and the offset into the step -- who the GPU
was waiting for when it waited (trace_gaps.py sums such gaps over the whole run)."""
import csv
import sys

src, dst = sys.argv[1], sys.argv[2]
anchor = sys.argv[3] if len(sys.argv) > 3 else "sample_scatter_kernel"
which = int(sys.argv[4]) if len(sys.argv) > 4 else 6
min_gap = float(sys.argv[5]) if len(sys.argv) > 5 else 15.0
with open(src) as f:
    rows = list(csv.DictReader(f))
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")) for r in rows)
marks = [i for i, e in enumerate(ev) if anchor in e[2]]
a, b = marks[which], marks[which + 1]
step = ev[a:b + 1]
t0 = step[0][0]
short = lambda n: n.replace("void ", "").replace("(anonymous namespace)::", "").replace("at::native::", "")[:64]      # noqa: E731
with open(dst, "w") as f:
    span = (step[-1][0] - t0) / 1e3
    busy = sum(e - s for s, e, _, _ in step[:-1]) / 1e3
    f.write(f"# step between launches {which} and {which + 1} of {anchor}: {len(step) - 1} dispatches, {span / 1e3:.3f} ms, "
            f"kernel time {busy / 1e3:.3f} ms\n#   at_ms   gap_us  queue  before -> after\n")
    end = step[0][1]
    for i in range(1, len(step)):
        s, e, name, q = step[i]
        gap = (s - end) / 1e3
        if gap >= min_gap:
            p = step[i - 1]
            pp = step[i - 2] if i >= 2 else p
            nn = step[i + 1] if i + 1 < len(step) else step[i]
            f.write(f"  {(s - t0) / 1e6:7.3f} {gap:8.1f}  {p[3]:>3}->{q:<3} [{short(pp[2])}] {short(p[2])}  ->  {short(name)} [{short(nn[2])}]\n")
        end = max(end, e)
print(open(dst).read())
