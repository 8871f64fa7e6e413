#!/usr/bin/env python3
"""Read the per-wave clock stamps of a diagnostic build of the long-block attention kernel
(csrc/ragged_attention.hip under -DBMA_LONG_STAMPS, BMA_RAGGED_STAMPS=<file>): where a wave's cycles go.

    python tools/stamps_report.py gpurun_out/stamps.bin [waves per workgroup]

Per wave: [start, end, wait (vmcnt + barrier at a pair), QK, softmax, PV, realtime start, realtime end, epilogues,
items, stages multiplied, stages walked]; cycles of s_memtime, realtime in 100 MHz ticks.
"""
import sys

import numpy as np

NW = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12).astype(np.int64)
wave = np.tile(np.arange(NW), len(a) // NW)
keep = a[:, 0] > 0
a, wave = a[keep], wave[keep]
t0, t1, wait, qk, sm, pv, r0, r1, epi, items, mine, walk = a.T
ghz = (t1 - t0).sum() / ((r1 - r0).sum() * 10.0)
tot = t1 - t0
print(f"{len(a)} waves ({len(a) // NW} workgroups); in-kernel clock {ghz:.2f} GHz; kernel span {(r1.max() - r0.min()) / 100.0:.1f} us;"
      f" a wave lives {tot.mean() / ghz / 1e3:.1f} us on average, {tot.max() / ghz / 1e3:.1f} at most")
print(f"per wave: {items.mean():.1f} items, {walk.mean():.1f} stages walked, {mine.mean():.1f} multiplied; cycles {tot.mean():9.0f} ="
      f" wait {wait.mean():8.0f} + QK {qk.mean():8.0f} + softmax {sm.mean():8.0f} + PV {pv.mean():8.0f} + epilogues {epi.mean():8.0f}"
      f" + rest {(tot - wait - qk - sm - pv - epi).mean():8.0f}")
print(f"per multiplied stage: QK {qk.sum() / mine.sum():6.0f}  softmax {sm.sum() / mine.sum():6.0f}  PV {pv.sum() / mine.sum():6.0f};"
      f"  per walked stage: wait {wait.sum() / walk.sum():6.0f};  per item: epilogue {epi.sum() / items.sum():6.0f}")
for w in range(NW):
    m = wave == w
    print(f"  wave {w}: multiplied {mine[m].mean():6.1f} of {walk[m].mean():6.1f}  total {tot[m].mean():9.0f}  wait {wait[m].mean():8.0f}"
          f"  QK {qk[m].mean():8.0f} softmax {sm[m].mean():8.0f} PV {pv[m].mean():8.0f}  epilogues {epi[m].mean():7.0f}")
