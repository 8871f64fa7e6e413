#!/usr/bin/env python3
"""Read the per-wave clock stamps of a diagnostic build of the long-block attention kernel
(csrc/ragged_attention.hip under -DBMA_LONG_STAMPS, BMA_RAGGED_STAMPS=<file>): where a wave's cycles go.

    python tools/stamps_report.py gpurun_out/stamps.bin [waves per workgroup]
"""
import sys

import numpy as np

NW = int(sys.argv[2]) if len(sys.argv) > 2 else 8
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 12).astype(np.int64)
wave = np.tile(np.arange(NW), len(a) // NW)
keep = a[:, 0] > 0
a, wave = a[keep], wave[keep]
t0, t1, t2, t3, wait, chunks, r0, r1, qk, sm, pv, mine = a.T
ghz = (t3 - t0).sum() / ((r1 - r0).sum() * 10.0)          # s_memrealtime ticks at 100 MHz
print(f"{len(a)} waves; in-kernel clock {ghz:.2f} GHz; kernel span {(r1.max() - r0.min()) / 100.0:.1f} us")
print(f"per wave (cycles): total {(t3 - t0).mean():8.0f} = prologue {(t1 - t0).mean():7.0f} + loop {(t2 - t1).mean():8.0f} + epilogue {(t3 - t2).mean():7.0f};"
      f"  stages walked {chunks.mean():.2f}, multiplied {mine.mean():.2f}")
print(f"loop: wait+barrier+issue {wait.mean():8.0f}  QK {qk.mean():8.0f}  softmax {sm.mean():8.0f}  PV {pv.mean():8.0f}"
      f"  -> per multiplied stage: QK {qk.sum() / mine.sum():6.0f}  softmax {sm.sum() / mine.sum():6.0f}  PV {pv.sum() / mine.sum():6.0f}")
for w in range(NW):
    m = wave == w
    print(f"  wave {w}: multiplied {mine[m].mean():5.2f} of {chunks[m].mean():5.2f}  loop {(t2 - t1)[m].mean():8.0f}  wait {wait[m].mean():8.0f}"
          f"  QK {qk[m].mean():7.0f} softmax {sm[m].mean():7.0f} PV {pv[m].mean():7.0f}  prologue {(t1 - t0)[m].mean():6.0f} epilogue {(t3 - t2)[m].mean():6.0f}")
