#!/usr/bin/env python3
"""Where a wave of bma_gemm_mid spends a unit of K (a -DBMA_MID_STAMPS build of the library; cdna_hip_programming.md 7,
"In-kernel stamps").

    make -C bimodalattack_amd/csrc OUTDIR=$PWD/bimodalattack_amd/lib_diag EXTRA=-DBMA_MID_STAMPS
    python tools/mid_stamps.py [--rows 644] [--n 22016] [--k 4096]

First the loop's own clock and the launch's anatomy (one launch with a start and an end stamp per wave only): shader cycles
against the 100 MHz counter over the k loop; when each of workgroups 0 / 100 / 200 and the LAST one (a second-round split
piece when the grid overflows the chip) entered, how long its prologue, k loop and epilogue took, when it left.  Then four
stamps per unit -- sub-steps 0-2 issued / this wave's DMA pieces of unit k+1 landed (vmcnt) / every fragment read landed
(lgkmcnt 0) / through B_k -- printed per wave, averaged over 30 units of the middle of the loop, in shader cycles: the unit,
then [issue of sub-step 3 + 0-2 of the next unit | vmcnt wait | lgkmcnt wait | barrier].  A stamp costs ~100 cycles.
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("BMA_LIB", os.path.join(REPO, "bimodalattack_amd", "lib_diag", "libbma_hip.so"))

import numpy as np  # noqa: E402
import torch  # noqa: E402

from bimodalattack_amd import ops  # noqa: E402
from bimodalattack_amd.native import lib  # noqa: E402

DEV = torch.device("cuda", 0)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=644)
    ap.add_argument("--n", type=int, default=22016)
    ap.add_argument("--k", type=int, default=4096)
    ap.add_argument("--flags", type=int, default=1)
    ap.add_argument("--out", default="/tmp/mid_stamps.bin")
    args = ap.parse_args()
    ops.GEMM_MID_MIN_K_OVER_N = 0.0
    ops.gemm_workspace(DEV)
    lib.bma_gemm_mid_set_plan(0, 0, -1, args.flags)
    g = torch.Generator(device=DEV).manual_seed(0)
    x = torch.randn((args.rows, args.k), generator=g, device=DEV).to(torch.bfloat16)
    ws = [(torch.randn((args.n, args.k), generator=g, device=DEV) * 0.02).to(torch.bfloat16) for _ in range(4)]
    for w in ws:                                                     # warm: clocks, code
        for _ in range(20):
            ops.gemm_mid(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    # first the loop's own clock: only a start and an end stamp per wave (cycles against the 100 MHz counter), after a warm-up
    # long enough for the chip to settle at the clock it holds under this load
    for _ in range(400):
        ops.gemm_mid(x, ws[_ % 4])
    os.environ["BMA_MID_STAMPS_FILE"] = args.out
    os.environ["BMA_MID_STAMPS_UNITS"] = "0"
    ops.gemm_mid(x, ws[0])
    torch.cuda.synchronize()
    full = np.fromfile(args.out, dtype=np.uint64).astype(np.int64).reshape(4, 8, 128)
    c = full[:3, :, 124:]
    ok = c[:, :, 3] > c[:, :, 1]
    cyc, ticks = (c[:, :, 2] - c[:, :, 0])[ok], (c[:, :, 3] - c[:, :, 1])[ok]
    if len(cyc):
        print(f"the k loop of workgroups 0 / 100 / 200: {np.median(cyc):.0f} shader cycles in {np.median(ticks) / 100.0:.1f} us = "
              f"{np.median(cyc / (ticks * 10.0)):.2f} GHz in-kernel clock (no per-unit stamps in this launch)")
    # the 100 MHz counter at kernel entry [122], loop start [125], loop end [127], after the epilogue's stores [123]
    t_first = min(int(full[g_, w_, 122]) for g_ in range(4) for w_ in range(8) if full[g_, w_, 122] > 0)
    for g_, name in enumerate(("workgroup 0", "workgroup 100", "workgroup 200", "the LAST workgroup (a second-round split piece when the grid overflows)")):
        e = full[g_, :, 122], full[g_, :, 125], full[g_, :, 127], full[g_, :, 123]
        if e[0].max() == 0:
            continue
        print(f"  {name}: enters at {(np.median(e[0]) - t_first) / 100.0:6.1f} us; prologue {np.median(e[1] - e[0]) / 100.0:5.1f} us, "
              f"k loop {np.median(e[2] - e[1]) / 100.0:5.1f} us, epilogue {np.median(e[3] - e[2]) / 100.0:5.1f} us; leaves at "
              f"{(np.median(e[3]) - t_first) / 100.0:6.1f} us")
    os.environ["BMA_MID_STAMPS_UNITS"] = "1"
    ops.gemm_mid(x, ws[0])
    os.environ["BMA_MID_STAMPS_FILE"] = ""
    torch.cuda.synchronize()
    e0.record()
    for _ in range(10):
        ops.gemm_mid(x, ws[1])
    e1.record()
    torch.cuda.synchronize()
    print(f"M={args.rows} N={args.n} K={args.k} flags={args.flags}: {1e2 * e0.elapsed_time(e1):.1f} us per launch (stamped build, stamps off in these)")
    a = np.fromfile(args.out, dtype=np.uint64).astype(np.int64).reshape(4, 8, 128)[:3]
    names, per_unit, units = ["issue", "vmcnt", "lgkmcnt", "barrier"], 4, 30
    ns = len(names)
    for wg in range(3):
        if a[wg].max() == 0:
            continue
        print(f"workgroup {wg * 100}:")
        for wave in range(8):
            flat = a[wg, wave, :per_unit * units]
            d = np.diff(flat)                                        # interval j ends at stamp j+1
            groups = per_unit // ns
            per = {n: [[] for _ in range(groups)] for n in names}
            for j, dt in enumerate(d):
                end = (j + 1) % ns
                ph = ((j + 1) // ns) % groups
                per[names[end]][ph].append(dt)
            unit = (flat[per_unit * (units - 1)] - flat[0]) / (units - 1.0)
            cells = [" ".join(f"{np.mean(per[n][ph]):5.0f}" for n in names) for ph in range(groups)]
            print(f"  wave {wave}: unit {unit:6.0f} cyc | " + " | ".join(cells))
        print("           per group: " + " / ".join(names))


if __name__ == "__main__":
    main()
