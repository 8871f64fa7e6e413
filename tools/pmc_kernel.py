#!/usr/bin/env python3
"""SQ / LDS / TA counters of one tools/kernel_bench.py case, one rocprofv3 --pmc pass per counter group.

    python tools/pmc_kernel.py --only ragged_attn/gemma --kernel ragged_attn --out gpurun_out/pmc_ragged [--env BMA_RAGGED_LONG=0]

Runs `rocprofv3 --pmc <group> --kernel-trace -- python3 tools/kernel_bench.py --only <case> --iters 3` per group (counters
in their own runs, kernel trace only: MI355X_MICROARCH.md's profiling recipe), keeps the launches whose kernel name
contains --kernel and prints the per-launch mean of every counter.  Groups whose counters the box does not list are
skipped, not guessed.
"""
import argparse
import csv
import glob
import os
import subprocess
import sys

GROUPS = [
    ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SALU"],
    ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_ACTIVE_INST_MISC"],
    ["SQ_WAIT_INST_LDS", "SQ_INST_CYCLES_VMEM", "SQ_INST_CYCLES_SALU", "SQ_THREAD_CYCLES_VALU"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CU_CYCLES", "GRBM_GUI_ACTIVE"],
    ["TA_BUSY_avr", "TA_TA_BUSY_sum", "TCP_PENDING_STALL_CYCLES_sum", "TCC_HIT_sum", "TCC_MISS_sum"],
    ["SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_INST_LEVEL_VMEM", "SQ_INST_LEVEL_LDS", "SQ_LEVEL_WAVES"],
    ["FETCH_SIZE"], ["WRITE_SIZE"],
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", required=True)
    ap.add_argument("--kernel", required=True)
    ap.add_argument("--out", required=True)
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--iters", default="3")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    env = dict(os.environ)
    for kv in args.env:
        k, v = kv.split("=", 1)
        env[k] = v
    avail = subprocess.run(["rocprofv3", "-L"], capture_output=True, text=True, env=env).stdout
    open(os.path.join(args.out, "avail.txt"), "w").write(avail)
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    summary = {}
    for gi, group in enumerate(GROUPS):
        have = [c for c in group if c in avail]
        if not have:
            print(f"group {gi}: none of {group} listed, skipped", flush=True)
            continue
        d = os.path.join(args.out, f"g{gi}")
        cmd = ["rocprofv3", "--pmc", *have, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", "p", "--",
               "python3", os.path.join(repo, "tools", "kernel_bench.py"), "--only", args.only, "--iters", args.iters, "--warmup", "1"]
        r = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, timeout=300)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode or not files:
            print(f"group {gi} {have}: rc={r.returncode}, no counters ({r.stderr[-300:]})", flush=True)
            continue
        acc, n = {}, {}
        for row in csv.DictReader(open(files[0])):
            if args.kernel not in row["Kernel_Name"]:
                continue
            c = row["Counter_Name"]
            acc[c] = acc.get(c, 0.0) + float(row["Counter_Value"])
            n[c] = n.get(c, 0) + 1
        for c in acc:
            summary[c] = acc[c] / n[c]
            print(f"{c:34s} {summary[c]:16.1f}   ({n[c]} launches)", flush=True)
    import json
    json.dump(summary, open(os.path.join(args.out, "summary.json"), "w"), indent=1)


if __name__ == "__main__":
    sys.exit(main())
