#!/usr/bin/env python3
"""Re-tune the library GEMMs of the batch-1 passes (gradient pass, prefix pass, winner re-score: every product with
at most 1024 rows) with COLD operands.

    python tools/tune_b1.py [--workloads gcg,joint,pgd,pgd_gcg,gemma_joint] [--rotate-mb 1024] [--ms 30]

Why: PyTorch's TunableOp times every rocBLAS / hipBLASLt solution of a shape on ONE set of operand buffers by default
(its rotating buffer covers the L2, 4 MB per XCD).  A batch-1 product streams 34-180 MB of weights once; timed on one
buffer those weights sit in the 256 MB Infinity Cache from the second iteration on, and the table's winners were
picked -- and timed 30-40 % too fast -- on cache-resident weights (round 2's table: gate/up at 65 rows 34.5 us in
tuning, 46.5 us inside the pass, where every layer's weights come from HBM).  With a rotating buffer larger than the
Infinity Cache every timed iteration reads operands that have left it, which is what the pass does.

The tool drops the <= 1024-row entries of bimodalattack_amd/tuning/<arch>.csv, runs one eager attack step of each
workload under TunableOp in tuning mode (graphs off: tuning cannot happen inside a capture) and writes the merged table
back.  Entries of the candidate forward (thousands of rows: their operands do not fit the cache anyway) are kept.
"""
import argparse
import os
import re
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workloads", default="gcg,joint,pgd,pgd_gcg,gemma_joint")
    ap.add_argument("--rotate-mb", type=int, default=1024)
    ap.add_argument("--ms", default="30", help="tuning time per candidate kernel")
    ap.add_argument("--max-rows", type=int, default=1024)
    ap.add_argument("--arch", default="gfx950")
    args = ap.parse_args()
    out_dir = os.path.join(REPO, "gpurun_out", "tune_b1")
    os.makedirs(out_dir, exist_ok=True)
    results = os.path.join(out_dir, "tunableop_b1.csv")
    seed = results.replace(".csv", "0.csv")
    dst = os.path.join(REPO, "bimodalattack_amd", "tuning", f"{args.arch}.csv")
    kept = dropped = 0
    with open(dst) as f, open(seed, "w") as g:
        for line in f:
            m = re.match(r"GemmTunableOp_\w+,[a-z]{2}_(\d+)_(\d+)_(\d+)_", line)
            if m and int(m.group(2)) <= args.max_rows:
                dropped += 1
                continue
            g.write(line)
            kept += 1
    shutil.copyfile(dst, os.path.join(out_dir, f"{args.arch}_before.csv"))
    print(f"seed: kept {kept} lines, dropped {dropped} entries with <= {args.max_rows} rows", flush=True)
    env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=results,
               PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=args.ms, PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="5",
               PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE=str(args.rotate_mb),
               BMA_GRAPH_GRADIENT="0", BMA_GRAPH_SCORING="0", BMA_GEMM_TUNING="off", MIOPEN_FIND_MODE="FAST",
               BMA_SKINNY_GEMM="0")          # every product through the library, so that every shape gets an entry
    for wl in [w for w in args.workloads.split(",") if w]:
        print(f"== tuning the batch-1 products of {wl} (rotating buffer {args.rotate_mb} MB)", flush=True)
        r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", wl, "--steps", "1", "--warmup", "1",
                            "--profile-steps", "0", "--no-cpu-baseline", "--extra-workloads", "none"], env=env, cwd=REPO,
                           stdout=subprocess.DEVNULL)
        print(f"   rc={r.returncode}; table now {sum(1 for _ in open(seed))} lines", flush=True)
    shutil.copyfile(seed, dst)
    shutil.copyfile(seed, os.path.join(out_dir, f"{args.arch}_after.csv"))
    print("wrote", dst, sum(1 for _ in open(dst)), "lines")


if __name__ == "__main__":
    main()
