#!/usr/bin/env python3
"""Offline GEMM selection for the attack's shapes (run on an MI355X):

    python tools/tune_gemms.py            # ~15 GPU-minutes; writes bimodalattack_amd/tuning/<arch>.csv
    BMA_TUNE_WORLDS=1 python tools/tune_gemms.py gemma_joint    # one workload, one world size

Runs bench.py under PyTorch TunableOp in tuning mode for the per-rank shapes of 1/2/4/8 GPUs
(one process emulating rank 0 of each world size) on the GCG-only and joint workloads.  The gradient pass is run eagerly here (tuning cannot happen inside a
graph capture); at run time its captured graph picks the tuned kernels up by lookup.
"""
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    import torch
    arch = torch.cuda.get_device_properties(0).gcnArchName.split(":")[0]
    out_dir = os.path.join(REPO, "gpurun_out", "tune")
    os.makedirs(out_dir, exist_ok=True)
    results = os.path.join(out_dir, "tunableop_results.csv")
    env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1",
               PYTORCH_TUNABLEOP_FILENAME=results, PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="100",
               PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="20", BMA_GRAPH_GRADIENT="0", BMA_GEMM_TUNING="off",
               MIOPEN_FIND_MODE="FAST")
    workloads = sys.argv[1:] or ["gcg", "joint"]
    dst = os.path.join(REPO, "bimodalattack_amd", "tuning", f"{arch}.csv")
    src = results.replace(".csv", "0.csv")
    if os.path.exists(dst) and not os.path.exists(src):
        shutil.copyfile(dst, src)            # keep what is already tuned: only new shapes are searched
    for wl in workloads:
        for world in [int(w) for w in os.environ.get("BMA_TUNE_WORLDS", "1,2,4,8").split(",")]:
            # rank 0's share of a `world`-GPU run, in one process (attack.EMULATE_WORLD, BMA_EMULATE_WORLD): the shapes
            # candidate dealing produces, not those of a smaller search width
            print(f"== tuning {wl} as rank 0 of {world}", flush=True)
            r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--workload", wl, "--steps", "1", "--warmup", "1",
                                "--no-cpu-baseline"], env=dict(env, BMA_EMULATE_WORLD=str(world)), cwd=REPO)
            if r.returncode != 0:
                print(f"   (failed with {r.returncode}; continuing)", flush=True)
    if not sys.argv[1:] or "gcg" in workloads:
        print("== tuning the row-count grid of ragged scoring", flush=True)
        subprocess.run([sys.executable, os.path.join(REPO, "tools", "tune_rows.py")], env=env, cwd=REPO)
    shutil.copyfile(src, dst)
    shutil.copyfile(src, os.path.join(out_dir, f"{arch}.csv"))
    print("wrote", dst, sum(1 for _ in open(dst)), "lines")


if __name__ == "__main__":
    main()
