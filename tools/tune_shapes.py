#!/usr/bin/env python3
"""GEMM selection for an explicit list of shapes: every (rows, out, in) product of the lists given, through
`torch.nn.functional.linear` under PyTorch TunableOp in tuning mode; winners are merged into
bimodalattack_amd/tuning/<arch>.csv (shapes already there are kept).  For filling holes the engine-driven tuners
(tune_gemms.py, tune_chunks.py) left, e.g. the Gemma-3-4b chunk sizes of 128..152 candidates:

    python tools/tune_shapes.py --rows 38144,40528,42912,45296 --shapes 2048x2560,1024x2560,2560x2048,20480x2560,2560x10240
    python tools/tune_shapes.py --rows 2560,2720,2880,3040 --shapes 262208x2560
"""
import argparse
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", required=True, help="comma-separated row counts (M)")
    ap.add_argument("--shapes", required=True, help="comma-separated OUTxIN weight shapes")
    ap.add_argument("--ms", default="60", help="tuning time per candidate kernel")
    ap.add_argument("--two-calls", action="store_true",
                    help="tune the calls fused.two_calls makes for each shape (row / column slices of one output: their own keys, "
                         "the leading dimension of a column slice is the whole output's) instead of the one-call product")
    args = ap.parse_args()
    out_dir = os.path.join(REPO, "gpurun_out", "tune")
    os.makedirs(out_dir, exist_ok=True)
    results = os.path.join(out_dir, "tunableop_shapes.csv")
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=results,
                      PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=args.ms, PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="10")
    import torch
    arch = torch.cuda.get_device_properties(0).gcnArchName.split(":")[0]
    dst = os.path.join(REPO, "bimodalattack_amd", "tuning", f"{arch}.csv")
    src = results.replace(".csv", "0.csv")
    if os.path.exists(dst) and not os.path.exists(src):
        shutil.copyfile(dst, src)
    import time
    t0 = time.perf_counter()
    rows = [int(r) for r in args.rows.split(",")]
    shapes = [tuple(int(v) for v in s.split("x")) for s in args.shapes.split(",")]
    dev, dt = "cuda", torch.bfloat16
    for (n, k) in shapes:
        w = (torch.randn((n, k), device=dev) * 0.02).to(dt)
        for m in rows:
            x = torch.randn((1, m, k), device=dev).to(dt)
            if args.two_calls:
                from bimodalattack_amd import fused
                with torch.no_grad():
                    if fused.two_calls(x, w) is None:
                        print(f"  {m} x {n} x {k}: one call (nothing to tune here)", flush=True)
                        continue
            else:
                torch.nn.functional.linear(x, w)
            torch.cuda.synchronize()
            print(f"  {m} x {n} x {k} done at {time.perf_counter() - t0:.0f} s", flush=True)
    print(f"TunableOp writes {src} when the process exits; copy it over {dst}", flush=True)


if __name__ == "__main__":
    main()
