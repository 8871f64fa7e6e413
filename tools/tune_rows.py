#!/usr/bin/env python3
"""GEMM selection for the row counts ragged scoring produces.

A ragged scoring forward computes `ragged_rows(needed)` rows: what the step's draw needs, on a coarse
grid (layout.py).  The rows needed vary from step to step (first replaced positions and duplicates are
random), so the candidate forward meets a small SET of GEMM shapes; this script walks that set -- every
grid point within +-4.5 sigma of the expected row count, for one GPU and for rank 0 of 2 / 4 / 8 -- through
the decoder's projection shapes under PyTorch TunableOp in tuning mode.  Run by tools/tune_gemms.py with
the TunableOp environment set; standalone:

    PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_FILENAME=out.csv \\
        python tools/tune_rows.py [--search-width 512 --n-opt 19 --tail 44 --topk 256]
"""
import argparse
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def row_counts(m, n_opt, L, n_replace, topk, worlds=(1, 2, 4, 8), sigmas=4.5):
    from bimodalattack_amd.layout import expected_row_counts
    return expected_row_counts(m, n_opt, L, n_replace, topk, worlds, sigmas)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--search-width", type=int, default=512)
    ap.add_argument("--n-opt", type=int, default=19)
    ap.add_argument("--tail", type=int, default=44, help="tokens per candidate behind the shared prefix")
    ap.add_argument("--n-replace", type=int, default=1)
    ap.add_argument("--topk", type=int, default=256)
    ap.add_argument("--hidden", type=int, default=4096)
    ap.add_argument("--intermediate", type=int, default=11008)
    ap.add_argument("--kv-hidden", type=int, default=4096)
    args = ap.parse_args()
    import torch
    dev, dt = "cuda", torch.bfloat16
    D, I, KV = args.hidden, args.intermediate, args.kv_hidden
    shapes = [(D + 2 * KV, D), (D, D), (I, D), (2 * I, D), (D, I)]   # fused q/k/v, o_proj, gate/up (alone, fused), down  as (out, in)
    weights = {s: (torch.randn(s, device=dev) * 0.02).to(dt) for s in shapes}
    counts = row_counts(args.search_width, args.n_opt, args.tail, args.n_replace, args.topk)
    print(f"{len(counts)} row counts x {len(shapes)} projection shapes: {counts}", flush=True)
    for M in counts:
        for s in shapes:
            x = torch.randn((1, M, s[1]), device=dev).to(dt)
            torch.nn.functional.linear(x, weights[s])
        torch.cuda.synchronize()
        print(f"  rows {M} done", flush=True)


if __name__ == "__main__":
    main()
