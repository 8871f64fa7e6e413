#!/usr/bin/env python3
"""The batch-1 gradient pass alone (the serial part of a multi-GPU step), for a kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d OUT -o gp -- python3 tools/grad_pass_profile.py --workload gcg
    python3 tools/trace_by_grid.py OUT/gp_kernel_trace.csv OUT/gp_by_grid.txt 60 --between-markers

`--iters` replays of the captured hipGraph (or eager passes with --eager) run between two marker kernels
(a float64 sum over 12345 zeros), so the fold can drop model construction and warm-up.  Prints wall ms per pass.
"""
import argparse
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch  # noqa: E402


def marker(dev):
    torch.zeros(12345, device=dev, dtype=torch.float64).sum()
    torch.cuda.synchronize(dev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gcg", choices=["gcg", "joint", "pgd", "gemma_joint"])
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--eager", action="store_true")
    ap.add_argument("--layers", type=int, default=32)
    args = ap.parse_args()
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack, logger
    from bimodalattack_amd.config import EngineOptions
    logger.setLevel("ERROR")
    dev = torch.device("cuda", 0)
    pgd = args.workload != "gcg"
    only_pgd = args.workload == "pgd"            # BASELINE configs[1]: the whole step is this pass (643 rows, pixels only)
    model, tok, proc, messages, goal, target, image, norm = build_plugins("joint" if only_pgd else args.workload, dev,
                                                                          torch.bfloat16, args.layers)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=pgd, gcg_attack=not only_pgd,
                              joint_eval=pgd and not only_pgd, images_folder="/tmp/bma_gp")
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, graph_gradient=not args.eager,
                                                                            strict=True))
    atk._prepare_prompt(messages, target)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    if pgd:
        image.requires_grad_(True)
    for _ in range(3):
        atk.compute_gradient(ids, image if pgd else None)
    torch.cuda.synchronize(dev)
    marker(dev)
    t0 = time.perf_counter()
    for _ in range(args.iters):
        atk.compute_gradient(ids, image if pgd else None)
    torch.cuda.synchronize(dev)
    ms = 1e3 * (time.perf_counter() - t0) / args.iters
    marker(dev)
    print(f"{args.workload}: gradient pass {ms:.2f} ms ({'eager' if args.eager else 'hipGraph replay'}), {args.iters} passes")


if __name__ == "__main__":
    main()
