#!/usr/bin/env python3
"""Which torch-side ops of one EAGER gradient pass launch copy / add / cat kernels, and from where:

    python3 tools/copy_probe.py --workload gemma_joint --min-elems 1000000

One pass under torch.profiler; every aten::copy_ / clone / contiguous / cat / add(_) whose first input has at least
--min-elems elements is listed with its input shapes and the chain of ops (and autograd nodes) that called it, folded by
that signature.  A trace by grid size says THAT a pass carries 4 strided copies per tower layer; this says who asks for them.
"""
import argparse
import collections
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch  # noqa: E402

WATCH = ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::add", "aten::add_", "aten::_to_copy",
         "aten::index_select", "aten::mul", "aten::zeros", "aten::zero_", "aten::fill_", "aten::sum")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gemma_joint", choices=["gcg", "joint", "pgd", "gemma_joint"])
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--min-elems", type=int, default=1_000_000)
    ap.add_argument("--depth", type=int, default=4)
    args = ap.parse_args()
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack, logger
    from bimodalattack_amd.config import EngineOptions
    logger.setLevel("ERROR")
    dev = torch.device("cuda", 0)
    pgd = args.workload != "gcg"
    only_pgd = args.workload == "pgd"
    model, tok, proc, messages, goal, target, image, norm = build_plugins("joint" if only_pgd else args.workload, dev,
                                                                          torch.bfloat16, args.layers)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=pgd, gcg_attack=not only_pgd,
                              joint_eval=pgd and not only_pgd, images_folder="/tmp/bma_gp")
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, graph_gradient=False, strict=True))
    atk._prepare_prompt(messages, target)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    if pgd:
        image.requires_grad_(True)
    for _ in range(2):
        atk.compute_gradient(ids, image if pgd else None)
    torch.cuda.synchronize(dev)
    from torch.profiler import ProfilerActivity, profile
    with profile(activities=[ProfilerActivity.CPU], record_shapes=True) as prof:
        atk.compute_gradient(ids, image if pgd else None)
        torch.cuda.synchronize(dev)
    fold = collections.Counter()
    for ev in prof.profiler.function_events:
        if ev.name not in WATCH:
            continue
        shapes = ev.input_shapes or []
        first = next((s for s in shapes if s), None)
        n = 1
        for d in (first or []):
            n *= d
        if first is None or n < args.min_elems:
            continue
        chain, p = [], ev.cpu_parent
        while p is not None and len(chain) < args.depth:
            chain.append(p.name.replace("autograd::engine::evaluate_function: ", "bwd:"))
            p = p.cpu_parent
        if chain and chain[0] in WATCH and chain[0] != ev.name:
            pass                                    # (kept: clone -> copy_ shows both; the fold makes that obvious)
        fold[(ev.name, str([s for s in shapes if s][:3]), " < ".join(chain))] += 1
    print(f"# {args.workload}: one eager gradient pass; ops with >= {args.min_elems} input elements, folded by (op, shapes, callers)")
    for (name, shapes, chain), c in sorted(fold.items(), key=lambda kv: -kv[1]):
        print(f"{c:5d}  {name:18s} {shapes:60s} {chain}")


if __name__ == "__main__":
    main()
