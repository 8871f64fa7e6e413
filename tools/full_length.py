#!/usr/bin/env python3
"""Run one BASELINE config at its STATED length and write the evidence ("next round" item 2).

    python tools/full_length.py --workload pgd|gcg|joint|gemma_joint [--steps N] --out profiles/r6_full_<name>.json

BASELINE.json: configs[1] PGD-only 100 steps, configs[2] GCG-only 250, configs[3] joint 600, configs[4] Gemma-3 joint with
dynamic_search 512 -> 128 over 600 (the reference's own schedule, bimodal_attack.py:919-923 -- no width override here).
Recorded: per-step wall time (host clock at the engine's step hook: the engine reads one packed outcome back per step, so
consecutive hooks are a step apart; no extra synchronisation is added), first 10 / median / last 10 / worst steps,
torch.cuda.max_memory_allocated and memory_reserved every 100 steps, every (rows, N, K) product shape that reaches
torch.nn.functional.linear FIRST inside the run proper (after step 1: a first-sight shape costs the library a lazy code
object load, 39 ms in NOTEBOOK r5), engine_state()["fallbacks"], and whether every loss is finite.
"""
import argparse
import json
import math
import os
import statistics
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

import bench  # noqa: E402

STATED = {"pgd": 100, "gcg": 250, "joint": 600, "gemma_joint": 600}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", required=True, choices=sorted(STATED))
    ap.add_argument("--steps", type=int, default=None)
    ap.add_argument("--out", required=True)
    ap.add_argument("--layers", type=int, default=32)
    args = ap.parse_args()
    import torch
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack, logger as gcg_logger
    from bimodalattack_amd.config import EngineOptions

    wl = bench.WORKLOADS[args.workload]
    steps = args.steps or STATED[args.workload]
    device = torch.device("cuda", 0)
    torch.cuda.set_device(device)
    model, tok, proc, messages, goal, target, image, norm = bench.build_plugins(args.workload, device, torch.bfloat16, args.layers)
    cfg_kw = dict(search_width=512, topk=256, n_replace=1, seed=1, verbosity="ERROR", pgd_attack=wl["pgd_attack"],
                  gcg_attack=wl["gcg_attack"], joint_eval=wl["joint_eval"], eps=64 / 255, alpha=4 / 255)
    if wl.get("gemma"):
        cfg_kw.update(dynamic_search=True, min_search_width=128)
    cfg = BimodalAttackConfig(num_steps=steps, images_folder=tempfile.mkdtemp(prefix="bma_full_"), **cfg_kw)

    # every product shape that reaches the library through torch.nn.functional.linear, with the step it was first seen in
    F = torch.nn.functional
    plain_linear = F.linear
    seen, state = {}, {"step": -1}

    def spy_linear(x, w, b=None):
        k = (x.numel() // x.shape[-1], w.shape[0], w.shape[1])
        if k not in seen:
            seen[k] = state["step"]
        return plain_linear(x, w, b)

    F.linear = spy_linear
    stamps, mem = [], []

    def hook(i: int) -> None:
        stamps.append(time.perf_counter())
        state["step"] = i
        if i % 100 == 0 or i == steps:
            mem.append(dict(step=i, max_allocated_GB=torch.cuda.max_memory_allocated(device) / 1e9,
                            reserved_GB=torch.cuda.memory_reserved(device) / 1e9, allocated_GB=torch.cuda.memory_allocated(device) / 1e9))
            bench.log(f"{args.workload}: step {i}/{steps}  reserved {mem[-1]['reserved_GB']:.1f} GB")

    gcg_logger.setLevel("ERROR")
    attack = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(step_hook=hook, save_images=False))
    t0 = time.perf_counter()
    res = attack.run(messages, goal, target, image)
    torch.cuda.synchronize(device)
    wall = time.perf_counter() - t0
    F.linear = plain_linear
    ms = [1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:])]
    late = sorted(((s, k) for k, s in seen.items() if s >= 2), key=lambda t: t[0])
    med = statistics.median(ms[2:]) if len(ms) > 2 else None
    worst = sorted(range(len(ms)), key=lambda i: -ms[i])[:8]
    widths = attack.n_scored
    out = dict(
        workload=wl["name"], steps_stated=STATED[args.workload], steps_run=len(ms), wall_s_incl_setup=wall,
        ms_per_step=dict(first10=[round(v, 2) for v in ms[:10]], median_after_2=med, mean_after_2=statistics.mean(ms[2:]) if len(ms) > 2 else None,
                         last10=[round(v, 2) for v in ms[-10:]], worst=[dict(step=i, ms=round(ms[i], 2)) for i in worst],
                         steps_over_1p3x_median=[i for i in range(2, len(ms)) if med and ms[i] > 1.3 * med]),
        candidates_scored=dict(first=widths[:3], last=widths[-3:], total=sum(widths), distinct_counts=len(set(widths))),
        candidate_forwards_per_s_after_2=(sum(widths[2:]) / (sum(ms[2:]) * 1e-3)) if len(ms) > 2 and wl["gcg_attack"] else None,
        memory=mem,
        gemm_shapes=dict(distinct=len(seen), first_seen_at_step_2_or_later=[dict(step=s, rows=k[0], N=k[1], K=k[2]) for s, k in late][:64],
                         count_first_seen_at_step_2_or_later=len(late)),
        engine=attack.engine_state(),
        losses=dict(first=[float(v) for v in res.losses[:3]], last=[float(v) for v in res.losses[-3:]], best=float(res.best_loss),
                    finite=bool(all(math.isfinite(float(v)) for v in res.losses))),
        all_ms=[round(v, 2) for v in ms],
    )
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    json.dump(bench._strict(out), open(args.out, "w"), indent=1)
    print(json.dumps(dict(workload=args.workload, steps=len(ms), median_ms=med, first10=out["ms_per_step"]["first10"][:4],
                          last10=out["ms_per_step"]["last10"][-3:], over_1p3x=len(out["ms_per_step"]["steps_over_1p3x_median"]),
                          late_shapes=len(late), reserved_GB=mem[-1]["reserved_GB"] if mem else None, fallbacks=out["engine"].get("fallbacks"),
                          finite=out["losses"]["finite"])))


if __name__ == "__main__":
    main()
