#!/usr/bin/env python3
"""Where the HOST spends a step (BMA_HOST_STAMPS=1 makes the engine stamp its hand-over points):

    BMA_EMULATE_WORLD=8 python3 tools/host_stamps.py [--workload gcg] [--steps 12]

Runs the attack as bench.py does (rank 0's share of a W-GPU step with BMA_EMULATE_WORLD), then folds the stamps: mean, median
and maximum host time between consecutive labels over the steps after the third.  At one GPU the 167 ms forward hides all of
it; at eight the forward is 24 ms and whatever the host needs between two launches beyond the queue's depth is GPU idle time.
"""
import argparse
import collections
import os
import statistics
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ["BMA_HOST_STAMPS"] = "1"
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gcg", choices=["gcg", "joint"])
    ap.add_argument("--steps", type=int, default=12)
    ap.add_argument("--layers", type=int, default=32)
    args = ap.parse_args()
    import bench
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack, logger
    from bimodalattack_amd.config import EngineOptions
    logger.setLevel("ERROR")
    dev = torch.device("cuda", 0)
    model, tok, proc, messages, goal, target, image, norm = bench.build_plugins(args.workload, dev, torch.bfloat16, args.layers)
    joint = args.workload == "joint"
    cfg = BimodalAttackConfig(num_steps=args.steps, search_width=512, seed=1, verbosity="ERROR", pgd_attack=joint, gcg_attack=True,
                              joint_eval=joint, early_stop=False, images_folder="/tmp/bma_hs")
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False))
    res = atk.run(messages, goal, target, image)
    torch.cuda.synchronize(dev)
    stamps = atk.host_stamps
    # cut into steps at "step"; drop the first three (warm-up: graph capture, first-use kernel loads)
    steps, cur = [], None
    for label, t in stamps:
        if label == "step":
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append((label, t))
    fold = collections.OrderedDict()
    period = []
    for a, b in zip(steps[3:-1], steps[4:]):
        period.append(1e3 * (b[0][1] - a[0][1]))
        seq = a + [("next step", b[0][1])]
        for (l0, t0), (l1, t1) in zip(seq, seq[1:]):
            fold.setdefault(f"{l0} -> {l1}", []).append(1e3 * (t1 - t0))
    print(f"# {args.workload}, emulate_world={os.environ.get('BMA_EMULATE_WORLD', '1')}: {len(period)} steps, period mean "
          f"{statistics.mean(period):.2f} ms (median {statistics.median(period):.2f}); final loss {res.losses[-1]:.4f}")
    print("#   segment (host clock)                 mean ms   median      max")
    for k, v in fold.items():
        print(f"    {k:36s} {statistics.mean(v):8.3f} {statistics.median(v):8.3f} {max(v):8.3f}")


if __name__ == "__main__":
    main()
