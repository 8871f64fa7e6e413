#!/usr/bin/env python3
"""GEMM selection for padded candidate scoring (models whose layout rules out ragged rows: Gemma-3).

Padded scoring runs chunks whose candidate count is a multiple of attack.CHUNK_QUANTUM (utils.plan_chunk,
attack._score_candidates), so whatever the search width -- the dynamic schedule of BASELINE configs[4] walks
through ~385 of them -- the decoder meets one GEMM shape set per multiple up to the chunk cap.  This script
scores 8, 16, ... candidates through the real engine under PyTorch TunableOp in TUNING mode and merges the
winners into bimodalattack_amd/tuning/<arch>.csv (shapes already in the file are kept, not re-timed).

    python tools/tune_chunks.py [--workload gemma_joint] [--budget-s 900]

About a minute per chunk size on the 4B shape (seven products, ~150 candidate kernels each, V = 262208 in the
head): the sizes are walked full chunk first, then remainders from both ends inwards, the file is rewritten
after every size, and the walk stops when the time budget is spent -- run it again to go on where it stopped.
"""
import argparse
import os
import shutil
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gemma_joint")
    ap.add_argument("--budget-s", type=float, default=900.0)
    args = ap.parse_args()
    import time
    t_start = time.perf_counter()
    out_dir = os.path.join(REPO, "gpurun_out", "tune")
    os.makedirs(out_dir, exist_ok=True)
    results = os.path.join(out_dir, "tunableop_chunks.csv")
    os.environ.update(PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_FILENAME=results,
                      PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="50", PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="10",
                      BMA_GEMM_TUNING="off", MIOPEN_FIND_MODE="FAST")
    import torch
    arch = torch.cuda.get_device_properties(0).gcnArchName.split(":")[0]
    dst = os.path.join(REPO, "bimodalattack_amd", "tuning", f"{arch}.csv")
    src = results.replace(".csv", "0.csv")
    if os.path.exists(dst) and not os.path.exists(src):
        shutil.copyfile(dst, src)                    # only shapes the file lacks are searched
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack, logger
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import segment_order
    logger.setLevel("ERROR")
    dev = torch.device("cuda", 0)
    model, tok, proc, messages, goal, target, image, norm = build_plugins(args.workload, dev, torch.bfloat16, 32)
    cfg = BimodalAttackConfig(num_steps=1, search_width=512, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=True,
                              joint_eval=True, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, graph_scoring=False))
    atk._prepare_prompt(messages, target)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    order = segment_order("pgd", atk.hf.model_type, single=True)
    from bimodalattack_amd.attack import CHUNK_QUANTUM
    q = CHUNK_QUANTUM

    def score(n):
        cand = ids.repeat(n, 1)
        cand[:, 0] = torch.arange(n, device=dev) + 5
        before = atk.score_stats["padded_calls"]
        atk.score_candidates(cand.contiguous(), order, feats)
        torch.cuda.synchronize()
        if os.path.exists(src):                      # this PyTorch appends results as they are found
            shutil.copyfile(src, dst)
        return atk.score_stats["padded_calls"] - before

    with torch.no_grad():
        feats = atk.hf.image_features(image)
        chunks = score(512)                         # learns the chunk cap: 512 = chunks-1 full ones + a remainder
        cap = -(-512 // chunks) if chunks > 1 else 512
        cap -= cap % q
        print(f"chunk cap {cap} candidates; full chunk tuned in {time.perf_counter() - t_start:.0f} s", flush=True)
        sizes = list(range(q, cap, q))
        walk = []
        while sizes:                                 # from both ends inwards
            walk.append(sizes.pop())
            if sizes:
                walk.append(sizes.pop(0))
        for n in walk:
            if time.perf_counter() - t_start > args.budget_s:
                print(f"time budget spent before {n} candidates; run again to continue", flush=True)
                break
            score(n)
            print(f"  {n} candidates done at {time.perf_counter() - t_start:.0f} s", flush=True)
    print(f"walk finished; TunableOp writes {src} when the process exits -- copy it over {dst}", flush=True)


if __name__ == "__main__":
    main()
