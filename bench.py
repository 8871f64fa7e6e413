#!/usr/bin/env python3
"""bench.py -- the attack-step benchmark of BASELINE.json on synthetic random-weight
LLaVA-1.5-7B (bf16), search_width = 512.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gcg|joint|pgd|pgd_gcg]

A "step" is one full pass of the hot path: gradient pass -> (PGD projection) ->
mask/top-k/scatter sampling -> retokenisation filter -> candidate splice + forward +
target cross-entropy -> argmin + bookkeeping.  Inputs (weights, embeddings, prompt
segments, image) are resident in HBM before the timed region.  With N > 1 (launched by
torch.distributed.run, one rank per GPU over RCCL) the candidates of every step are
sharded across ranks; the work per step is fixed, so scaling is "strong".

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement").  `value` is whole-step
candidate forwards per second (all ranks, wall clock, max over ranks); the per-phase
rate the reference's tables quote (search_width / loss-phase seconds) is given beside it.
"""

from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)
# the CLIP patch-embedding convolution would otherwise trigger MIOpen's exhaustive kernel search on a
# fresh machine (~2 minutes before the first PGD step); must be set before MIOpen initialises
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

_T0 = time.perf_counter()


def log(msg: str) -> None:
    """Progress on stderr (a run that stays silent for minutes is taken to be hung)."""
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:6.1f}s] {msg}", file=sys.stderr, flush=True)


HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 is the measured copy ceiling
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 (not the 2:1-sparse headline)

# SURVEY.md 8, top: fixed synthetic segment lengths
SEG = dict(
    gcg=dict(before=21, optim=19, after=6, target=20),                              # S = 66
    joint=dict(before_img=5, n_img=576, before_suffix=18, optim=19, after=6, target=20),   # S = 644
)
WORKLOADS = {
    "gcg": dict(pgd_attack=False, gcg_attack=True, joint_eval=False,
                name="GCG-only, LLaVA-1.5-7B-shaped bf16, search_width=512 (BASELINE configs[2])"),
    "joint": dict(pgd_attack=True, gcg_attack=True, joint_eval=True,
                  name="Joint GCG+PGD joint_eval, LLaVA-1.5-7B-shaped bf16, search_width=512 (BASELINE configs[3])"),
    "pgd_gcg": dict(pgd_attack=True, gcg_attack=True, joint_eval=False,
                    name="PGD+GCG non-joint, LLaVA-1.5-7B-shaped bf16, search_width=512"),
    "pgd": dict(pgd_attack=True, gcg_attack=False, joint_eval=False,
                name="PGD-only, LLaVA-1.5-7B-shaped bf16 (BASELINE configs[1])"),
    "gemma_joint": dict(pgd_attack=True, gcg_attack=True, joint_eval=True, gemma=True,
                        name="Joint GCG+PGD, Gemma-3-4b-it-shaped bf16, dynamic_search 512->128 (BASELINE configs[4])"),
}


def build_plugins(workload: str, device, dtype, layers: int):
    """Synthetic tokenizer (32000 printable-ASCII words; embedding table has 32064 rows),
    LLaVA-1.5-7B-shaped random-weight model, prompt strings of the fixed segment lengths."""
    from bimodalattack_amd import synthetic as S
    if workload == "gemma_joint":
        # Gemma-3 layout 20|19|3|256|6|20: suffix in FRONT of the image
        tok = S.build_tokenizer(262144, 0, 0)
        tok.chat_template = S.GEMMA_TEMPLATE
        proc = S.Gemma3Processor(tok, S.GEMMA_TEMPLATE)
        log("tokenizer built; building Gemma-3-4b-shaped model on the device")
        model = S.gemma3_4b_shaped(dtype=dtype, device=device, seed=0)
        log(f"model ready on {model.device} ({sum(p.numel() for p in model.parameters()) / 1e9:.2f} B parameters)")
        goal, target = S.synthetic_prompt(tok, 18, 20, seed=0)        # <start_of_turn>user + 18 + BOS = 20
        image = S.synthetic_image(896, 896, seed=0, device=device)
        return model, tok, proc, goal, goal, target, image, S.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    tok = S.build_tokenizer(32000, 0, 0)
    proc = S.SyntheticProcessor(tok)
    log("tokenizer built; building LLaVA-1.5-7B-shaped model on the device")
    model = S.llava_15_7b_shaped(dtype=dtype, device=device, seed=0, text_layers=layers)
    log(f"model ready on {model.device} ({sum(p.numel() for p in model.parameters()) / 1e9:.2f} B parameters)")
    if workload == "gcg":
        # GCG-only template renders the bare content: before = BOS + goal tokens
        n_goal, after_txt = SEG["gcg"]["before"] - 1, SEG["gcg"]["after"]
        # the 6 "after" tokens ride in the message after the placeholder
        goal, target = S.synthetic_prompt(tok, n_goal, SEG["gcg"]["target"], seed=0)
        after, _ = S.synthetic_prompt(tok, after_txt, 1, seed=1)
        messages = f"{goal} {{optim_str}} {after}"
        image = None
    else:
        # PGD template "USER: <image>\n{text} \nASSISTANT: ": before_img = BOS + "USER:",
        # before_suffix = BOS + goal, after = trailing words + "ASSISTANT:"
        seg = SEG["joint"]
        goal, target = S.synthetic_prompt(tok, seg["before_suffix"] - 1, seg["target"], seed=0)
        after, _ = S.synthetic_prompt(tok, seg["after"] - 1, 1, seed=1)
        messages = f"{goal} {{optim_str}} {after}"
        # the real Llama tokenizer spends 5 tokens on "<s>USER: "; the word-level stand-in
        # spends 2, so three filler words keep the segment lengths of SURVEY.md 8
        fill, _ = S.synthetic_prompt(tok, seg["before_img"] - 2, 1, seed=2)
        tpl = fill + " USER: <image>\n{{ messages[0]['content'][0]['text'] }} \nASSISTANT: "
        tok.chat_template = tpl
        proc.chat_template = tpl
        image = S.synthetic_image(336, 336, seed=0, device=device)
    norm = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)
    return model, tok, proc, messages, goal, target, image, norm


def cpu_baseline(args, model, tok, proc, messages, goal, target, image, norm, cfg_kw):
    """The oracle (CPU restatement of the reference loop, pinned by the reference's
    goldens) timed on this box's host cores on a bounded sample of the same workload:
    same model shape and prompt, 1 step at search_width = args.cpu_width."""
    import torch
    from bimodalattack_amd.config import BimodalAttackConfig
    from oracle.attack_loop import run_oracle          # the checker / baseline, never the product path

    cores = os.cpu_count() or 1
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = cores
    # a 1-GPU box owns a 16-core share of the host whatever cpu_count() says; more threads
    # than that oversubscribe the share and run slower
    threads = args.cpu_threads if args.cpu_threads else min(usable, 16)
    torch.set_num_threads(threads)
    log(f"cpu baseline: moving the model to the host ({threads} threads)")
    t0 = time.perf_counter()
    cmodel = model.to("cpu")                 # the GPU measurement is over: move, do not copy
    cimage = None if image is None else image.detach().clone().cpu()
    # bf16 GEMMs are slow on hosts without AVX512-BF16/AMX: probe, and fall back to fp32
    a = torch.randn(512, 4096).to(cmodel.dtype)
    b = torch.randn(4096, 4096).to(cmodel.dtype)
    (a @ b)
    tp = time.perf_counter()
    (a @ b)
    rate = 2 * 512 * 4096 * 4096 / (time.perf_counter() - tp)
    cpu_dtype = cmodel.dtype
    if rate < 2e11:
        cmodel = cmodel.float()
        cpu_dtype = torch.float32
    t_copy = time.perf_counter() - t0
    log(f"cpu baseline: host GEMM probe {rate / 1e12:.2f} TFLOP/s in {cmodel.dtype}; running the oracle loop in {cpu_dtype}")
    kw = dict(cfg_kw, num_steps=args.cpu_steps, search_width=args.cpu_width, images_folder=tempfile.mkdtemp(prefix="bma_cpu_"))
    t0 = time.perf_counter()
    res, trace, _ = run_oracle(cmodel, tok, proc, messages, goal, target, cimage, BimodalAttackConfig(**kw), normalize=norm)
    wall = time.perf_counter() - t0
    log(f"cpu baseline: done in {wall:.1f} s")
    n = sum(len(st["losses"][0]) if st["losses"] else 1 for st in trace)
    loss_s = sum(res["loss_times"])
    try:
        cpu_name = next(l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name"))
    except Exception:
        cpu_name = "unknown"
    return {
        "value": n / wall, "unit": "candidate_forwards/s", "cores": threads, "kind": "port",
        "sample": f"{args.cpu_steps} step(s) at search_width={args.cpu_width} (reduced from 512), same model shape, "
                  f"prompt, host dtype {str(cpu_dtype).replace('torch.', '')}, incl. init-buffer scoring; "
                  f"{n} candidates in {wall:.1f} s",
        "attack_steps_per_s": args.cpu_steps / wall, "scoring_phase_cand_per_s": n / loss_s if loss_s else None,
        "gradient_pass_s": sum(res["gradient_times"]) / max(1, len(res["gradient_times"])),
        "cpu_model": cpu_name, "host_cores_total": cores, "copy_to_host_s": round(t_copy, 2),
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="gcg", choices=sorted(WORKLOADS))
    ap.add_argument("--search-width", type=int, default=512)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer layers is NOT the benchmark")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-width", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-prefix-reuse", action="store_true")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    if os.environ.get("BMA_DIST_BACKEND", "nccl") != "nccl":
        local = 0                                               # rehearsal: every rank on the one GPU
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BMA_DIST_BACKEND", "nccl")     # "gloo" only to rehearse ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from bimodalattack_amd import BimodalAttackConfig, native
    from bimodalattack_amd.attack import BimodalAttack, logger as gcg_logger
    from bimodalattack_amd.config import EngineOptions
    native.check_single_hip_runtime()

    wl = WORKLOADS[args.workload]
    dtype = torch.bfloat16
    model, tok, proc, messages, goal, target, image, norm = build_plugins(args.workload, device, dtype, args.layers)
    cfg_kw = dict(search_width=args.search_width, topk=256, n_replace=1, seed=1, verbosity="ERROR",
                  pgd_attack=wl["pgd_attack"], gcg_attack=wl["gcg_attack"], joint_eval=wl["joint_eval"],
                  eps=64 / 255, alpha=4 / 255)
    if wl.get("gemma"):
        cfg_kw.update(dynamic_search=True, min_search_width=128)
    total = args.warmup + args.steps
    cfg = BimodalAttackConfig(num_steps=total, images_folder=tempfile.mkdtemp(prefix="bma_bench_"), **cfg_kw)

    marks = {}

    def barrier_clock():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t = time.perf_counter()
        torch.cuda.synchronize(device)
        return t

    def hook(i: int) -> None:
        log(f"step {i}/{total}")
        if i == args.warmup:
            for k_ in attack.score_stats:
                attack.score_stats[k_] = 0
            native.profile_enable(True)          # tallies cover exactly the timed region
            marks["t0"] = barrier_clock()
        if i == total:
            marks["t1"] = barrier_clock()

    gcg_logger.setLevel("ERROR")
    attack = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(
        step_hook=hook, save_images=False, prefix_reuse=not args.no_prefix_reuse))
    log("engine constructed; running")
    res = attack.run(messages, goal, target, image)
    log("run finished")
    prof = native.profile_read()
    native.profile_enable(False)

    elapsed = marks["t1"] - marks["t0"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed = attack.n_scored[args.warmup:]
    n_cand = sum(timed)
    loss_s = sum(res.loss_times[args.warmup:])
    grad_per_step = len(res.gradient_times) // total
    grad_s = sum(res.gradient_times[args.warmup * grad_per_step:])
    samp_s = sum(res.sampling_times[args.warmup:]) if res.sampling_times else 0.0
    pgd_s = sum(res.pgd_times[args.warmup:]) if res.pgd_times else 0.0

    # ---- per-kernel rooflines from the in-library HIP events (this run, timed region) ----
    kernels = {}
    for name, p in prof.items():
        if p["launches"] == 0:
            continue
        gbs = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else None
        kernels[name] = dict(symbol=p["symbol"], launches=p["launches"], avg_us=1e3 * p["ms"] / p["launches"],
                             algorithmic_MB_per_launch=p["bytes"] / p["launches"] / 1e6,
                             achieved_GBps=gbs, frac_of_8TBps=None if gbs is None else gbs / HBM_PEAK_GBS)
    dom = max(kernels, key=lambda k: prof[k]["ms"]) if kernels else None
    pmc = None
    try:
        with open(os.path.join(REPO, "profiles", "r1_pmc_traffic.json")) as f:
            pmc = json.load(f)
    except Exception:
        pass
    roofline = None
    if dom:
        k = kernels[dom]
        traffic = None
        # HBM bytes per launch from the committed PMC passes (profiles/r1_pmc_traffic.json):
        # the entry for this kernel whose launch shape has the same algorithmic bytes
        # (ragged scoring computes a slightly different row count every step: the PMC pass of the nearest
        # shape, within 6 %, scaled by its measured traffic-to-algorithmic ratio)
        live = k["algorithmic_MB_per_launch"] * 1e6
        near = [e for e in (pmc or {}).get("entries", []) if e["kernel"] == dom
                and abs(e["algorithmic_bytes"] - live) <= 0.06 * e["algorithmic_bytes"]]
        if near:
            e = min(near, key=lambda e: abs(e["algorithmic_bytes"] - live))
            traffic = e["ratio_to_algorithmic"] * live
        roofline = dict(bound="hbm", kernel=k["symbol"], achieved=k["achieved_GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=k["frac_of_8TBps"], traffic=traffic,
                        algorithmic_bytes_per_launch=k["algorithmic_MB_per_launch"] * 1e6,
                        avg_launch_us=k["avg_us"], launches=k["launches"],
                        note="dominant hand-written kernel by summed device time inside the timed region; achieved = "
                             "algorithmic bytes / HIP-event time on the launch stream (bma_profile_*); traffic = FETCH_SIZE*2 + "
                             "WRITE_SIZE from separate rocprofv3 --pmc passes of the nearest launch shape, scaled by its traffic / "
                             "algorithmic ratio to this run's average launch (profiles/r1_pmc_traffic.json)")

    # ---- the candidate forward: MFMA-bound, algorithmic FLOPs with prefix reuse ----------
    tc = model.config.text_config
    p_layer = 4 * tc.hidden_size ** 2 + 3 * tc.hidden_size * tc.intermediate_size
    p_lm = tc.num_hidden_layers * p_layer
    seg = SEG["gcg" if args.workload == "gcg" else "joint"]
    if wl.get("gemma"):   # everything behind before_img is per candidate in the Gemma layout
        seg = dict(before_img=20, optim=19, before_suffix=3, n_img=256, after=6, target=20)
    new_tok = (sum(seg.values()) - seg["before_img"] - 1) if wl.get("gemma") else (seg["optim"] + seg["after"] + seg["target"] - 1)
    full_tok = sum(v for k_, v in seg.items())
    # rows the scoring forwards needed behind the shared prefix, per candidate: with ragged scoring
    # a candidate's suffix tokens in front of its first replaced position are not recomputed
    ss = attack.score_stats
    need_tok = ss["rows_needed"] / ss["candidates"] if ss["candidates"] else float(new_tok)
    done_tok = ss["rows"] / ss["candidates"] if ss["candidates"] else float(new_tok)
    flops_cand = 2 * p_lm * need_tok + 2 * tc.hidden_size * tc.vocab_size * seg["target"]
    fwd = None
    if loss_s > 0 and n_cand:
        ach = flops_cand * n_cand / loss_s / 1e12
        fwd = dict(bound="mfma", achieved=ach, peak=MFMA_PEAK_TFLOPS * world, unit="TFLOP/s",
                   frac=ach / (MFMA_PEAK_TFLOPS * world),
                   algorithmic_flops_per_candidate=flops_cand, rows_needed_per_candidate=need_tok,
                   rows_computed_per_candidate=done_tok, new_tokens_per_candidate=new_tok,
                   full_recompute_tokens_per_candidate=full_tok,
                   ragged_calls=ss["ragged_calls"], padded_calls=ss["padded_calls"],
                   note="scoring phase (splice + forward + CE + all-gather + argmin) over the timed steps, this "
                        "rank's candidates; flops count the rows the ragged forward NEEDS (tokens from the first "
                        "replaced suffix position on), not the padded block; GEMMs are hipBLASLt/rocBLAS inside "
                        "the HuggingFace model")

    out = {
        "metric": "candidate_forwards_per_sec", "value": n_cand / elapsed, "unit": "candidate_forwards/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": wl["name"], "search_width": args.search_width, "topk": 256, "n_optim": 19,
                   "target_tokens": seg["target"], "seq_len": full_tok, "candidates_per_step_after_filter":
                   n_cand / max(1, len(timed)), "sharding": f"candidates/{world}" if world > 1 else "none",
                   "text_layers": tc.num_hidden_layers, "prefix_reuse": not args.no_prefix_reuse},
        "attack_steps_per_sec": args.steps / elapsed,
        "scoring_phase_candidate_forwards_per_sec": n_cand / loss_s if loss_s else None,
        "phase_s_per_step": {"gradient": grad_s / args.steps, "pgd": pgd_s / args.steps,
                             "sampling_incl_filter": samp_s / args.steps, "scoring": loss_s / args.steps},
        "roofline": roofline, "kernels": kernels, "forward_roofline": fwd,
        "final_loss": res.losses[-1],
        "gradient_pass_ms_each": [round(1e3 * t, 2) for t in res.gradient_times],
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            out["cpu_baseline"] = cpu_baseline(args, model, tok, proc, messages, goal, target, image, norm, cfg_kw)
        except Exception as e:  # the GPU numbers stand on their own
            out["cpu_baseline"] = {"value": None, "unit": "candidate_forwards/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
    else:
        out["cpu_baseline"] = None
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
