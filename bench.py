#!/usr/bin/env python3
"""bench.py -- the attack-step benchmark of BASELINE.json on synthetic random-weight
LLaVA-1.5-7B (bf16), search_width = 512.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload gcg|joint|pgd|pgd_gcg|gemma_joint|opt125m]
                    [--extra-workloads joint,pgd,gemma_joint | none]

A "step" is one full pass of the hot path: gradient pass -> (PGD projection) ->
mask/top-k/scatter sampling -> retokenisation filter -> candidate splice + forward +
target cross-entropy -> argmin + bookkeeping.  Inputs (weights, embeddings, prompt
segments, image) are resident in HBM before the timed region.  With N > 1 (one rank per
GPU over RCCL: under torch.distributed.run, or started by this script itself when it is
called bare with --gpus N) the candidates of every step are sharded across ranks; the work
per step is fixed, so scaling is "strong".  The K timed steps carry no instrumentation; the
per-kernel HIP-event brackets behind `roofline`, `gemms` and `kernels` run in
--profile-steps extra steps AFTER the timed region.

Rank 0 prints ONE JSON line (see DESIGN.md "Measurement"), at most LINE_LIMIT (6000) characters of strict JSON:
the contract's fields, `roofline`, `forward_roofline`, `cpu_baseline`, `finite` and one short entry per extra
workload (`build_line`).  Everything else -- per-GEMM tables, per-kernel figures, engine state, the notes that say
how each number was taken -- goes to the detail file (--detail, default gpurun_out/bench_detail.json).  `value` is
whole-step candidate forwards per second (all ranks, wall clock, max over ranks); the per-phase rate the reference's
tables quote (search_width / loss-phase seconds) is given beside it.  A workload whose timed steps hold a non-finite
loss is marked `"finite": false`; when it is the headline workload the process exits with status 3 after printing.

The line's `value` / `config` are the --workload run (default: BASELINE configs[2], GCG-only).  On one GPU the
other single-GPU BASELINE configurations -- joint (configs[3]) and PGD-only (configs[1]) on the SAME LLaVA model,
Gemma-3 joint with the dynamic width schedule (configs[4]) on its own model -- run after it, --extra-steps timed
steps each, and land under `workloads`: {name: {ms_per_step, value, phase_s_per_step, roofline, gradient_pass,
engine}}.  Every workload's batch-1 gradient pass (replayed from hipGraphs in the timed steps, where neither Python
hooks nor the in-library event brackets can see it) is profiled once more EAGERLY after the timed region: every
library GEMM it issues is recorded (aten.mm / addmm / bmm, forward and backward) and each distinct product is timed
back to back over the layers' own weights with HIP events -- `gradient_pass.gemms`, with FLOPs, bytes, us and both
roofline fractions; for PGD-only its dominant product IS the step's `roofline`.
"""

from __future__ import annotations

import argparse
import json
import math
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
# multi-process GPU work on this driver stack needs dmabuf IPC (RCCL fails with hipIpcGetMemHandle otherwise); the
# launcher usually exports it already
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

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
    # BASELINE configs[0] / BASELINE.md 3: the plumbing case, timed IN FULL on the host cores beside the GPU run
    "opt125m": dict(pgd_attack=False, gcg_attack=True, joint_eval=False, fp32=True, search_width=16, topk=256, cpu_full=True,
                    name="GCG-only, OPT-125M-shaped fp32, search_width=16, 10 steps (BASELINE configs[0])"),
}


def wl_segments(workload: str, attack) -> dict:
    """Token counts of one candidate sequence as the engine split it: the shared prefix (what sits in front of
    the suffix -- under causal attention identical in every candidate), everything else, the target."""
    seg = {k: int(v.shape[1]) for k, v in attack.seg.items() if k != "target_in"}
    cfg = attack.config
    from bimodalattack_amd.layout import segment_order, split_at_suffix
    mt = attack.hf.model_type
    if cfg.pgd_attack and cfg.joint_eval:
        order = segment_order("pgd", mt, single=True)
    elif cfg.pgd_attack and cfg.gcg_attack:
        order = segment_order("gcg", mt, single=True)
    elif cfg.pgd_attack:
        order = segment_order("gcg_pgd", mt)
    else:
        order = segment_order("gcg", mt, no_joint_eval=True)
    n_img = {"llava": 576, "gemma3": 256}.get(mt, 0)
    n_opt = 19
    length = lambda name: n_opt if name == "optim" else (n_img if name == "image" else seg[name])   # noqa: E731
    prefix, tail = split_at_suffix(order)
    return {"shared_prefix": sum(length(n) for n in prefix), "per_candidate": sum(length(n) for n in tail if n != "target"),
            "target": seg["target"]}


_MODELS: dict = {}


def build_model(kind: str, device, dtype, layers: int, share: bool = False):
    """Random-weight model of a public config shape ("llava", "gemma", "opt"); with `share` built once per process
    and kind (bench.py's workloads on one model shape run on ONE model object)."""
    from bimodalattack_amd import synthetic as S
    key = (kind, str(device), dtype, layers)
    if not share or key not in _MODELS:
        if kind == "gemma":
            log("building Gemma-3-4b-shaped model on the device")
            model = S.gemma3_4b_shaped(dtype=dtype, device=device, seed=0, text_layers=34 if layers == 32 else layers)
        elif kind == "opt":
            model = S.opt_125m_shaped(50272, dtype=dtype, device=device, seed=0)
        else:
            log("building LLaVA-1.5-7B-shaped model on the device")
            model = S.llava_15_7b_shaped(dtype=dtype, device=device, seed=0, text_layers=layers)
        log(f"model ready on {model.device} ({sum(p.numel() for p in model.parameters()) / 1e9:.2f} B parameters)")
        if not share:
            return model
        _MODELS[key] = model
    return _MODELS[key]


def build_plugins(workload: str, device, dtype, layers: int, share: bool = False):
    """Synthetic tokenizer (32000 printable-ASCII words; embedding table has 32064 rows),
    LLaVA-1.5-7B-shaped random-weight model, prompt strings of the fixed segment lengths.  Workloads on one model
    shape share the model object (and the engine's derived weight copies with it); tokenizer, processor and
    prompt are built per workload (the PGD workloads write their own chat template)."""
    from bimodalattack_amd import synthetic as S
    if workload == "gemma_joint":
        # Gemma-3 layout 20|19|3|256|6|20: suffix in FRONT of the image
        tok = S.build_tokenizer(262144, 0, 0)
        # the real Gemma tokenizer spends 3 tokens between the suffix and the image and 6 behind it; the word-level
        # stand-in would spend 2 and 2, so filler words keep SURVEY.md 8's segment lengths (S = 324)
        f1, _ = S.synthetic_prompt(tok, 1, 1, seed=3)
        f4, _ = S.synthetic_prompt(tok, 4, 1, seed=4)
        tpl = ("{{ bos_token }}<start_of_turn>user\n"
               "{% for item in messages[0]['content'] %}"
               "{% if item['type'] == 'text' %}{{ item['text'] }}{% elif item['type'] == 'image' %} " + f1 + " <start_of_image>{% endif %}"
               "{% endfor %} <end_of_turn> " + f4 + " <start_of_turn>model\n")
        tok.chat_template = tpl
        proc = S.Gemma3Processor(tok, tpl)
        model = build_model("gemma", device, dtype, layers, share)
        goal, target = S.synthetic_prompt(tok, 18, 20, seed=0)        # <start_of_turn>user + 18 + BOS = 20
        image = S.synthetic_image(896, 896, seed=0, device=device)
        return model, tok, proc, goal, goal, target, image, S.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    if workload == "opt125m":
        # OPT-125M shape (768 / 12 layers / 12 heads / 3072), fp32, vocabulary = the tokenizer's 50272 words
        tok = S.build_tokenizer(50272, 0, 0)
        proc = S.SyntheticProcessor(tok)
        model = build_model("opt", device, dtype, layers, share)
        goal, target = S.synthetic_prompt(tok, SEG["gcg"]["before"] - 1, SEG["gcg"]["target"], seed=0)
        after, _ = S.synthetic_prompt(tok, SEG["gcg"]["after"], 1, seed=1)
        return model, tok, proc, f"{goal} {{optim_str}} {after}", goal, target, None, None
    tok = S.build_tokenizer(32000, 0, 0)
    proc = S.SyntheticProcessor(tok)
    model = build_model("llava", device, dtype, layers, share)
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


def cpu_baseline(args, wl, model, tok, proc, messages, goal, target, image, norm, cfg_kw):
    """The oracle (CPU restatement of the reference loop, pinned by the reference's goldens) timed on this box's
    host cores.  7B / 4B shapes: a bounded sample of the same workload (same model shape and prompt,
    --cpu-steps steps at search_width --cpu-width).  BASELINE configs[0] (`cpu_full`): the whole
    configuration, 10 steps at search_width 16, as BASELINE.md 3 promises."""
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
    cpu_dtype = cmodel.dtype
    rate = None
    if cpu_dtype != torch.float32:
        # bf16 GEMMs are slow on hosts without AVX512-BF16/AMX: probe, and fall back to fp32
        a = torch.randn(512, 4096).to(cmodel.dtype)
        b = torch.randn(4096, 4096).to(cmodel.dtype)
        (a @ b)
        tp = time.perf_counter()
        (a @ b)
        rate = 2 * 512 * 4096 * 4096 / (time.perf_counter() - tp)
        if rate < 2e11:
            cmodel = cmodel.float()
            cpu_dtype = torch.float32
    t_copy = time.perf_counter() - t0
    full = bool(wl.get("cpu_full"))
    steps, width = (args.warmup + args.steps, cfg_kw["search_width"]) if full else (args.cpu_steps, args.cpu_width)
    log(f"cpu baseline: running the oracle loop in {cpu_dtype}, {steps} steps at search_width {width}"
        + (f" (host GEMM probe {rate / 1e12:.2f} TFLOP/s)" if rate else ""))
    kw = dict(cfg_kw, num_steps=steps, search_width=width, images_folder=tempfile.mkdtemp(prefix="bma_cpu_"))
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
    what = (f"the WHOLE configuration: {steps} steps at search_width={width}" if full else
            f"{steps} step(s) at search_width={width} (reduced from {cfg_kw['search_width']})")
    return {
        "value": n / wall, "unit": "candidate_forwards/s", "cores": threads, "kind": "port",
        "sample": f"{what}, same model shape, prompt, host dtype {str(cpu_dtype).replace('torch.', '')}, incl. init-buffer "
                  f"scoring; {n} candidates in {wall:.1f} s",
        "attack_steps_per_s": steps / wall, "scoring_phase_cand_per_s": n / loss_s if loss_s else None,
        "gradient_pass_s": sum(res["gradient_times"]) / max(1, len(res["gradient_times"])),
        "final_loss": res["losses"][-1],
        "cpu_model": cpu_name, "host_cores_total": cores, "copy_to_host_s": round(t_copy, 2),
    }


class GemmTimer:
    """HIP-event brackets around the decoder's linear layers during the profiled steps (never inside the
    timed region): the library GEMMs are launched on torch's current stream, which is where the events are
    recorded.  Only calls with at least `min_rows` rows are tallied -- the candidate forward, not the batch-1
    passes (those are replayed from hipGraphs and run no Python anyway)."""

    ROLES = ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj", "lm_head")
    _BY_MODEL: dict = {}

    @classmethod
    def for_model(cls, model, fused_qkv_blocks):
        """One timer (one set of module hooks) per model object: the workloads of a run share the model."""
        t = cls._BY_MODEL.get(id(model))
        if t is None:
            t = cls._BY_MODEL[id(model)] = cls(model, fused_qkv_blocks)
        t.on, t.rec = False, []
        return t

    def __init__(self, model, fused_qkv_blocks, min_rows: int = 1024):
        import torch
        self.torch, self.on, self.min_rows, self.rec = torch, False, min_rows, []
        fused_q = {id(b.q_proj): sum(m.out_features for m in (b.q_proj, b.k_proj, b.v_proj)) for b in fused_qkv_blocks}
        skip = {id(m) for b in fused_qkv_blocks for m in (b.k_proj, b.v_proj)}
        for name, mod in model.named_modules():
            role = name.rsplit(".", 1)[-1]
            if not isinstance(mod, torch.nn.Linear) or role not in self.ROLES or "vision" in name or "multi_modal" in name:
                continue
            if id(mod) in skip:
                continue            # their output is a slice of the block's one fused q/k/v product
            n_out = fused_q.get(id(mod), mod.out_features)
            # gate_proj and up_proj are the same product shape served by the same library kernel: one line, as in
            # a kernel trace folded by (symbol, grid)
            label = "qkv_proj" if id(mod) in fused_q else ("gate_up_proj" if role in ("gate_proj", "up_proj") else role)
            mod.register_forward_pre_hook(self._pre)
            mod.register_forward_hook(self._post(label, n_out, mod.in_features))

    # products issued outside an nn.Linear (the fused gate/up product, fused.py) are bracketed through these
    def begin(self, x):
        if x.numel() // x.shape[-1] < self.min_rows:
            return None
        e = self.torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, e0, label, x, n_out, k_in):
        if e0 is not None:
            e1 = self.torch.cuda.Event(enable_timing=True)
            e1.record()
            self.rec.append((label, x.numel() // x.shape[-1], n_out, k_in, e0, e1))

    def _pre(self, mod, args):
        if self.on and args and args[0].numel() // args[0].shape[-1] >= self.min_rows:
            e = self.torch.cuda.Event(enable_timing=True)
            e.record()
            mod._bma_t0 = e

    def _post(self, label, n_out, k_in):
        def hook(mod, args, out):
            e0 = mod.__dict__.pop("_bma_t0", None)
            if e0 is not None:
                e1 = self.torch.cuda.Event(enable_timing=True)
                e1.record()
                self.rec.append((label, args[0].numel() // args[0].shape[-1], n_out, k_in, e0, e1))
        return hook

    def table(self) -> dict:
        """{"role MxNxK": dict(launches, avg_us, flops_per_launch, achieved_TFLOPs, frac_of_peak, total_ms)}"""
        out = {}
        for label, M, N, K, e0, e1 in self.rec:
            d = out.setdefault(f"{label} M={M} N={N} K={K}", dict(role=label, M=M, N=N, K=K, launches=0, total_ms=0.0))
            d["launches"] += 1
            d["total_ms"] += e0.elapsed_time(e1)
        for d in out.values():
            d["avg_us"] = 1e3 * d["total_ms"] / d["launches"]
            d["flops_per_launch"] = 2.0 * d["M"] * d["N"] * d["K"]
            d["achieved_TFLOPs"] = d["flops_per_launch"] / (d["avg_us"] * 1e-6) / 1e12
            d["frac_of_peak"] = d["achieved_TFLOPs"] / MFMA_PEAK_TFLOPS
        return out


class GradPassProfile:
    """The GEMMs of ONE batch-1 gradient pass (library products, and those on the engine's own bma_gemm_nt), measured
    outside the timed region.

    The timed steps replay the pass from hipGraphs: no Python runs, so neither module hooks nor the in-library event
    brackets see it (and GemmTimer deliberately ignores products under 1024 rows).  Here the same work runs once
    EAGERLY under a dispatch mode that records every aten.mm / addmm / bmm -- forward and backward, the autograd
    engine's thread included -- with its operand tensors; then each distinct product (op, shapes, strides) is
    timed: its recorded calls, which walk the layers' own weights, so every launch streams a different weight from
    HBM as in the real pass, are captured into one hipGraph and replayed between two HIP events on the stream they
    run on.  Per product: FLOPs 2MNK, operand bytes, us, TFLOP/s against the dense MFMA peak, GB/s against HBM."""

    OPS = ("mm", "addmm", "bmm")

    def __init__(self, torch):
        self.torch = torch
        self.calls = []

    def record(self, fn):
        torch = self.torch
        from torch.utils._python_dispatch import TorchDispatchMode
        outer = self

        class Rec(TorchDispatchMode):
            def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                name = getattr(getattr(func, "overloadpacket", None), "__name__", "")
                if name in outer.OPS and all(torch.is_tensor(a) and a.is_cuda for a in args[-2:]):
                    outer.calls.append((name, func, args, dict(kwargs or {})))
                return func(*args, **(kwargs or {}))

        from bimodalattack_amd import ops

        def nt_hook(x, w):          # products the engine routes to its own skinny kernel never reach aten
            outer.calls.append(("gemm_nt", ops.gemm_nt, (x, w), {}))

        def mid_hook(x, w):
            outer.calls.append(("gemm_mid", ops.gemm_mid, (x, w), {}))

        ops.GEMM_NT_HOOK = nt_hook
        ops.GEMM_MID_HOOK = mid_hook
        try:
            with Rec():
                out = fn()
        finally:
            ops.GEMM_NT_HOOK = None
            ops.GEMM_MID_HOOK = None
        torch.cuda.synchronize()
        return out

    @staticmethod
    def _mnk(name, args):
        a, b = args[-2], args[-1]
        if name == "bmm":
            return a.shape[0], a.shape[1], b.shape[2], a.shape[2]
        if name in ("gemm_nt", "gemm_mid"):         # (x (..., K), w (N, K))
            return 1, a.numel() // a.shape[-1], b.shape[0], b.shape[1]
        return 1, a.shape[0], b.shape[1], a.shape[1]

    def table(self, roles: dict, max_launches: int = 96) -> list:
        """One entry per distinct product, in order of summed time."""
        torch = self.torch
        groups = {}
        for name, func, args, kw in self.calls:
            a, b = args[-2], args[-1]
            key = (name, tuple(a.shape), tuple(b.shape), tuple(a.stride()), tuple(b.stride()), str(a.dtype))
            groups.setdefault(key, []).append((func, args, kw))
        out = []
        for key, calls in groups.items():
            name = key[0]
            batch, M, N, K = self._mnk(name, calls[0][1])
            run = calls[:max_launches]
            for func, args, kw in run[:2]:
                func(*args, **kw)
            torch.cuda.synchronize()
            how = "hipGraph of the recorded calls"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            try:
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    keep = [func(*args, **kw) for func, args, kw in run]
                g.replay()
                torch.cuda.synchronize()
                e0.record()
                g.replay()
                e1.record()
                torch.cuda.synchronize()
                del keep, g
            except Exception:
                torch.cuda.synchronize()
                how = "eager loop over the recorded calls"
                e0.record()
                for func, args, kw in run:
                    func(*args, **kw)
                e1.record()
                torch.cuda.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / len(run)
            es = calls[0][1][-1].element_size()
            flops = 2.0 * batch * M * N * K
            nbytes = float(es * batch * (M * K + K * N + M * N))
            b_shape = key[2]
            # a linear layer's weight is the (K,N) operand, a transposed view of the stored (N,K) matrix -- or, for the
            # input gradients through the transposed copies, of the stored (K,N) one
            role = roles.get((N, K)) if name != "bmm" else None
            if role and name == "gemm_nt":
                role += " [bma_gemm_nt]"
            out.append(dict(op=name, role=role or f"{name} {tuple(key[1])} x {tuple(b_shape)}", batch=batch, M=M, N=N, K=K,
                            dtype=key[5].replace("torch.", ""), launches_per_pass=len(calls), avg_us=us,
                            total_ms_per_pass=us * len(calls) / 1e3, flops_per_launch=flops, bytes_per_launch=nbytes,
                            achieved_TFLOPs=flops / (us * 1e-6) / 1e12, frac_of_mfma_peak=flops / (us * 1e-6) / 1e12 / MFMA_PEAK_TFLOPS,
                            achieved_GBps=nbytes / (us * 1e-6) / 1e9, frac_of_8TBps=nbytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            timed_as=how))
        out.sort(key=lambda d: -d["total_ms_per_pass"])
        return out


def gemm_roles(tc, vocab_rows: int) -> dict:
    """(N, K) of a product -> which decoder projection it is (forward: y = x W^T; `dX`: the input gradient)."""
    D = tc.hidden_size
    inter = getattr(tc, "intermediate_size", None) or getattr(tc, "ffn_dim")
    heads = tc.num_attention_heads
    hd = getattr(tc, "head_dim", None) or D // heads
    kvh = getattr(tc, "num_key_value_heads", None) or heads
    q, kv = heads * hd, kvh * hd
    r = {}

    def put(n, k, name):
        r.setdefault((n, k), name)
        r.setdefault((k, n), name + " dX")

    put(2 * inter, D, "gate_up_proj (fused)")
    put(q + 2 * kv, D, "qkv_proj (fused)")
    put(D, inter, "down_proj")
    put(D, q, "o_proj")
    put(inter, D, "gate_proj / up_proj")
    put(q, D, "q_proj")
    put(kv, D, "k_proj / v_proj")
    put(vocab_rows, D, "lm_head / token scores")
    return r


LINE_LIMIT = 6000          # the driver keeps the last 8000 characters of stdout: the ONE line must fit with room to spare


def _strict(o):
    """JSON without NaN / Infinity (not JSON: strict parsers refuse the whole line): non-finite floats become null."""
    if isinstance(o, float):
        return o if math.isfinite(o) else None
    if isinstance(o, dict):
        return {str(k): _strict(v) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_strict(v) for v in o]
    return o


def _sig(o, digits: int = 5):
    """Floats to `digits` significant figures (the line is a summary; the detail file keeps every bit)."""
    if isinstance(o, float):
        return float(f"{o:.{digits}g}") if math.isfinite(o) else None
    if isinstance(o, dict):
        return {k: _sig(v, digits) for k, v in o.items()}
    if isinstance(o, (list, tuple)):
        return [_sig(v, digits) for v in o]
    return o


def _short_roofline(r, keys=("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "traffic_source",
                             "avg_launch_us", "launches")):
    if not r:
        return None
    d = {k: r[k] for k in keys if k in r}
    if isinstance(d.get("kernel"), str):
        d["kernel"] = d["kernel"][:120]
    if isinstance(d.get("traffic_source"), str):
        d["traffic_source"] = d["traffic_source"].split(" (")[0][:80] + " (committed rocprofv3 --pmc pass)"
    return d


def build_line(out: dict, detail_file=None) -> dict:
    """The ONE stdout line from a full result: the contract's fields, `roofline` / `forward_roofline` / `cpu_baseline`,
    one short entry per extra workload -- no prose, no tables (those live in the detail file).  Stays under LINE_LIMIT
    characters: optional blocks are dropped, least important first, if it would not."""
    head = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config")
    line = {k: out.get(k) for k in head}
    line["finite"] = bool(out.get("finite", False))
    line["final_loss"] = out.get("final_loss")
    for k in ("attack_steps_per_sec", "scoring_phase_candidate_forwards_per_sec", "phase_s_per_step", "projected_value"):
        if out.get(k) is not None:
            line[k] = out[k]
    line["roofline"] = _short_roofline(out.get("roofline"))
    line["forward_roofline"] = _short_roofline(out.get("forward_roofline"), ("bound", "achieved", "peak", "unit", "frac"))
    gp = out.get("gradient_pass") or {}
    if "replayed_pass_ms" in gp:
        line["gradient_pass"] = {k: gp.get(k) for k in ("replayed_pass_ms", "passes_per_step", "gemm_ms_per_pass",
                                                          "gemms_frac_of_8TBps", "gemms_frac_of_mfma_peak")}
    if out.get("kernels"):
        line["kernels"] = {n: {"avg_us": k.get("avg_us"), "frac_of_8TBps": k.get("frac_of_8TBps_hbm", k.get("frac_of_8TBps"))}
                           for n, k in out["kernels"].items()}
    cb = out.get("cpu_baseline")
    if cb:
        cb = dict(cb)
        if isinstance(cb.get("sample"), str):
            cb["sample"] = cb["sample"][:200]
        line["cpu_baseline"] = cb
    else:
        line["cpu_baseline"] = None
    if out.get("workloads"):
        ws = {}
        for name, w in out["workloads"].items():
            if "error" in w:
                ws[name] = {"error": str(w["error"])[:160], "finite": False}
                continue
            r = w.get("roofline") or {}
            ws[name] = {"ms_per_step": w.get("ms_per_step"), "value": w.get("value"), "unit": w.get("unit"),
                        "steps": w.get("steps"), "final_loss": w.get("final_loss"), "finite": bool(w.get("finite", False)),
                        "roofline": {"kernel": str(r.get("kernel"))[:100], "frac": r.get("frac"), "bound": r.get("bound")},
                        "forward_frac": (w.get("forward_roofline") or {}).get("frac"),
                        "gradient_pass_ms": (w.get("gradient_pass") or {}).get("replayed_pass_ms")}
        line["workloads"] = ws
    if out.get("rccl"):
        rc = out["rccl"]
        line["rccl"] = {k: rc.get(k) for k in ("backend", "world", "rccl_version", "collectives_per_step",
                                               "allgather_bytes_per_step", "state_broadcast_bytes_per_step",
                                               "tp_off_ms", "tp_on_ms", "chosen", "tp_graph", "tp_fallbacks", "tp_error", "tp_note")
                        if rc.get(k) is not None or k in ("tp_off_ms", "tp_on_ms", "chosen")}
        if isinstance(line["rccl"].get("tp_note"), str):
            line["rccl"]["tp_note"] = line["rccl"]["tp_note"][:160]
        if isinstance(line["rccl"].get("tp_fallbacks"), dict):
            line["rccl"]["tp_fallbacks"] = {k: str(v)[:100] for k, v in line["rccl"]["tp_fallbacks"].items()}
        line["rccl"]["devices"] = len(rc.get("device_names") or [])
    if out.get("scaling_table"):
        line["scaling_table"] = out["scaling_table"]
    eng = out.get("engine") or {}
    line["engine"] = {"fallbacks": eng.get("fallbacks"), "graphs": len(eng.get("graphs_captured") or []),
                      "tuned_gemms": eng.get("tuned_gemms"), "collectives": eng.get("collectives")}
    line["detail_file"] = detail_file
    line = _strict(_sig(line))
    for drop in ("kernels", "engine", "gradient_pass", "phase_s_per_step", "rccl", "workloads"):
        if len(json.dumps(line)) <= LINE_LIMIT:
            break
        line.pop(drop, None)
    return line


def self_launch(args) -> None:
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks as CHILD processes
    (python -m torch.distributed.run, rendezvous on 127.0.0.1) before this process has touched the GPU, pass
    their output through and exit with their code.  Nothing is exec'ed."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "8")
    log(f"--gpus {args.gpus} without a launcher: starting {args.gpus} ranks with torch.distributed.run on port {port}")
    sys.exit(subprocess.run(cmd, env=env).returncode)


_PMC = {}


def committed_profile(name: str):
    """A committed measurement file under profiles/ (the newest round that has one), parsed; (None, None) when absent."""
    if name not in _PMC:
        hit = (None, None)
        for tag in ("r6", "r5", "r4"):
            path = os.path.join(REPO, "profiles", f"{tag}_{name}")
            try:
                with open(path) as f:
                    hit = (json.load(f), f"profiles/{tag}_{name}")
                break
            except Exception:
                continue
        _PMC[name] = hit
    return _PMC[name]


def pmc_traffic(kernel: str, live_bytes: float):
    """(HBM bytes per launch, source) from the COMMITTED rocprofv3 --pmc passes (FETCH_SIZE x2 + WRITE_SIZE over
    tools/kernel_bench.py -- not measured in this run): the entry of this kernel whose launch shape is within 6 % of
    this run's algorithmic bytes, scaled by its measured traffic / algorithmic ratio.  (None, None) without one."""
    pmc, src = committed_profile("pmc_traffic.json")
    near = [e for e in (pmc or {}).get("entries", []) if e["kernel"] == kernel and e.get("valid", e["ratio_to_algorithmic"] >= 0.97)
            and abs(e["algorithmic_bytes"] - live_bytes) <= 0.06 * e["algorithmic_bytes"]]
    if not near:
        return None, None
    e = min(near, key=lambda e: abs(e["algorithmic_bytes"] - live_bytes))
    at = (pmc or {}).get("generated_at_commit")
    return e["ratio_to_algorithmic"] * live_bytes, f"{src}{'@' + at if at else ''} ({e['pmc_key']}), committed file, not measured in this run"


COPY_CEILING_GBS = 6290.0     # measured streaming-copy ceiling of the part (MI355X_MICROARCH.md)


def measure(args, workload: str, steps: int, warmup: int, n_prof: int, device, world: int, rank: int, primary: bool,
            engine_kw=None):
    """One workload: construct the engine on the (shared) model, run warmup + steps + n_prof attack steps with the clock
    around the `steps`, then profile.  Returns (result dict, what cpu_baseline needs).  `engine_kw`: engine options of
    this leg (the multi-GPU A/B of the tensor-parallel gradient pass)."""
    import torch
    import torch.distributed as dist
    from bimodalattack_amd import BimodalAttackConfig, native
    from bimodalattack_amd.attack import BimodalAttack, logger as gcg_logger
    from bimodalattack_amd.config import EngineOptions

    wl = WORKLOADS[workload]
    dtype = torch.float32 if wl.get("fp32") else torch.bfloat16
    sw = args.search_width if (args.search_width is not None and primary) else wl.get("search_width", 512)
    log(f"=== workload {workload}: {wl['name']}")
    model, tok, proc, messages, goal, target, image, norm = build_plugins(workload, device, dtype, args.layers, share=True)
    cfg_kw = dict(search_width=sw, topk=wl.get("topk", 256), n_replace=1, seed=1, verbosity="ERROR",
                  pgd_attack=wl["pgd_attack"], gcg_attack=wl["gcg_attack"], joint_eval=wl["joint_eval"],
                  eps=64 / 255, alpha=4 / 255)
    timed_end = warmup + steps
    # The engine pipelines across the step boundary (gradient_ahead: step i+1's gradient pass is queued during step i;
    # fuse_pgd_only likewise), so the timed region [hook(warmup), hook(timed_end)) holds K gradient passes and K scoring
    # passes only if a step follows it: at least one untimed step runs behind the region.
    total = timed_end + max(n_prof, 1)
    width_of = None
    if wl.get("gemma"):
        # BASELINE configs[4]: dynamic_search 512 -> 128 over 600 steps (reference :919-923).  K timed steps sample
        # that schedule at evenly spaced points (mean width 271 at K = 5; the 600-step mean is 272); warm-up runs
        # the widest step, the profiled steps the middle one.
        from bimodalattack_amd.layout import dynamic_width
        cfg_kw.update(dynamic_search=True, min_search_width=128)
        SCHED = 600

        def width_of(i: int) -> int:
            if i < warmup:
                v = 0
            elif i < timed_end:
                v = int(round((i - warmup + 0.5) * SCHED / steps))
            else:
                v = SCHED // 2
            return dynamic_width(min(v, SCHED - 1), sw, SCHED, 128, True)
    cfg = BimodalAttackConfig(num_steps=total, images_folder=tempfile.mkdtemp(prefix="bma_bench_"), **cfg_kw)

    marks = {}

    def barrier_clock():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(device)
        t = time.perf_counter()
        torch.cuda.synchronize(device)
        return t

    def marker() -> None:
        """A float64 reduction no step launches: tools/trace_by_grid.py --between-markers keeps what lies between."""
        torch.zeros(12345, device=device, dtype=torch.float64).sum()
        torch.cuda.synchronize(device)

    def hook(i: int) -> None:
        log(f"{workload}: step {i}/{total}" + (" (profiled, outside the timed region)" if timed_end <= i < total else ""))
        if i == warmup:
            for k_ in attack.score_stats:
                attack.score_stats[k_] = 0
            if primary:
                marker()                         # outside the clock: a kernel trace can be cut to the timed steps
            marks["t0"] = barrier_clock()
        if i == timed_end:
            marks["t1"] = barrier_clock()
            if primary:
                marker()
            marks["stats"] = dict(attack.score_stats)
            marks["ids"] = attack._last["sampled"][:1].clone() if getattr(attack, "_last", None) else None
            if n_prof:
                native.profile_enable(True)      # event brackets cost a few us per launch: kept OUT of the timing (resets the tallies)
                gemms.on = True
        if i == total and n_prof:
            torch.cuda.synchronize(device)
            gemms.on = False

    gcg_logger.setLevel("ERROR")
    attack = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(
        step_hook=hook, save_images=False, prefix_reuse=not args.no_prefix_reuse, width_override=width_of, **(engine_kw or {})))
    gemms = GemmTimer.for_model(model, attack.fused.qkv if attack.fused.enabled else [])
    gemms.rec = []
    attack.fused.gemm_probe = gemms
    log(f"{workload}: engine constructed; running")
    res = attack.run(messages, goal, target, image)
    log(f"{workload}: run finished")
    prof = native.profile_read()
    native.profile_enable(False)

    elapsed = marks["t1"] - marks["t0"]
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    timed = attack.n_scored[warmup:timed_end]
    n_cand = sum(timed)
    emulate = attack.emulate_world if (attack.emulate_world > 1 and world == 1) else 0
    ss = marks.get("stats", attack.score_stats)
    # with BMA_EMULATE_WORLD=W this process scores rank 0's share only: `value` counts the candidates ACTUALLY scored,
    # the whole-job figure a W-GPU run would print is a projection and is labelled as one
    n_done = ss["candidates"] if (emulate and ss["candidates"]) else n_cand
    loss_s = sum(res.loss_times[warmup:timed_end])
    grad_per_step = len(res.gradient_times) // total
    grad_s = sum(res.gradient_times[warmup * grad_per_step:timed_end * grad_per_step])
    samp_s = sum(res.sampling_times[warmup:timed_end]) if res.sampling_times else 0.0
    pgd_s = sum(res.pgd_times[warmup:timed_end]) if res.pgd_times else 0.0

    # ---- per-kernel rooflines from the in-library HIP events (profiled steps after the timed region) ----
    kb, kb_src = committed_profile("kernel_bench.json")
    kernels = {}
    for name, p in prof.items():
        if p["launches"] == 0:
            continue
        gbs = p["bytes"] / (p["ms"] * 1e-3) / 1e9 if p["ms"] > 0 else None
        k = dict(symbol=p["symbol"], launches=p["launches"], avg_us=1e3 * p["ms"] / p["launches"],
                 algorithmic_MB_per_launch=p["bytes"] / p["launches"] / 1e6,
                 achieved_GBps=gbs, frac_of_8TBps=None if gbs is None else gbs / HBM_PEAK_GBS, total_ms=p["ms"])
        if gbs is not None and gbs > COPY_CEILING_GBS:
            # faster than a streaming copy can move data through HBM: part of this launch's bytes never left the
            # 256 MB Infinity Cache (its input was just written by the producer, or its output is consumed at once).
            # The HBM-honest figure is the standalone one (tools/kernel_bench.py, cold operands).
            mb = k["algorithmic_MB_per_launch"]
            alone = [v for c, v in (kb or {}).items() if isinstance(v, dict) and c.split("/")[0] == name
                     and abs(v.get("algorithmic_MB", 0) - mb) <= 0.1 * mb]
            k["cache_resident"] = True
            k["live_frac_exceeds_copy_ceiling"] = gbs / COPY_CEILING_GBS
            if alone:
                a = min(alone, key=lambda v: abs(v["algorithmic_MB"] - mb))
                k["standalone"] = dict(avg_us=a["avg_us"], achieved_GBps=a["achieved_GBps"],
                                       frac_of_8TBps=a["achieved_GBps"] / HBM_PEAK_GBS, source=kb_src)
                k["frac_of_8TBps_hbm"] = a["achieved_GBps"] / HBM_PEAK_GBS
        kernels[name] = k
    gemm_table = gemms.table()

    # ---- the batch-1 gradient pass, eagerly, GEMM by GEMM (see GradPassProfile) ------------------------
    tc = getattr(model.config, "text_config", None) or model.config
    grad_profile = None
    if n_prof and not args.no_gradient_profile and (wl["pgd_attack"] or wl["gcg_attack"]):
        try:
            log(f"{workload}: profiling one eager gradient pass")
            ids = marks.get("ids")
            if ids is None:
                ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(device)
            gp = GradPassProfile(torch)
            img = attack.final_image if wl["pgd_attack"] else None
            with torch.enable_grad():
                attack.gradient_pass_eager(ids, img)            # lazy initialisations out of the recording
                gp.record(lambda: attack.gradient_pass_eager(ids, img))
            rows = gp.table(gemm_roles(tc, attack.embedding_layer.num_embeddings))
            timed_passes = res.gradient_times[warmup * grad_per_step:timed_end * grad_per_step]
            pass_ms = 1e3 * sum(timed_passes) / max(1, len(timed_passes))
            gemm_ms = sum(r["total_ms_per_pass"] for r in rows)
            flops = sum(r["flops_per_launch"] * r["launches_per_pass"] for r in rows)
            nbytes = sum(r["bytes_per_launch"] * r["launches_per_pass"] for r in rows)
            grad_profile = dict(
                replayed_pass_ms=pass_ms, passes_per_step=grad_per_step, gemm_ms_per_pass=gemm_ms,
                gemm_share_of_pass=gemm_ms / pass_ms if pass_ms else None, gemm_launches_per_pass=len(gp.calls),
                gemm_TFLOP_per_pass=flops / 1e12, gemm_GB_per_pass=nbytes / 1e9,
                gemms_achieved_TFLOPs=flops / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None,
                gemms_frac_of_mfma_peak=flops / (gemm_ms * 1e-3) / 1e12 / MFMA_PEAK_TFLOPS if gemm_ms else None,
                gemms_achieved_GBps=nbytes / (gemm_ms * 1e-3) / 1e9 if gemm_ms else None,
                gemms_frac_of_8TBps=nbytes / (gemm_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if gemm_ms else None,
                gemms=rows[:14],
                note="one gradient pass run eagerly AFTER the timed region; every aten.mm/addmm/bmm it issues (forward and "
                     "backward) recorded, each distinct product timed over its recorded calls (the layers' own weights, so "
                     "every launch streams from HBM) as one hipGraph between two HIP events; replayed_pass_ms is the "
                     "hipGraph replay of the whole pass in the timed steps")
            del gp
        except Exception as e:
            grad_profile = dict(error=f"{type(e).__name__}: {e}")
            torch.cuda.synchronize(device)

    # ---- the dominant kernel of a step BY DEVICE TIME, hand-written or library: the headline roofline ----
    es = 2 if dtype != torch.float32 else 4
    by_time = [("hip:" + k, v["total_ms"] / max(1, n_prof)) for k, v in kernels.items()] + \
              [("gemm:" + k, v["total_ms"] / max(1, n_prof)) for k, v in gemm_table.items()]
    if grad_profile and "gemms" in grad_profile:
        by_time += [(f"grad:{i}", r["total_ms_per_pass"] * grad_per_step) for i, r in enumerate(grad_profile["gemms"])]
    roofline = None
    if by_time:
        top = max(by_time, key=lambda kv: kv[1])[0]
        if top.startswith("gemm:"):
            gk = gemm_table[top[5:]]
            traffic, tsrc = pmc_traffic("gemm_" + gk["role"], es * (gk["M"] * gk["K"] + gk["N"] * gk["K"] + gk["M"] * gk["N"]))
            roofline = dict(bound="mfma", kernel=f"hipBLASLt/rocBLAS GEMM {top[5:]} (decoder {gk['role']}, {('bf16' if es == 2 else 'f32')})",
                            achieved=gk["achieved_TFLOPs"], peak=MFMA_PEAK_TFLOPS, unit="TFLOP/s", frac=gk["frac_of_peak"],
                            traffic=traffic, traffic_source=tsrc,
                            algorithmic_flops_per_launch=gk["flops_per_launch"], avg_launch_us=gk["avg_us"], launches=gk["launches"],
                            note="dominant kernel of the step by summed device time; achieved = 2*M*N*K / HIP-event time around "
                                 "the library call on torch's current stream, over the profiled steps that follow the timed "
                                 "region; the rocprofv3 kernel-trace line of the same launch shape is in "
                                 "profiles/r6_bench_*_kernel_by_grid.txt; peak = 2.5 PFLOP/s dense bf16; traffic = HBM bytes "
                                 "of this launch shape from the committed PMC passes (FETCH_SIZE x2 + WRITE_SIZE)")
        elif top.startswith("grad:"):
            r = grad_profile["gemms"][int(top[5:])]
            mfma_bound = r["frac_of_mfma_peak"] >= r["frac_of_8TBps"]
            roofline = dict(bound="mfma" if mfma_bound else "hbm",
                            kernel=(f"{'bma_' + r['op'] if r['op'] in ('gemm_nt', 'gemm_mid') else 'hipBLASLt/rocBLAS GEMM'} of the batch-1 gradient pass: "
                                    f"{r['role']}, M={r['M']} N={r['N']} K={r['K']} {r['dtype']}"),
                            achieved=r["achieved_TFLOPs"] if mfma_bound else r["achieved_GBps"],
                            peak=MFMA_PEAK_TFLOPS if mfma_bound else HBM_PEAK_GBS, unit="TFLOP/s" if mfma_bound else "GB/s",
                            frac=r["frac_of_mfma_peak"] if mfma_bound else r["frac_of_8TBps"],
                            frac_of_mfma_peak=r["frac_of_mfma_peak"], frac_of_8TBps=r["frac_of_8TBps"], traffic=None,
                            algorithmic_flops_per_launch=r["flops_per_launch"], algorithmic_bytes_per_launch=r["bytes_per_launch"],
                            avg_launch_us=r["avg_us"], launches=r["launches_per_pass"] * grad_per_step,
                            note="dominant kernel of the step by summed device time: a product of the gradient pass (which the "
                                 "timed steps replay from a hipGraph); measured by GradPassProfile after the timed region -- the "
                                 "product's recorded calls over the layers' own weights, one hipGraph between two HIP events; "
                                 "both fractions given, `bound` names the larger")
        else:
            k = kernels[top[4:]]
            live = k["algorithmic_MB_per_launch"] * 1e6
            traffic, tsrc = pmc_traffic(top[4:], live)
            roofline = dict(bound="hbm", kernel=k["symbol"], achieved=k["achieved_GBps"], peak=HBM_PEAK_GBS, unit="GB/s",
                            frac=k.get("frac_of_8TBps_hbm", k["frac_of_8TBps"]), traffic=traffic, traffic_source=tsrc,
                            algorithmic_bytes_per_launch=live,
                            avg_launch_us=k["avg_us"], launches=k["launches"],
                            note="dominant kernel of the step by summed device time; achieved = algorithmic bytes / HIP-event "
                                 "time on the launch stream (bma_profile_*) over the profiled steps that follow the timed region; "
                                 "traffic = FETCH_SIZE*2 + WRITE_SIZE from separate rocprofv3 --pmc passes (committed file)")

    # ---- the candidate forward: MFMA-bound, algorithmic FLOPs with prefix reuse ----------
    inter = getattr(tc, "intermediate_size", None) or getattr(tc, "ffn_dim")
    heads = tc.num_attention_heads
    head_dim = getattr(tc, "head_dim", None) or tc.hidden_size // heads
    kv_heads = getattr(tc, "num_key_value_heads", None) or heads
    n_mlp = 3 if hasattr(tc, "intermediate_size") else 2
    p_layer = tc.hidden_size * head_dim * (2 * heads + 2 * kv_heads) + n_mlp * tc.hidden_size * inter
    p_lm = tc.num_hidden_layers * p_layer
    seg = wl_segments(workload, attack)
    full_tok = sum(seg.values())
    new_tok = full_tok - seg["shared_prefix"] - 1
    need_tok = ss["rows_needed"] / ss["candidates"] if ss["candidates"] else float(new_tok)
    done_tok = ss["rows"] / ss["candidates"] if ss["candidates"] else float(new_tok)
    flops_cand = 2 * p_lm * need_tok + 2 * tc.hidden_size * tc.vocab_size * seg["target"]
    fwd = None
    if loss_s > 0 and n_done and wl["gcg_attack"]:
        ach = flops_cand * n_done / loss_s / 1e12
        fwd = dict(bound="mfma", achieved=ach, peak=MFMA_PEAK_TFLOPS * world, unit="TFLOP/s",
                   frac=ach / (MFMA_PEAK_TFLOPS * world),
                   algorithmic_flops_per_candidate=flops_cand, rows_needed_per_candidate=need_tok,
                   rows_computed_per_candidate=done_tok, new_tokens_per_candidate=new_tok,
                   full_recompute_tokens_per_candidate=full_tok,
                   ragged_calls=ss["ragged_calls"], padded_calls=ss["padded_calls"],
                   note="scoring phase (splice + forward + CE + all-gather + argmin) over the timed steps, the candidates "
                        "ACTUALLY scored against all ranks' peak; flops count the rows the ragged forward NEEDS (tokens from "
                        "the first replaced suffix position on), not the padded block; GEMMs are hipBLASLt/rocBLAS "
                        "inside the HuggingFace model")

    out = {
        "metric": "candidate_forwards_per_sec", "value": n_done / elapsed, "unit": "candidate_forwards/s",
        "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
        "dtype": "f32" if dtype == torch.float32 else "bf16", "data": "synthetic",
        "config": {"workload": wl["name"], "search_width": sw, "topk": cfg_kw["topk"], "n_optim": 19,
                   "target_tokens": seg["target"], "seq_len": full_tok, "candidates_per_step_after_filter":
                   n_cand / max(1, len(timed)), "sharding": f"candidates/{world}" if world > 1 else "none",
                   "text_layers": tc.num_hidden_layers, "prefix_reuse": not args.no_prefix_reuse,
                   "width_schedule": None if width_of is None else
                   {"of": "600-step dynamic_search 512->128, sampled evenly", "timed_widths": [width_of(i) for i in range(warmup, timed_end)]}},
        "attack_steps_per_sec": steps / elapsed,
        "scoring_phase_candidate_forwards_per_sec": n_done / loss_s if loss_s else None,
        "phase_s_per_step": {"gradient": grad_s / steps, "pgd": pgd_s / steps,
                             "sampling_incl_filter": samp_s / steps, "scoring": loss_s / steps},
        "roofline": roofline, "forward_roofline": fwd, "gradient_pass": grad_profile, "gemms": gemm_table, "kernels": kernels,
        "profiled_steps_after_timed_region": n_prof,
        "engine": attack.engine_state(),
        "final_loss": res.losses[timed_end - 1],
        # every loss of the timed steps, checked: a step loop that went non-finite was not the workload (and trivial
        # operands flatter the GEMMs)
        "finite": bool(all(math.isfinite(float(v)) for v in res.losses[warmup:timed_end])),
        "losses_timed": [float(v) for v in res.losses[warmup:timed_end]],
        "gradient_pass_ms_each": [round(1e3 * t, 2) for t in res.gradient_times[:timed_end * grad_per_step]],
    }
    if emulate:
        # a single process doing rank 0's share of an `emulate`-rank run (GEMM tuning aid): NOT a multi-GPU measurement
        out["config"]["emulate_world"] = emulate
        out["config"]["sharding"] = f"EMULATED: rank 0's share of candidates/{emulate}, in one process on one GPU"
        out["config"]["candidates_scored_per_step"] = n_done / max(1, len(timed))
        out["projected_value"] = n_cand / elapsed
        out["projection"] = (f"what {emulate} GPUs WOULD print as `value` if every rank took this rank's time and the "
                             "collectives cost nothing: an upper bound from one process, not a measurement")
    keep = dict(model=model, tok=tok, proc=proc, messages=messages, goal=goal, target=target, image=image, norm=norm,
                cfg_kw=cfg_kw, wl=wl, collectives=attack.shard.n_collectives, total_steps=total)
    del attack
    return out, keep


def rccl_info(torch, dist, device, world: int, keep: dict) -> dict:
    """What a multi-rank line needs to be believed: backend, world, every rank's device, the collectives of a step."""
    names = [None] * world
    mine = f"rank {dist.get_rank()}: {torch.cuda.get_device_name(device)} (cuda:{device.index}, " \
           f"{torch.cuda.get_device_properties(device).total_memory / 2**30:.0f} GiB)"
    dist.all_gather_object(names, mine)
    sw = keep["cfg_kw"]["search_width"]
    per = -(-sw // world)
    img = keep["image"]
    sync_bytes = sw * 19 * 8 + (0 if img is None else img.numel() * 4)
    return dict(backend=dist.get_backend(), world=world, device_names=names,
                rccl_version=".".join(str(v) for v in torch.cuda.nccl.version()) if dist.get_backend() == "nccl" else None,
                collectives_per_step=round((keep["collectives"] - 1) / max(1, keep["total_steps"]), 2),
                allgather_bytes_per_step=4 * per * world, allgather_bytes_per_rank=4 * per,
                state_broadcast_bytes_per_step=sync_bytes,
                what="per step: one broadcast of rank 0's packed sampled ids (+ PGD image), one all_gather_into_tensor of "
                     "ceil(N/W) fp32 losses per rank (+inf padded); plus one loss gather for the initial suffix")


TP_LEG_LIMIT_S = float(os.environ.get("BMA_TP_LEG_LIMIT_S", "150"))


class _LineGuard:
    """Several GPUs: a child of rank 0, started before anything touches the GPU, that HOLDS the first leg's finished line
    while the tensor-parallel leg runs and prints it if rank 0 goes away without saying it is done -- a fault inside a
    collective, or the launcher's SIGTERM after another rank died: exits no Python `except` or watchdog thread sees.  It
    reads a pipe to its end: a held line followed by the DONE mark (or nothing at all) prints nothing."""
    DONE = "\0DONE"
    CODE = ("import sys\n"
            "data = sys.stdin.read()\n"
            "if data and not data.rstrip('\\n').endswith('\\0DONE'):\n"
            "    line = data.split('\\n', 1)[0]\n"
            "    if line.strip():\n"
            "        sys.stdout.write(line + '\\n')\n"
            "        sys.stdout.flush()\n")

    def __init__(self):
        import subprocess
        self.p = subprocess.Popen([sys.executable, "-c", self.CODE], stdin=subprocess.PIPE, text=True)

    def hold(self, line: str) -> None:
        try:
            self.p.stdin.write(line + "\n")
            self.p.stdin.flush()
        except Exception:
            pass

    def release(self) -> None:
        """Rank 0 prints its own line from here on: the held one must not appear."""
        if self.p is None:
            return
        try:
            self.p.stdin.write(self.DONE + "\n")
            self.p.stdin.flush()
            self.p.stdin.close()
            self.p.wait(10)
        except Exception:
            pass
        self.p = None


_GUARD = None            # rank 0 of a multi-GPU run only
TP_HUNG_STATUS = 4       # exit status of every rank when the tensor-parallel leg's watchdog fires


def tp_ab(args, out: dict, device, world: int, rank: int) -> None:
    """Several GPUs: the SAME timed block a second time (same seed, same steps, same model object) with the batch-1
    gradient pass TENSOR-PARALLEL over the ranks (EngineOptions.tp_gradient, as one hipGraph with its RCCL all-reduces
    inside on the nccl backend) instead of replicated on each -- the replicated pass is the serial term of a sharded step
     and the tensor-parallel one has never met xGMI, so the first multi-GPU run carries its own A/B.  The
    better leg becomes the headline (`value`, `ms_per_step`, phases, losses); both are reported under
    `rccl: {tp_off_ms, tp_on_ms, chosen}`.  The first leg's line is complete before the second starts: if the second
    leg raises, or exceeds TP_LEG_LIMIT_S (a collective that never returns), the first leg's line is what is printed."""
    import threading
    import torch
    rc = out["rccl"]
    rc.update(tp_off_ms=out["ms_per_step"], tp_on_ms=None, chosen="off")
    if os.environ.get("BMA_BENCH_TP_AB", "1") in ("0", "false", "False"):
        rc["tp_note"] = "A/B skipped (BMA_BENCH_TP_AB=0)"
        return
    done = threading.Event()

    def watchdog():
        if done.wait(TP_LEG_LIMIT_S):
            return
        # a rank stuck in a collective cannot be unwound from Python: print what the finished leg measured and leave
        if rank == 0:
            rc["tp_note"] = f"tensor-parallel leg did not finish within {TP_LEG_LIMIT_S:.0f} s; the replicated leg stands"
            out["cpu_baseline"] = None
            if _GUARD is not None:
                _GUARD.release()
            print(json.dumps(build_line(out, None), allow_nan=False), flush=True)
        # Every rank leaves with TP_HUNG_STATUS, never 0: a collective that did not return is a deadlock, and a launcher or CI
        # must see one.  The first leg's line is already on stdout: "status 4 + a JSON line" reads as "the
        # replicated leg is valid, the tensor-parallel leg hung" (DESIGN.md 8).
        os._exit(TP_HUNG_STATUS)

    if rank == 0 and _GUARD is not None:
        held = dict(out, cpu_baseline=None)
        held["rccl"] = dict(rc, tp_note="the process ended inside the tensor-parallel leg (a fault, or the launcher's signal after "
                                        "another rank died); the replicated leg stands")
        _GUARD.hold(json.dumps(build_line(held, None), allow_nan=False))
    if os.environ.get("BMA_BENCH_TP_CRASH") and rank == 0:        # test hook: rank 0 dies as a faulting collective would
        import signal
        os.kill(os.getpid(), signal.SIGKILL)
    threading.Thread(target=watchdog, daemon=True).start()
    try:
        log("tensor-parallel gradient pass: the same timed block again (A/B against the replicated pass)")
        on, _ = measure(args, args.workload, args.steps, args.warmup, 0, device, world, rank, primary=True,
                        engine_kw=dict(tp_gradient="graph"))
        eng = on.get("engine") or {}
        rc["tp_on_ms"] = on["ms_per_step"]
        rc["tp_graph"] = "gradient_tp" in (eng.get("graphs_captured") or [])
        fb = {k: v for k, v in (eng.get("fallbacks") or {}).items() if k in ("tp_gradient", "graph_gradient_tp")}
        if fb:
            rc["tp_fallbacks"] = {k: str(v)[:120] for k, v in fb.items()}
        if "tp_gradient" not in fb and on.get("finite") and on["ms_per_step"] < out["ms_per_step"]:
            rc["chosen"] = "on"
            for k in ("value", "ms_per_step", "attack_steps_per_sec", "scoring_phase_candidate_forwards_per_sec", "phase_s_per_step",
                      "final_loss", "finite", "losses_timed", "engine", "forward_roofline", "gradient_pass_ms_each"):
                out[k] = on.get(k)
            out["config"]["gradient_pass"] = "tensor-parallel over the ranks"
            rc["tp_note"] = ("headline = the tensor-parallel leg; `roofline` and the per-kernel tables were taken in the replicated "
                             "leg's profiled steps (the scoring kernels are the same in both)")
    except Exception as e:              # the replicated leg's line stands on its own
        rc["tp_error"] = f"{type(e).__name__}: {e}"[:200]
        try:
            torch.cuda.synchronize(device)
        except Exception:
            pass
    finally:
        done.set()


def scaling_table(dist, out: dict, solo, world: int) -> dict:
    """north_star's table for THIS N, from this run alone: steps/s and candidate forwards/s on N GPUs, the same job on ONE of
    these GPUs (the run's own N = 1 leg: every rank ran it by itself, rank 0's is quoted, all are listed), the efficiency
    value_N / (N x value_1), and every rank's dominant-kernel roofline fraction.  (The driver computes its own efficiency
    from separate N = 1 / 2 / 4 / 8 runs; this one needs no post-processing and no second box.)  Every rank calls it."""
    r = out.get("roofline") or {}
    mine = dict(frac=r.get("frac"), solo=None if not solo or "error" in solo else solo.get("candidate_forwards_per_sec"))
    per_rank = [None] * world
    dist.all_gather_object(per_rank, mine)
    t = dict(n_gpus=world, attack_steps_per_sec=out.get("attack_steps_per_sec"), candidate_forwards_per_sec=out.get("value"),
             ms_per_step=out.get("ms_per_step"), own_n1_leg=solo,
             own_n1_leg_candidate_forwards_per_sec_per_rank=[p_["solo"] for p_ in per_rank],
             efficiency_vs_own_n1=None,
             dominant_kernel=dict(kernel=str(r.get("kernel"))[:90], bound=r.get("bound"), unit=r.get("unit"), peak=r.get("peak"),
                                  frac_per_rank=[p_["frac"] for p_ in per_rank]),
             gradient_pass=(out.get("config") or {}).get("gradient_pass", "replicated on every rank"))
    if solo and "error" not in solo and solo.get("candidate_forwards_per_sec") and out.get("value"):
        if solo["candidate_forwards_per_sec"] > 0:
            t["efficiency_vs_own_n1"] = out["value"] / (world * solo["candidate_forwards_per_sec"])
        elif solo.get("attack_steps_per_sec"):
            t["efficiency_vs_own_n1"] = out["attack_steps_per_sec"] / (world * solo["attack_steps_per_sec"])
    return t


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="timed steps (default 5; 8 for opt125m: 2 + 8 = its 10 steps)")
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="gcg", choices=sorted(WORKLOADS))
    ap.add_argument("--extra-workloads", default=None,
                    help="comma list of further workloads measured after --workload and reported under `workloads` "
                         "(default: joint,pgd,gemma_joint when --workload gcg runs on one GPU; 'none' to skip)")
    ap.add_argument("--extra-steps", type=int, default=5, help="timed steps of each extra workload (2 warm-up steps)")
    ap.add_argument("--search-width", type=int, default=None)
    ap.add_argument("--layers", type=int, default=32, help="debug only: fewer layers is NOT the benchmark")
    ap.add_argument("--profile-steps", type=int, default=2,
                    help="extra steps AFTER the timed region with HIP-event brackets around kernels (rooflines)")
    ap.add_argument("--no-gradient-profile", action="store_true", help="skip the eager per-GEMM profile of the gradient pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-width", type=int, default=32)
    ap.add_argument("--cpu-steps", type=int, default=2)
    ap.add_argument("--cpu-threads", type=int, default=0)
    ap.add_argument("--no-prefix-reuse", action="store_true")
    ap.add_argument("--detail", default=None, help="where the full result goes (default gpurun_out/bench_detail.json); "
                                                   "the stdout line is the short summary of it")
    args = ap.parse_args()
    if args.steps is None:
        args.steps = 8 if args.workload == "opt125m" else 5

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        self_launch(args)                       # never returns
    args.gpus = world
    if os.environ.get("BMA_BENCH_LAUNCH_PROBE"):        # CPU test of the launcher: report the rank layout, touch nothing
        sys.stdout.write(json.dumps({"probe": True, "rank": rank, "local_rank": local, "n_gpus": world,
                                     "master": os.environ.get("MASTER_ADDR")}) + "\n")     # one write: ranks share the pipe
        sys.stdout.flush()
        return

    global _GUARD
    if world > 1 and rank == 0 and os.environ.get("BMA_BENCH_TP_AB", "1") not in ("0", "false", "False"):
        try:
            _GUARD = _LineGuard()               # (a plain child process, started while this one has not touched the GPU)
        except Exception as e:                  # no guard, no harm: the watchdog and the early print still stand
            log(f"line guard not started: {type(e).__name__}: {e}")
            _GUARD = None

    import torch
    import torch.distributed as dist

    if os.environ.get("BMA_DIST_BACKEND", "nccl") != "nccl":
        local = 0                                               # rehearsal: every rank on the one GPU
    elif world > 1 and torch.cuda.device_count() < world:
        sys.exit(f"bench.py: {world} ranks asked for, {torch.cuda.device_count()} GPUs visible "
                 "(BMA_DIST_BACKEND=gloo rehearses the ranks on one GPU)")
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("BMA_DIST_BACKEND", "nccl")     # "gloo" only to rehearse ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from bimodalattack_amd import native
    native.check_single_hip_runtime()

    n_prof = max(0, args.profile_steps)
    solo = None
    if world > 1 and os.environ.get("BMA_BENCH_N1_LEG", "1") not in ("0", "false", "False"):
        # The run's OWN one-GPU leg: every rank runs the whole job by itself (EngineOptions.shard=False: no collectives), a few
        # steps, before the sharded legs -- what `scaling_table.efficiency_vs_own_n1` divides by (same box, same build, same clock
        # state as the N-GPU legs; the driver's separate N = 1 run is another box hours apart).
        try:
            s1, _ = measure(args, args.workload, min(args.steps, 5), 2, 0, device, 1, 0, primary=True,
                            engine_kw=dict(shard=False, tp_gradient=False))
            solo = dict(ms_per_step=s1["ms_per_step"], candidate_forwards_per_sec=s1["value"], attack_steps_per_sec=s1["attack_steps_per_sec"],
                        steps=s1["steps"], finite=s1["finite"])
        except Exception as e:
            solo = dict(error=f"{type(e).__name__}: {e}"[:200])
            torch.cuda.synchronize(device)
    out, keep = measure(args, args.workload, args.steps, args.warmup, n_prof, device, world, rank, primary=True,
                        engine_kw=dict(tp_gradient=False) if world > 1 else None)
    if world > 1:
        out["rccl"] = rccl_info(torch, dist, device, world, keep)
        tp_ab(args, out, device, world, rank)
        out["scaling_table"] = scaling_table(dist, out, solo, world)

    # ---- the other single-GPU BASELINE configurations, on the same models, under `workloads` ----------
    if args.extra_workloads is None:
        emulating = int(os.environ.get("BMA_EMULATE_WORLD", "0") or 0) > 1
        extra = ["joint", "pgd", "gemma_joint"] if (args.workload == "gcg" and world == 1 and args.layers == 32
                                                     and args.search_width is None and not emulating) else []
    else:
        extra = [w for w in args.extra_workloads.split(",") if w and w != "none"]
    if extra:
        out["workloads"] = {}
        short = ("ms_per_step", "value", "unit", "steps", "warmup", "attack_steps_per_sec",
                 "scoring_phase_candidate_forwards_per_sec", "phase_s_per_step", "roofline", "forward_roofline",
                 "gradient_pass", "engine", "config", "final_loss", "finite", "losses_timed")
        for w in extra:
            if w not in WORKLOADS or w == args.workload:
                continue
            try:
                r, _ = measure(args, w, args.extra_steps, 2, n_prof, device, world, rank, primary=False)
                out["workloads"][w] = {k: r[k] for k in short}
                if w == "pgd":
                    out["workloads"][w]["note"] = ("PGD-only scores no candidates: `value` counts one forward per step, "
                                                   "i.e. it IS attack steps per second")
            except Exception as e:       # the headline line stands on its own
                out["workloads"][w] = {"error": f"{type(e).__name__}: {e}"}
                torch.cuda.synchronize(device)
            log(f"workload {w} done")

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        try:
            _MODELS.clear()
            out["cpu_baseline"] = cpu_baseline(args, keep["wl"], keep["model"], keep["tok"], keep["proc"], keep["messages"],
                                               keep["goal"], keep["target"], keep["image"], keep["norm"], keep["cfg_kw"])
        except Exception as e:  # the GPU numbers stand on their own
            out["cpu_baseline"] = {"value": None, "unit": "candidate_forwards/s", "cores": os.cpu_count(), "kind": "port",
                                   "sample": f"failed: {type(e).__name__}: {e}"}
    else:
        out["cpu_baseline"] = None
    status = 0
    if rank == 0:
        detail = args.detail or os.path.join(REPO, "gpurun_out", "bench_detail.json")
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail)), exist_ok=True)
            with open(detail, "w") as fh:
                json.dump(_strict(out), fh, indent=1)
            log(f"full detail (per-GEMM tables, kernels, engine state, notes) written to {detail}")
        except OSError as e:
            log(f"detail file not written: {e}")
            detail = None
        line = build_line(out, None if detail is None else os.path.relpath(detail, REPO))
        if _GUARD is not None:
            _GUARD.release()
        print(json.dumps(line, allow_nan=False), flush=True)
        if not line["finite"]:
            log("NON-FINITE loss inside the timed steps of the headline workload: the number above is not a measurement")
            status = 3
    if world > 1:
        # the line is out: a teardown that hangs or raises (ranks that left the tensor-parallel leg by different doors) must
        # not turn it into a failed run
        import threading
        timer = threading.Timer(60.0, lambda: os._exit(status))
        timer.daemon = True
        timer.start()
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as e:
            log(f"process group teardown: {type(e).__name__}: {e}")
        timer.cancel()
    if status:
        sys.exit(status)


if __name__ == "__main__":
    main()
