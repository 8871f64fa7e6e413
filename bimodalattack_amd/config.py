"""API types of the attack engine.

Field-for-field compatible with the reference's dataclasses so that callers
(`experiments.py:86-112` in the reference) can switch packages unchanged:

* ``BimodalAttackConfig``  <- reference bimodalattack/bimodal_attack.py:42-70 (27 fields)
* ``BimodalAttackResult``  <- reference bimodalattack/bimodal_attack.py:73-85 (11 fields)
* ``GCGConfig``            alias; nanoGCG's name for the same config.

Engine-only knobs (backend, sharding, RNG placement) deliberately live OUTSIDE
the dataclass -- see ``EngineOptions`` -- so the dataclass stays drop-in.
"""

from __future__ import annotations

import os
from dataclasses import dataclass, field
from typing import List, Optional, Union


@dataclass
class BimodalAttackConfig:
    # -- GCG side ---------------------------------------------------------
    num_steps: int = 250
    optim_str_init: Union[str, List[str]] = "x x x x x x x x x x x x x x x x x x x"
    search_width: int = 512
    batch_size: Optional[int] = None
    topk: int = 256
    n_replace: int = 1
    buffer_size: int = 0
    use_mellowmax: bool = False          # declared by the reference, never read
    mellowmax_alpha: float = 1.0         # declared by the reference, never read
    early_stop: bool = False
    allow_non_ascii: bool = False
    filter_ids: bool = True
    add_space_before_target: bool = False  # declared by the reference, never read
    seed: Optional[int] = None
    verbosity: str = "INFO"
    dynamic_search: bool = False
    min_search_width: int = 32
    # -- PGD side ---------------------------------------------------------
    alpha: float = 0.01
    eps: float = 0.1
    pgd_attack: bool = False
    gcg_attack: bool = True
    debug_output: bool = False
    joint_eval: bool = False
    experiment_folder: str = "experiments/missing_folder"  # never read
    images_folder: str = "experiments/missing_folder/images"
    pgd_after_gcg: bool = False
    model: str = "llava"                 # declared by the reference, never read


# nanoGCG spelling, named in BASELINE.json's north_star.
GCGConfig = BimodalAttackConfig


@dataclass
class BimodalAttackResult:
    best_loss: float
    best_string: str
    losses: List[float]
    strings: List[str]
    adversarial_suffixes: List[str]
    model_outputs: List[str]
    gradient_times: List[float]
    sampling_times: List[float]
    loss_times: List[float]
    pgd_times: List[float]
    total_times: List[float] = None


@dataclass
class EngineOptions:
    """Knobs of the MI355X engine that the reference has no field for.

    Read from keyword arguments of :func:`bimodalattack_amd.run` or from the
    environment (``BMA_*``), never from ``BimodalAttackConfig``.
    """

    # "model": draw the sampling randoms on the model's device exactly where the
    # reference draws them (bimodal_attack.py:151,159).  "cpu": draw them from the
    # CPU generator and upload -- reproduces the reference's CPU path bit for bit
    # ("RNG").
    rng_device: str = "model"
    # Re-use the keys/values of the segments in front of the suffix across all
    # candidates of a step (identical maths under causal attention).
    prefix_reuse: bool = True
    # Ask the model for logits on the target rows only (`logits_to_keep`).
    target_rows_only: bool = True
    # Capture the fixed-shape batch-1 gradient pass (forward + backward) into a hipGraph
    # on its first call (after one eager run) and replay it every step: the pass is launch-bound
    # (~1500 small kernels around 7 ms of weight streaming).  Falls back to eager, once and
    # for good, if the model's forward cannot be captured.
    graph_gradient: bool = True
    # Likewise for the batch-1 re-scoring of the step winner with the image (reference
    # :605-612; every PGD mode): prefix pass + tail forward + CE are launch-bound at batch 1.
    graph_rescore: bool = True
    # ... and for the per-step image-features + shared-prefix pass of joint scoring (vision tower
    # forward, then the prompt+image prefix through the LM with a recording cache).
    graph_prefix: bool = True
    # Run RMSNorm / SwiGLU / rotary embedding of Llama-family models through the fused
    # one-pass kernels while scoring candidates (no autograd there); see fused.py.
    fused_elementwise: bool = True
    # Gradient pass (batch 1, <= 1024 rows): keep a transposed copy of every decoder
    # projection weight so the backward product streams weight rows along the reduction like the
    # forward one (the faster library form for such shapes).  Costs one more copy of the LM weights.
    backward_weight_copies: bool = True
    # q_proj/k_proj/v_proj of an attention block as one GEMM against the concatenated weight (16-bit
    # models; one more copy of those matrices): fewer partly filled tile rounds, one weight stream.
    fuse_qkv: bool = True
    # Gradient pass: ask the library attention for `is_causal` instead of handing it the (1,1,S,S) mask tensor
    # HuggingFace builds for inputs_embeds calls (the causal flash kernels instead of the masked ones: -3 ms per
    # pass at the 643 tokens of the image prompt).  Model families with plain causal text attention only.
    maskless_b1_attention: bool = True
    # joint mode, image in front of the suffix: the scoring prefix pass runs with autograd and serves as the first
    # 599 rows of the next gradient pass, which then runs 44 rows forward instead of 644 (attack._GradPrefix)
    grad_prefix_reuse: bool = True
    # vision towers whose head width is not a multiple of 32 (SigLIP: 72): zero-pad q/k/v to a width the library's
    # attention kernels are built for (prefix_attention.padded_heads_attention); same attention, faster kernels
    pad_vision_heads: bool = True
    # CLIP's QuickGELU (x * sigmoid(1.702 x): three launches forward, five backward on a launch-bound tower) as one
    # launch each way, bit-identical (bma_quick_gelu)
    fuse_quick_gelu: bool = True
    # the vision tower's q/k/v projections (with their biases) as one product, forward and input-gradient
    fuse_tower_qkv: bool = True
    # gate_proj / up_proj of a gated MLP as one GEMM against their chunk-interleaved weights (16-bit models; one
    # more copy of those two matrices, two in the gradient pass): see fused.py.
    fuse_gate_up: bool = True
    # every residual add of a decoder layer fused into the RMSNorm that follows it (also across the layer boundary) and
    # q/k rotary as one launch: see fused.py (_layer_forward); known llama- / gemma3-style layer structures only.
    fuse_add_norm: bool = True
    # Gemma-3's per-head q_norm / k_norm inside the rotary launch of the no-grad scoring forward (bma_qknorm_rope2): one
    # pass over q and k instead of two
    fuse_qk_rope: bool = True
    # batch-1 gradient pass over a short text-only prompt (<= 80 tokens; llama-style attention blocks with 128-wide heads and
    # no grouped heads): rotary embedding + causal attention between the fused q/k/v projection and o_proj as ONE launch
    # forward and ONE backward (bma_b1_attention) instead of HuggingFace's rotary + library attention and, under autograd,
    # the library's attention backward, a counter fill, the rotary backward and a concatenation.
    fuse_b1_attention: bool = True
    # batch-1 gradient pass: products with at most 96 rows (16-bit, bias-free decoder projections and their input
    # gradients through the transposed copies) on the hand-written weight-streaming kernel bma_gemm_nt instead of the
    # library (process-wide switch: ops.SKINNY_GEMM).
    skinny_gemm: bool = True
    causal_attention: bool = True       # batch-1 causal attention of a long prompt (and of the rows behind a reused prefix) on csrc/causal_attention.hip, forward and backward
    mid_gemm: bool = True               # the 599-644-row products of the pass with the image in the prompt on csrc/gemm_mid.hip
    # Several GPUs: run the batch-1 gradient pass TENSOR-PARALLEL over the ranks instead of redundantly on each --
    # q/k/v/gate/up cut by output rows (whole heads), o/down by input columns, two all-reduces per decoder layer and
    # direction (540 KB each at LLaVA width) -- the one lever left on the serial quarter of an 8-GPU step.
    # Correctness is tested (2 ranks, equal to the replicated pass); its speed has never been measured on real xGMI
    # (this pool has one GPU per box), so it is OFF by default; `bench.py --gpus N` times it against the replicated pass.
    tp_gradient: bool = False
    # ... and that tensor-parallel pass as ONE hipGraph with its RCCL all-reduces inside (nccl backend only).  Opt-in:
    # no run on two or more GPUs has been recorded yet.  The capture's outcome is agreed on by all ranks --
    # one that fails makes every rank run the pass eagerly (dist.CandidateSharder.all_ok).
    tp_graph: bool = False
    # joint_eval: the step's loss is the winner's row of the candidate batch (scored with the image, in the
    # re-score's own segment order) instead of a second, batch-1 forward of the same sequence (:605-612).
    joint_winner_from_batch: bool = True
    # GEMM tuning aid (tools/tune_gemms.py): in a single process, score only what rank 0 of an N-rank run
    # would score, with that run's row budget -- the exact GEMM shapes of the multi-GPU run.  Results of the
    # attack are meaningless with it.
    emulate_world: int = 0
    # Attend to the shared prefix without copying its keys/values into every candidate
    # (two flash launches + a merge kernel; llama-family text models).  See prefix_attention.py.
    shared_prefix_attention: bool = True
    shared_prefix_min_tokens: int = 1
    # On top of it: compute, per candidate, only the tokens from its first replaced suffix
    # position on -- the ones in front equal the parent suffix, whose keys/values are shared
    # (layout.ragged_plan).  ~ (n_opt-1)/(2L) fewer rows through every GEMM, norm and MLP gate.
    ragged_suffix: bool = True
    # GEMM selection: "auto" loads bimodalattack_amd/tuning/<arch>.csv into PyTorch's
    # TunableOp in lookup-only mode when its validators (torch / hipBLASLt / rocBLAS versions,
    # arch) match this process; "off" leaves the library heuristics alone.
    gemm_tuning: str = "auto"
    # PGD-only: the forward that scores the updated image at step i IS the forward of step
    # i+1's gradient pass (same ids, same image).  Run it once: one forward+backward per step
    # instead of two forwards and a backward.  Same numbers, same lists in the result.
    fuse_pgd_only: bool = True
    # GCG-only steps and joint steps whose winner's loss comes from the batch: the NEXT step's gradient pass needs nothing from the host -- its
    # input is the winner, a device-side argmin of the losses -- so it is queued right behind the scoring forward,
    # before the host reads this step's outcome.  The host's work at the step boundary (one packed read-back,
    # decoding, the buffer, logging) and the launch of the gradient graph then happen while the GPU is busy instead
    # of in front of an idle one.  Same kernels on the same inputs in the same stream order; the phase times of the
    # result come from stream events instead of host clocks around synchronisations.
    gradient_ahead: bool = True
    # ... and with the gradient pass queued ahead, the ragged plan of the NEXT scoring forward is made while that pass
    # runs: the random draws of a sampling step (positions, top-k ranks) do not depend on the gradient, so they are
    # made -- in the same order of the same generator -- before the pass is queued, copied to the host behind it, and
    # the host plans on "virtual" ids (the parent with a placeholder per drawn (position, rank)) that coincide exactly
    # where the real candidates must; the real ids never visit the host before the forward (the retokenisation filter
    # still gets its copy, beside the forward), they are gathered on the device.
    early_plan: bool = True
    # The retokenisation filter (reference :166-186) BEFORE scoring, as the reference runs it, instead of beside it with the
    # losses masked afterwards: True / False force one order; None (default) picks per step from the filter's survivor rate
    # over the last steps -- score-everything while (almost) everything survives, filter-first once a real share is
    # rejected (attack.FILTER_COST_RATIO: below 97 % on one GPU, 79 % on eight).  Same winners either way.
    filter_first: Optional[bool] = None
    # Before the first step: one product of every decoder projection shape at every row count the run's ragged forwards
    # can meet (layout.expected_row_counts), so that the library's lazy loading of a kernel it has not used in this process
    # -- ~39 ms the first time a new row count shows up -- happens in set-up and not inside a step.
    warm_gemms: bool = True
    # Candidates per forward chunk; None = size analytically from free HBM.
    chunk: Optional[int] = None
    # Padded scoring (no ragged rows: Gemma-3's layout, fp32 models) runs chunks whose candidate count is a
    # multiple of this -- a short last chunk is padded with copies of its last candidate, whose losses are
    # dropped -- so a decaying search width (reference :919-923) meets a handful of GEMM shapes, not hundreds.
    # 1 switches it off.  Ignored when config.batch_size fixes the chunk.
    chunk_quantum: int = 8
    # Measurement only: the candidate count of loop step i (bench.py samples the dynamic-width schedule of a
    # 600-step run, reference :919-923, at evenly spaced points of a handful of timed steps).  None: the
    # reference's schedule of this run's own num_steps.
    width_override: object = None
    # Raise instead of falling back when a fast path (hipGraph capture, shared-prefix attention, ragged
    # scoring, prefix reuse) fails on this model.  Off by default: an unknown model family must still run.
    strict: bool = False
    # Write images_folder/{i}.png every step (reference side effect, :744).
    save_images: bool = True
    # Record per-step internals (sampled ids, N after filter, best_idx, ...).
    trace: Optional[list] = None
    # Debugging aid: a list that receives one dict per scoring call (route taken, sizes, count of non-finite losses as a
    # device scalar); see tools/nan_bisect.py.
    score_log: Optional[list] = None
    # torch.distributed process group used to shard candidate scoring; None =
    # the default group when initialised, else single process.
    group: object = None
    # Called as step_hook(i) at the start of step i and once more, with i = the number
    # of steps run, after the last one (bench.py brackets its timed region with it).
    step_hook: object = None
    # Round per-candidate losses to the model dtype before the argmin, as the
    # reference's model-dtype cross-entropy does ("quirks").
    loss_in_model_dtype: bool = True

    @classmethod
    def from_env(cls, **overrides) -> "EngineOptions":
        opts = cls()
        env = os.environ
        if "BMA_RNG_DEVICE" in env:
            opts.rng_device = env["BMA_RNG_DEVICE"]
        if "BMA_PREFIX_REUSE" in env:
            opts.prefix_reuse = env["BMA_PREFIX_REUSE"] not in ("0", "false", "False")
        if "BMA_TARGET_ROWS_ONLY" in env:
            opts.target_rows_only = env["BMA_TARGET_ROWS_ONLY"] not in ("0", "false", "False")
        if "BMA_GRAPH_RESCORE" in env:
            opts.graph_rescore = env["BMA_GRAPH_RESCORE"] not in ("0", "false", "False")
        if "BMA_GRAPH_PREFIX" in env:
            opts.graph_prefix = env["BMA_GRAPH_PREFIX"] not in ("0", "false", "False")
        if "BMA_GRAPH_GRADIENT" in env:
            opts.graph_gradient = env["BMA_GRAPH_GRADIENT"] not in ("0", "false", "False")
        if "BMA_FUSED_ELEMENTWISE" in env:
            opts.fused_elementwise = env["BMA_FUSED_ELEMENTWISE"] not in ("0", "false", "False")
        if "BMA_EMULATE_WORLD" in env:
            opts.emulate_world = int(env["BMA_EMULATE_WORLD"])
        if "BMA_JOINT_WINNER_FROM_BATCH" in env:
            opts.joint_winner_from_batch = env["BMA_JOINT_WINNER_FROM_BATCH"] not in ("0", "false", "False")
        if "BMA_FUSE_QKV" in env:
            opts.fuse_qkv = env["BMA_FUSE_QKV"] not in ("0", "false", "False")
        if "BMA_MASKLESS_B1_ATTENTION" in env:
            opts.maskless_b1_attention = env["BMA_MASKLESS_B1_ATTENTION"] not in ("0", "false", "False")
        if "BMA_GRAD_PREFIX_REUSE" in env:
            opts.grad_prefix_reuse = env["BMA_GRAD_PREFIX_REUSE"] not in ("0", "false", "False")
        if "BMA_PAD_VISION_HEADS" in env:
            opts.pad_vision_heads = env["BMA_PAD_VISION_HEADS"] not in ("0", "false", "False")
        if "BMA_FUSE_QUICK_GELU" in env:
            opts.fuse_quick_gelu = env["BMA_FUSE_QUICK_GELU"] not in ("0", "false", "False")
        if "BMA_FUSE_TOWER_QKV" in env:
            opts.fuse_tower_qkv = env["BMA_FUSE_TOWER_QKV"] not in ("0", "false", "False")
        if "BMA_FUSE_GATE_UP" in env:
            opts.fuse_gate_up = env["BMA_FUSE_GATE_UP"] not in ("0", "false", "False")
        if "BMA_TP_GRADIENT" in env:
            opts.tp_gradient = env["BMA_TP_GRADIENT"] not in ("0", "false", "False")
        if "BMA_TP_GRAPH" in env:
            opts.tp_graph = env["BMA_TP_GRAPH"] not in ("0", "false", "False")
        if "BMA_SKINNY_GEMM" in env:
            opts.skinny_gemm = env["BMA_SKINNY_GEMM"] not in ("0", "false", "False")
        if "BMA_CAUSAL_ATTENTION" in env:
            opts.causal_attention = env["BMA_CAUSAL_ATTENTION"] not in ("0", "false", "False")
        if "BMA_MID_GEMM" in env:
            opts.mid_gemm = env["BMA_MID_GEMM"] not in ("0", "false", "False")
        if "BMA_FUSE_ADD_NORM" in env:
            opts.fuse_add_norm = env["BMA_FUSE_ADD_NORM"] not in ("0", "false", "False")
        if "BMA_FUSE_B1_ATTENTION" in env:
            opts.fuse_b1_attention = env["BMA_FUSE_B1_ATTENTION"] not in ("0", "false", "False")
        if "BMA_FUSE_QK_ROPE" in env:
            opts.fuse_qk_rope = env["BMA_FUSE_QK_ROPE"] not in ("0", "false", "False")
        if "BMA_BACKWARD_WEIGHT_COPIES" in env:
            opts.backward_weight_copies = env["BMA_BACKWARD_WEIGHT_COPIES"] not in ("0", "false", "False")
        if "BMA_SHARED_PREFIX_ATTENTION" in env:
            opts.shared_prefix_attention = env["BMA_SHARED_PREFIX_ATTENTION"] not in ("0", "false", "False")
        if "BMA_RAGGED_SUFFIX" in env:
            opts.ragged_suffix = env["BMA_RAGGED_SUFFIX"] not in ("0", "false", "False")
        if "BMA_GEMM_TUNING" in env:
            opts.gemm_tuning = env["BMA_GEMM_TUNING"]
        if "BMA_FUSE_PGD_ONLY" in env:
            opts.fuse_pgd_only = env["BMA_FUSE_PGD_ONLY"] not in ("0", "false", "False")
        if "BMA_GRADIENT_AHEAD" in env:
            opts.gradient_ahead = env["BMA_GRADIENT_AHEAD"] not in ("0", "false", "False")
        if "BMA_EARLY_PLAN" in env:
            opts.early_plan = env["BMA_EARLY_PLAN"] not in ("0", "false", "False")
        if "BMA_FILTER_FIRST" in env:
            opts.filter_first = env["BMA_FILTER_FIRST"] not in ("0", "false", "False")
        if "BMA_WARM_GEMMS" in env:
            opts.warm_gemms = env["BMA_WARM_GEMMS"] not in ("0", "false", "False")
        if "BMA_CHUNK_QUANTUM" in env:
            opts.chunk_quantum = max(1, int(env["BMA_CHUNK_QUANTUM"]))
        if "BMA_CHUNK" in env:
            opts.chunk = int(env["BMA_CHUNK"])
        if "BMA_STRICT" in env:
            opts.strict = env["BMA_STRICT"] not in ("0", "false", "False")
        if "BMA_SAVE_IMAGES" in env:
            opts.save_images = env["BMA_SAVE_IMAGES"] not in ("0", "false", "False")
        for k, v in overrides.items():
            if v is None:
                continue
            if not hasattr(opts, k):
                raise TypeError(f"unknown engine option {k!r}")
            setattr(opts, k, v)
        if opts.rng_device not in ("model", "cpu"):
            raise ValueError(f"rng_device must be 'model' or 'cpu', got {opts.rng_device!r}")
        return opts
