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
    """Knobs of the MI355X engine that the reference has no field for: the ones a test or bench.py flips.

    Read from keyword arguments of :func:`bimodalattack_amd.run` or from the environment (``BMA_<NAME>``), never from
    ``BimodalAttackConfig``.  Every one of them only changes speed (tests/test_attack_gpu.py runs the reference's golden
    trajectories with each restructuring off).  Finer switches that exist for A/B measurements only are module constants
    read from the environment once (round 5: 42 options -> 27): ``ops.OWN_KERNELS`` (each hand-written batch-1 kernel),
    ``attack.MASKLESS_B1_ATTENTION`` / ``CHUNK_QUANTUM`` / ``EMULATE_WORLD``, ``fused.FUSE_QK_ROPE``,
    ``hf_adapter.PAD_VISION_HEADS`` / ``FUSE_QUICK_GELU`` / ``FUSE_TOWER_QKV``.
    """

    # "model": draw the sampling randoms on the model's device exactly where the reference draws them
    # (bimodal_attack.py:151,159).  "cpu": draw them from the CPU generator and upload -- reproduces the reference's CPU
    # path bit for bit ("RNG").
    rng_device: str = "model"
    # Re-use the keys/values of the segments in front of the suffix across all candidates of a step (identical maths under
    # causal attention).
    prefix_reuse: bool = True
    # Ask the model for logits on the target rows only (`logits_to_keep`).
    target_rows_only: bool = True
    # Capture the fixed-shape batch-1 gradient pass (forward + backward) into a hipGraph on its first call (after one
    # eager run) and replay it every step: the pass is launch-bound (~1500 small kernels around 7 ms of weight
    # streaming).  Falls back to eager, once and for good, if the model's forward cannot be captured.
    graph_gradient: bool = True
    # Likewise for the batch-1 work of the SCORING side: the re-scoring of the step winner with the image (reference
    # :605-612; every PGD mode) and the per-step image-features + shared-prefix pass of joint scoring (vision tower
    # forward, then the prompt+image prefix through the LM with a recording cache).
    graph_scoring: bool = True
    # Run RMSNorm / SwiGLU / rotary embedding of Llama- and Gemma-family models through the fused one-pass kernels; see
    # fused.py.
    fused_elementwise: bool = True
    # Derived copies of the decoder weights (16-bit models; they belong to the MODEL object and are rebuilt when a source
    # weight changes): a transposed copy of every bias-free projection, so that the gradient pass's input-gradient products
    # stream weight rows along the reduction like the forward ones; q/k/v concatenated into one product; gate/up
    # chunk-interleaved into one product (+ its transpose).  ~28 GB for LLaVA-1.5-7B on a 288 GB device; False gives the
    # memory back for a few per cent of speed.
    derived_weight_copies: bool = True
    # The hand-written batch-1 kernels of the gradient pass: bma_gemm_nt (<= 96 rows), bma_gemm_mid (560-672 rows: the
    # image in the prompt), bma_b1_attention (rotary + attention of a prompt of <= 80 tokens in one launch each way) and
    # bma_causal_attention (longer prompts, the rows behind a reused prefix, the vision tower).  False: the library's
    # products and attention everywhere (process-wide switches: ops.SKINNY_GEMM / MID_GEMM / CAUSAL_ATTENTION).
    own_b1_kernels: bool = True
    # joint mode, image in front of the suffix: the scoring prefix pass runs with autograd and serves as the first 599
    # rows of the next gradient pass, which then runs 44 rows forward instead of 644 (attack._GradPrefix)
    grad_prefix_reuse: bool = True
    # every residual add of a decoder layer fused into the RMSNorm that follows it (also across the layer boundary) and
    # q/k rotary as one launch: see fused.py (_layer_forward); known llama- / gemma3-style layer structures only.
    fuse_add_norm: bool = True
    # Several GPUs: run the batch-1 gradient pass TENSOR-PARALLEL over the ranks instead of redundantly on each -- q/k/v/
    # gate/up cut by output rows (whole heads), o/down by input columns, two all-reduces per decoder layer and direction
    # (540 KB each at LLaVA width) -- the one lever left on the serial quarter of an 8-GPU step.  True:
    # eagerly; "graph": as ONE hipGraph with its RCCL all-reduces inside (nccl backend; the capture's outcome is agreed on
    # by all ranks, dist.CandidateSharder.all_ok).  Correctness is tested (2 ranks, equal to the replicated pass); its
    # speed has never been measured on real xGMI (this pool has one GPU per box), so it is OFF by default;
    # `bench.py --gpus N` times it against the replicated pass and reports both.
    tp_gradient: object = False
    # joint_eval: the step's loss is the winner's row of the candidate batch (scored with the image, in the re-score's
    # own segment order) instead of a second, batch-1 forward of the same sequence (:605-612).
    joint_winner_from_batch: bool = True
    # Attend to the shared prefix without copying its keys/values into every candidate (llama-family text models).  See
    # prefix_attention.py.
    shared_prefix_attention: bool = True
    # On top of it: compute, per candidate, only the tokens from its first replaced suffix position on -- the ones in
    # front equal the parent suffix, whose keys/values are shared (layout.ragged_plan).
    ragged_suffix: bool = True
    # GEMM selection: "auto" loads bimodalattack_amd/tuning/<arch>.csv into PyTorch's TunableOp in lookup-only mode when
    # its validators match this process; "off" leaves the library heuristics alone.
    gemm_tuning: str = "auto"
    # PGD-only: the forward that scores the updated image at step i IS the forward of step i+1's gradient pass (same
    # ids, same image).  Run it once.  Same numbers, same lists in the result.
    fuse_pgd_only: bool = True
    # GCG-only steps and joint steps whose winner's loss comes from the batch: the NEXT step's gradient pass is queued
    # right behind the scoring forward, before the host reads this step's outcome (one packed read-back per step).
    gradient_ahead: bool = True
    # ... and the ragged plan of the NEXT scoring forward is made while that pass runs, from the random draws of the
    # sampling step (made -- in the same order of the same generator -- before the pass is queued).
    early_plan: bool = True
    # The retokenisation filter (reference :166-186) BEFORE scoring, as the reference runs it, instead of beside it with the
    # losses masked afterwards: True / False force one order; None (default) picks per step from the filter's survivor rate
    # over the last steps -- score-everything while (almost) everything survives, filter-first once a real share is
    # rejected (attack.FILTER_COST_RATIO: below 97 % on one GPU, 79 % on eight).  Same winners either way.
    filter_first: Optional[bool] = None
    # Before the first step: one product of every decoder projection shape at every row count the run's ragged forwards
    # can meet (layout.expected_row_counts), so that the library's lazy loading of a kernel happens in set-up.
    warm_gemms: bool = True
    # Candidates per forward chunk; None = size analytically from free HBM.
    chunk: Optional[int] = None
    # Several ranks (torch.distributed initialised): None = shard the candidates over them; False = every rank runs the whole
    # job on its own, no collectives (bench.py's own one-GPU leg inside an N-GPU run: what N GPUs are measured against).
    shard: Optional[bool] = None
    # Measurement only: the candidate count of loop step i (bench.py samples the dynamic-width schedule of a 600-step run,
    # reference :919-923, at evenly spaced points of a handful of timed steps).  None: the reference's schedule.
    width_override: object = None
    # Raise instead of falling back when a fast path (hipGraph capture, shared-prefix attention, ragged scoring, prefix
    # reuse) fails on this model.  Off by default: an unknown model family must still run.
    strict: bool = False
    # Write images_folder/{i}.png every step (reference side effect, :744).
    save_images: bool = True
    # Record per-step internals (sampled ids, N after filter, best_idx, ...).
    trace: Optional[list] = None
    # Called as step_hook(i) at the start of step i and once more, with i = the number of steps run, after the last one
    # (bench.py brackets its timed region with it).
    step_hook: object = None
    # Round per-candidate losses to the model dtype before the argmin, as the reference's model-dtype cross-entropy does
    # ("quirks").
    loss_in_model_dtype: bool = True

    _BOOLS = ("prefix_reuse", "target_rows_only", "graph_gradient", "graph_scoring", "fused_elementwise", "derived_weight_copies",
              "own_b1_kernels", "grad_prefix_reuse", "fuse_add_norm", "joint_winner_from_batch", "shared_prefix_attention",
              "ragged_suffix", "fuse_pgd_only", "gradient_ahead", "early_plan", "filter_first", "warm_gemms", "strict", "save_images")

    # Names round 5's consolidation retired (42 options -> 27): an A/B script that still exports one would silently measure
    # the default -- from_env says so once per name.  Value: what replaces it.
    _RETIRED_ENV = {
        "BMA_GRAPH_RESCORE": "BMA_GRAPH_SCORING", "BMA_GRAPH_PREFIX": "BMA_GRAPH_SCORING", "BMA_SCORE_GRAPHS": "BMA_GRAPH_SCORING",
        "BMA_BACKWARD_WEIGHT_COPIES": "BMA_DERIVED_WEIGHT_COPIES", "BMA_FUSE_QKV": "BMA_DERIVED_WEIGHT_COPIES",
        "BMA_FUSE_GATE_UP": "BMA_DERIVED_WEIGHT_COPIES", "BMA_FUSE_TOWER_QKV": "BMA_DERIVED_WEIGHT_COPIES",
        "BMA_FUSE_QK_ROPE": "BMA_FUSE_ADD_NORM", "BMA_FUSE_QUICK_GELU": "BMA_FUSED_ELEMENTWISE", "BMA_PAD_VISION_HEADS": None,
        "BMA_TP_GRAPH": "BMA_TP_GRADIENT=graph", "BMA_SHARED_PREFIX_MIN_TOKENS": None, "BMA_CHUNK_QUANTUM": None,
    }
    _RETIRED_KW = {"group": "the default process group is used when torch.distributed is initialised (dist.CandidateSharder)",
                   "score_log": "set `attack.score_log = []` on the BimodalAttack object (tools/nan_bisect.py)",
                   "emulate_world": "the BMA_EMULATE_WORLD environment variable", "tp_graph": "tp_gradient='graph'",
                   "graph_rescore": "graph_scoring", "graph_prefix": "graph_scoring", "backward_weight_copies": "derived_weight_copies",
                   "fuse_qkv": "derived_weight_copies", "fuse_gate_up": "derived_weight_copies", "skinny_gemm": "own_b1_kernels",
                   "mid_gemm": "own_b1_kernels", "causal_attention": "own_b1_kernels"}
    _warned_retired = set()

    @classmethod
    def from_env(cls, **overrides) -> "EngineOptions":
        opts = cls()
        env = os.environ
        off = ("0", "false", "False")
        for name, instead in cls._RETIRED_ENV.items():
            if name in env and name not in cls._warned_retired:
                cls._warned_retired.add(name)
                import logging
                logging.getLogger("gcg").warning(
                    f"{name} is set but no longer read (engine options were consolidated in round 5): "
                    + (f"use {instead}" if instead else "it has no replacement"))
        for name in cls._BOOLS:
            key = "BMA_" + name.upper()
            if key in env:
                setattr(opts, name, env[key] not in off)
        if "BMA_RNG_DEVICE" in env:
            opts.rng_device = env["BMA_RNG_DEVICE"]
        if "BMA_GEMM_TUNING" in env:
            opts.gemm_tuning = env["BMA_GEMM_TUNING"]
        if "BMA_TP_GRADIENT" in env:
            opts.tp_gradient = "graph" if env["BMA_TP_GRADIENT"] == "graph" else env["BMA_TP_GRADIENT"] not in off
        if "BMA_CHUNK" in env:
            opts.chunk = int(env["BMA_CHUNK"])
        for k, v in overrides.items():
            if v is None:
                continue
            if k in cls._RETIRED_KW:
                raise TypeError(f"engine option {k!r} was retired in round 5: {cls._RETIRED_KW[k]}")
            if k.startswith("_") or not hasattr(opts, k):
                raise TypeError(f"unknown engine option {k!r}")
            setattr(opts, k, v)
        if opts.rng_device not in ("model", "cpu"):
            raise ValueError(f"rng_device must be 'model' or 'cpu', got {opts.rng_device!r}")
        if opts.tp_gradient not in (False, True, "graph"):
            raise ValueError(f"tp_gradient must be False, True or 'graph', got {opts.tp_gradient!r}")
        return opts
