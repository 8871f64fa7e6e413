"""The attack engine: the per-step loop of joint GCG (token suffix) + PGD (image
pixels) optimisation, MI355X-native, behind the reference's ``run()`` API.

Mirrors /root/reference/bimodalattack/bimodal_attack.py:
  ``BimodalAttack.__init__`` :193-249      chat-template defaults, forbidden tokens
  ``BimodalAttack.run``      :251-824      prompt split, step loop, bookkeeping
  ``init_buffer``            :826-906      ``candidate_sampling`` :908-951
  ``compute_gradient``       :953-1028     ``perform_pgd_step``   :1030-1037
  ``_build_input_embeds``    :1112-1225    ``_compute_candidates_loss_original`` :1278-1310
  ``_save_image``            :1312-1317    ``run``                :1323-1338

What runs where.  torch supplies device memory, streams, autograd through the
HuggingFace model (whose GEMMs are rocBLAS/hipBLASLt) and torch.distributed (RCCL).
Everything the loop itself computes goes through hand-written gfx950 kernels in
libbma_hip.so: the target-slice cross-entropy and its gradient, forbidden-token mask +
top-k, candidate scatter, the embedding gather/splice and the L-inf projection
(``ops``).  There is no fallback path: on a CPU model the engine raises.

Differences from the reference that do NOT change results:
logits only on the T target rows; shared-prefix keys/values computed once per step;
analytic chunk sizing; one batched tokenizer round trip; PNG encoding off the
critical path; candidate scoring sharded over ranks with one all-gather of losses.
Reference quirks that are kept: step = alpha*eps; second BOS in the PGD prompt; the
gradient pass uses the llava segment order and the UNSCALED embedding table for every
model; greedy acceptance of the step winner; losses rounded to the model dtype.
"""

from __future__ import annotations

import bisect
import copy
import logging
import os
import queue
import threading
import time
from typing import Dict, List, Optional, Union

import numpy as np
import torch
from torch import Tensor

from . import ops
from .config import BimodalAttackConfig, BimodalAttackResult, EngineOptions
from .dist import CandidateSharder
from .fused import DeferredNormMissed, FusedInference
from . import gemm_tuning
from .hf_adapter import HFAdapter
from .layout import expected_row_counts, ragged_plan, ragged_rows, dynamic_width, segment_order, split_at_suffix, unique_rows
from .utils import INIT_CHARS, FilterJob, get_nonascii_toks, is_oom, plan_chunk

logger = logging.getLogger("gcg")
if not logger.hasHandlers():
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s [%(filename)s:%(lineno)d] %(message)s", datefmt="%Y-%m-%d %H:%M:%S"))
    logger.addHandler(_h)
    logger.setLevel(logging.INFO)
logger.propagate = False

# Filter policy (reference :166-186 filters BEFORE it scores, :930-941).  Scoring every sampled candidate while the host runs
# the tokenizer round trip, and masking afterwards, is free as long as (almost) every candidate survives; with a tokenizer
# that rejects a share r of them, r of the scoring phase is wasted, while filtering first leaves the GPU idle for the round
# trip.  Break-even survivor rate = 1 - world * (filter seconds / scoring seconds of one GPU): 4.5 ms against 168 ms at
# search width 512 on an MI355X (profiles/r4_bench_driver.json: phase_s_per_step) -- 0.973 on one GPU, 0.79 on eight.
FILTER_COST_RATIO = float(os.environ.get("BMA_FILTER_COST_RATIO", "0.027"))
FILTER_RATE_STEPS = 4            # the survivor rate is the mean over this many steps

_OFF = ("0", "false", "False")
# Switches that exist for A/B measurements (tools/, NOTEBOOK.md), read from the environment ONCE -- engine options until
# round 4, module constants since (options are what a test or bench.py flips; tests that need one of
# these monkeypatch the constant):
# gradient pass: ask the library attention for `is_causal` instead of handing it the (1,1,S,S) mask tensor HuggingFace
# builds for inputs_embeds calls (model families with plain causal text attention only: -3 ms per pass at 643 tokens)
MASKLESS_B1_ATTENTION = os.environ.get("BMA_MASKLESS_B1_ATTENTION", "1") not in _OFF
# padded scoring (no ragged rows: Gemma-3's layout, fp32 models) runs chunks whose candidate count is a multiple of this
# -- a short last chunk is padded with copies of its last candidate, whose losses are dropped -- so a decaying search
# width (reference :919-923) meets a handful of GEMM shapes, not hundreds.  1 switches it off.
CHUNK_QUANTUM = max(1, int(os.environ.get("BMA_CHUNK_QUANTUM", "8")))
# GEMM tuning aid (tools/tune_gemms.py): in a single process, score only what rank 0 of an N-rank run would score, with
# that run's row budget -- the exact GEMM shapes of the multi-GPU run.  Results of the attack are meaningless with it.
EMULATE_WORLD = int(os.environ.get("BMA_EMULATE_WORLD", "0") or 0)
# debugging aid: host clock stamps at the loop's hand-over points (BimodalAttack.host_stamps; tools/host_stamps.py folds them) --
# where the host spends a step matters once the forward is short (W = 8: 24 ms of GPU work per step)
HOST_STAMPS = os.environ.get("BMA_HOST_STAMPS", "0") not in ("0", "false", "False", "")
SHARED_PREFIX_MIN_TOKENS = 1     # shortest prefix worth the shared-prefix attention route

TEMPLATE_PGD = "USER: <image>\n{{ messages[0]['content'][0]['text'] }} \nASSISTANT: "
TEMPLATE_GCG = "{% for message in messages %}{{ message['content'] }}{% endfor %}"


class AttackBuffer:
    """The `size` lowest-loss suffixes seen so far (reference :91-124; `size` 0 keeps only the
    latest one).  Two parallel lists kept in ascending loss order by binary insertion: a newcomer
    lands behind entries of equal loss, and when the pool is full it first displaces the current
    worst entry -- whatever its own loss, as the reference does (the loop only offers suffixes that
    beat the worst one, :618-620)."""

    def __init__(self, size: int):
        self.size = size
        self._loss: List[float] = []
        self._ids: List[Tensor] = []

    def __len__(self) -> int:
        return len(self._loss)

    def add(self, loss: float, optim_ids: Tensor) -> None:
        if self.size == 0:
            self._loss, self._ids = [loss], [optim_ids]
            return
        if len(self._loss) >= self.size:
            del self._loss[-1], self._ids[-1]
        at = bisect.bisect_right(self._loss, loss)
        self._loss.insert(at, loss)
        self._ids.insert(at, optim_ids)

    def get_best_ids(self) -> Tensor:
        return self._ids[0]

    def get_lowest_loss(self) -> float:
        return self._loss[0]

    def get_highest_loss(self) -> float:
        return self._loss[-1]

    def log_buffer(self, tokenizer) -> None:
        if not logger.isEnabledFor(logging.INFO):
            return
        shown = (tokenizer.batch_decode(ids)[0].replace("\\", "\\\\").replace("\n", "\\n") for ids in self._ids)
        logger.info("buffer:" + "".join(f"\nloss: {l} | string: {t}" for l, t in zip(self._loss, shown)))


class _PinnedRing:
    """Pinned staging blocks the engine keeps for its small host-to-device uploads.  Each slot remembers the event
    behind its last copy and is reused once that has passed -- with eight slots and a few uploads per step, always.
    (Blocks from PyTorch's pinned allocator come back only when the stream has reached the point where they were
    dropped; with the next gradient pass always queued ahead that is late, and every miss is a new pinned allocation,
    which stops the host until the stream has drained.)"""

    def __init__(self, slots: int = 8):
        self.blocks: List[Optional[Tensor]] = [None] * slots
        self.events: List[Optional[torch.cuda.Event]] = [None] * slots
        self.at = 0

    def upload(self, host: np.ndarray, device) -> Tensor:
        arr = np.ascontiguousarray(host)
        if arr.nbytes == 0:
            return torch.from_numpy(arr).to(device)
        i, self.at = self.at, (self.at + 1) % len(self.blocks)
        if self.events[i] is not None:
            self.events[i].synchronize()
        blk = self.blocks[i]
        if blk is None or blk.numel() < arr.nbytes:
            blk = self.blocks[i] = torch.empty(max(arr.nbytes, 1 << 16), dtype=torch.uint8, pin_memory=True)
        blk.numpy()[:arr.nbytes] = arr.reshape(-1).view(np.uint8)
        dev = blk[:arr.nbytes].to(device, non_blocking=True)
        self.events[i] = torch.cuda.Event()
        self.events[i].record()
        return dev.view(torch.from_numpy(arr[:0].reshape(-1)).dtype).view(arr.shape)


class _Span:
    """GPU time between two points of the current stream, read later: a phase timer that does not stop the host."""

    def __init__(self):
        self.a = torch.cuda.Event(enable_timing=True)
        self.b = torch.cuda.Event(enable_timing=True)
        self.a.record()

    def stop(self) -> "_Span":
        self.b.record()
        return self

    def seconds(self) -> float:
        self.b.synchronize()
        return self.a.elapsed_time(self.b) * 1e-3


class _PngWriter:
    """images_folder/{i}.png every step (reference :744, :1312-1317) with the same
    truncating *255 -> uint8 quantisation, done on the device; the 8-bit pixels are
    copied to the host and a worker thread does the encoding."""

    def __init__(self):
        self.q: "queue.Queue" = queue.Queue()
        self.t = threading.Thread(target=self._work, daemon=True)
        self.err: Optional[BaseException] = None
        self.t.start()

    def _work(self) -> None:
        from PIL import Image
        while True:
            item = self.q.get()
            if item is None:
                return
            arr, path = item
            try:
                Image.fromarray(arr).save(path)
            except BaseException as e:  # surfaced by close()
                self.err = e

    def submit(self, image: Tensor, path: str) -> None:
        px = (image.detach().squeeze(0) * 255).to(torch.uint8).permute(1, 2, 0).contiguous()
        self.q.put((px.cpu().numpy(), path))

    def close(self) -> None:
        self.q.put(None)
        self.t.join()
        if self.err is not None:
            raise self.err


def _transformers_version() -> str:
    try:
        import transformers
        return f"transformers {transformers.__version__}"
    except Exception:
        return "transformers ?"


class _Share:
    """This rank's share of a scoring call (BimodalAttack._share_out)."""

    def __init__(self):
        self.mine: Optional[Tensor] = None     # the candidates this rank scores (distinct ones when dealt)
        self.host_mine = self.host_par = self.inv_mine = None   # their ids / the parent's on the host; distinct -> all (the ragged plan)
        self.pick = None                       # virtual ids: which rows of `sampled` the distinct candidates of `host_mine` are
        self.dealt = None                      # (order over distinct candidates, device index candidate -> gathered slot, distinct count, first positions)
        self.want_ragged = False
        self.world = 1
        self.emulate = 0


class _Scored:
    """What the forward(s) over a share left behind (BimodalAttack._score_share)."""

    def __init__(self):
        self.losses = self.match = None
        self.log: dict = {}
        self.cache = None
        self.segs = ()


class _RunState:
    """What one run() carries from step to step (BimodalAttack._run)."""

    def __init__(self):
        self.buffer = None                     # AttackBuffer
        self.optim_ids: Optional[Tensor] = None
        self.image: Optional[Tensor] = None
        self.image_original: Optional[Tensor] = None
        self.pending = None                    # a gradient pass computed / queued for the NEXT step: (gradients, seconds, span)
        self.fuse_pgd = False
        self.trace = None
        self.writer = None
        self.losses: List[float] = []
        self.strings: List[str] = []
        self.suffixes: List[str] = []
        self.outputs: List[str] = []
        self.t_grad: List[float] = []
        self.t_samp: List[float] = []
        self.t_loss: List[float] = []
        self.t_pgd: List[float] = []
        self.t_total: List[float] = []


class _StepState:
    """What the phases of ONE step hand each other (BimodalAttack._step_*)."""

    def __init__(self, i: int):
        self.i = i
        self.st: Optional[dict] = None         # this step's trace record (EngineOptions.trace), or None
        self.span = None                       # gradient_ahead: stream events of the pass queued in the step before
        self.g_tok = self.g_img = None
        self.grad_time = self.pgd_time = self.samp_time = 0.0
        self.pgd_span = None
        self.image_synced = False
        self.flying = None                     # (gradient span, host seconds of the sampling call, PGD span) while the pass may still run
        self.virtual = None                    # early_plan: host stand-ins of this step's candidates, known before they exist
        self.sampled_all = self.job = None
        self.t0 = 0.0                          # start of the scoring phase
        self.prefetch_s = 0.0
        self.t_read = None                     # when the host had this step's outcome (gradient_ahead: the stream is still busy then)
        self.ids_host = None
        self.sampled = None
        self.n = 0
        self.current_loss = None


class BimodalAttack:
    def __init__(self, model, tokenizer, processor, config: BimodalAttackConfig, normalize=None,
                 options: Optional[EngineOptions] = None):
        self.model, self.tokenizer, self.processor = model, tokenizer, processor
        self.config, self.normalize = config, normalize
        self.opt = options or EngineOptions.from_env()
        if model.device.type != "cuda":
            raise RuntimeError(
                f"bimodalattack_amd runs on an AMD GPU only (model is on {model.device}); there is no CPU path. "
                "Move the model to the device first.")
        self.hf = HFAdapter(model, processor, normalize)
        self.embedding_layer = self.hf.embedding
        self.not_allowed_ids = None if config.allow_non_ascii else get_nonascii_toks(tokenizer, device=model.device)
        self.mask_bits = ops.build_mask_bits(self.not_allowed_ids, self.embedding_layer.num_embeddings, model.device)
        self.stop_flag = False
        self.shard = CandidateSharder(enabled=self.opt.shard)
        self.emulate_world = EMULATE_WORLD             # (tools: rank 0's share of an N-rank run in one process)
        self.score_log: Optional[list] = None          # debugging aid (tools/nan_bisect.py): one dict per scoring call when set to a list
        self._chunk_cap: Optional[int] = None      # what an OOM taught us; kept across steps
        self._grad_graph = None                    # None: not tried yet; False: eager for good
        self._prefix_cache: Dict[tuple, tuple] = {}
        # what candidate scoring computed (calls with more than one candidate): `rows` token rows went
        # through the model behind the shared prefix for `candidates` candidates; bench.py reads this
        self.score_stats = dict(candidates=0, rows=0, rows_needed=0, ragged_calls=0, padded_calls=0)
        self._rescore_graphs: Dict[tuple, object] = {}
        self._tp_checked: Optional[bool] = None         # tensor-parallel gradient pass applicable to this model / world size?
        self._tp_graph = None                           # its hipGraph (None: not tried; False: eager for good)
        self._gp = None                            # _GradPrefix: scoring prefix reused by the gradient pass (joint mode)
        self._gp_flag: Optional[bool] = None
        self._feat_graph = None                    # image -> image features (no autograd)
        self._prefix_graphs: Dict[tuple, object] = {}   # image features -> prefix keys/values
        self._match: Optional[Tensor] = None
        self.opt_b1_min = int(os.environ.get("BMA_B1_MIN_TOKENS", "2"))
        self._t_read = 0.0                         # host clock at the last packed read-back (gradient_ahead's phase books)
        self._parent_host: Optional[List[int]] = None   # the current suffix ids as the host last read them (early_plan)
        self._pins = _PinnedRing()                 # staging blocks of the small uploads (_upload)
        self._filter_pin: Optional[Tensor] = None  # the retokenisation filter's host copy of the sampled ids
        self._early_pin: Optional[Tensor] = None   # the host copy of the draws made ahead
        self._early: Optional[dict] = None         # draws of the next sampling step made ahead of its gradient pass (_draw_ahead)
        self._rb: Optional[Tensor] = None          # pinned block of the step's packed read-back (_read_later)
        self._stage: dict = {}                     # pinned staging buffer of the ragged index maps (one upload per step)
        self.graphs_captured: List[str] = []       # hipGraphs in use, by what they replay
        self.host_stamps: Optional[list] = [] if HOST_STAMPS else None     # (label, host clock) pairs, in order
        self.fallbacks: Dict[str, str] = {}        # fast path -> why it was abandoned for the slower one
        own = bool(self.opt.own_b1_kernels)               # the hand-written batch-1 kernels: one option, four process-wide switches
        ops.SKINNY_GEMM = own and ops.OWN_KERNELS["skinny_gemm"]
        ops.MID_GEMM = own and ops.OWN_KERNELS["mid_gemm"]
        ops.CAUSAL_ATTENTION = own and ops.OWN_KERNELS["causal_attention"]
        copies = bool(self.opt.derived_weight_copies)
        from .fused import FUSE_QK_ROPE
        self.fused = FusedInference(model, self.opt.fused_elementwise, copies, copies, copies, self.opt.fuse_add_norm, FUSE_QK_ROPE,
                                    own and ops.OWN_KERNELS["b1_attention"], own and ops.OWN_KERNELS["causal_attention"])
        self.tuned_gemms = gemm_tuning.enable(self.opt.gemm_tuning, model.device)
        self.fused.round_split = bool(self.tuned_gemms)      # (fused.round_cut's arithmetic is the tuned solutions' tile shape)
        self.hf.fused = self.fused
        logger.info(f"Fused forward admitted: {self.fused.admitted}; refused: {self.fused.refused}")
        # A fast path refused because the installed transformers' SOURCE no longer reads like what it was checked against (or is
        # not available) costs speed silently -- results are unchanged, strict=False keeps running: say so once, at WARNING
        # (VERDICT r5 item 14).  Refusals the model's own architecture explains (a bias, grouped heads, fp32) stay at INFO.
        drift = {k: v for k, v in self.fused.refused.items()
                 if any(t in str(v) for t in ("source not available", "statement for statement", "with nothing else touching",
                                              "not one this module patches", "not a modelling file"))}
        if drift:
            logger.warning(f"bimodalattack_amd: {len(drift)} fast path(s) NOT admitted on this model because the installed transformers "
                           f"({_transformers_version()}) does not read like the versions they were checked against -- the attack runs "
                           f"on the slower HuggingFace route for them, results unchanged: {drift}")
        if hasattr(model.config, "model_type"):
            logger.info(f"Model type: {model.config.model_type}")
        if model.dtype in (torch.float32, torch.float64):
            logger.warning(f"Model is in {model.dtype}. Use a lower precision data type for faster optimization.")
        if not getattr(tokenizer, "chat_template", None):                      # :233-249
            tpl = TEMPLATE_PGD if config.pgd_attack else TEMPLATE_GCG
            logger.warning("Tokenizer does not have a chat template. Using custom chat template for "
                           + ("GCG+PGD attack." if config.pgd_attack else "GCG only attack."))
            tokenizer.chat_template = tpl
            self.processor.chat_template = tpl

    def _fallback(self, what: str, exc: BaseException, msg: str) -> None:
        """A fast path failed on this model and a slower, equivalent one takes over.  Never silent:
        recorded in ``fallbacks`` (bench.py prints ``engine_state()``), logged, and fatal with the
        ``strict`` engine option -- which the GPU tests set for the model families the fast paths
        are written for, so a regression fails a test instead of showing up as a slower number."""
        self.fallbacks[what] = f"{type(exc).__name__}: {exc}"
        if self.opt.strict:
            raise RuntimeError(f"strict engine: {msg} ({type(exc).__name__}: {exc})") from exc
        logger.warning(f"{msg} ({type(exc).__name__}: {exc})")

    def engine_state(self) -> dict:
        """Which fast paths this attack actually ran through."""
        hf = self.hf
        return dict(tuned_gemms=self.tuned_gemms, prefix_ok=hf.prefix_ok, shared_ok=hf.shared_ok, ragged_ok=hf.ragged_ok,
                    graphs_captured=list(self.graphs_captured), fallbacks=dict(self.fallbacks),
                    fused_elementwise=bool(self.fused.enabled), fusions=dict(self.fused.admitted),
                    refused=dict(self.fused.refused), filter_first_steps=len(getattr(self, "filter_first_steps", [])),
                    chunk_cap=self._chunk_cap,
                    warmed_row_counts=list(getattr(self, "warmed", {}).get("row_counts", [])),
                    collectives=self.shard.n_collectives if self.shard.enabled else 0)

    # ------------------------------------------------------------------ setup
    def _encode(self, text, special: bool) -> Tensor:
        kw = {} if special else {"add_special_tokens": False}
        ids = self.tokenizer(text, padding=False, return_tensors="pt", **kw)["input_ids"]
        return ids.to(self.model.device, torch.int64)

    def _prepare_prompt(self, messages, target: str) -> None:
        cfg, tok = self.config, self.tokenizer
        msgs = [{"role": "user", "content": messages}] if isinstance(messages, str) else copy.deepcopy(messages)
        tail = msgs[-1]
        if isinstance(tail["content"], str) and "{optim_str}" not in tail["content"]:
            tail["content"] = tail["content"] + " {optim_str}"
        if cfg.pgd_attack:
            if isinstance(tail["content"], str):
                tail["content"] = [{"type": "text", "text": tail["content"]}, {"type": "image"}]
            elif isinstance(tail["content"], list) and not any(it.get("type") == "image" for it in tail["content"]):
                tail["content"].append({"type": "image"})
        prompt = self.processor.apply_chat_template(msgs, add_generation_prompt=True)
        if tok.bos_token and prompt.startswith(tok.bos_token):
            prompt = prompt.replace(tok.bos_token, "")
        logger.info(f"Prompt after removing BOS token: {prompt}")

        ids: Dict[str, Tensor] = {}
        if cfg.pgd_attack:
            if self.hf.is_gemma_processor:
                head, rest = prompt.split("{optim_str}", 1)
                if "<start_of_image>" not in rest:
                    raise ValueError("Expected <start_of_image> token in Gemma PGD prompt.")
                mid, marker, after = rest.partition("<start_of_image>")
                texts = dict(before_img=head.strip(), before_suffix=(mid + marker).strip(), after=after.strip())
            else:
                marker = next((m for m in ("<start_of_image>", "<image>") if m in prompt), None)
                if marker is None:
                    raise ValueError("No image token found in prompt for PGD attack")
                head, rest = prompt.split(marker, 1)
                mid, after = rest.split("{optim_str}", 1)
                texts = dict(before_img=head, before_suffix=mid, after=after)
            # both "before" pieces go through the tokenizer's default special-token
            # handling, so the PGD layout carries a second BOS mid-sequence (:346-351)
            ids["before_img"] = self._encode(texts["before_img"], True)
            ids["before_suffix"] = self._encode(texts["before_suffix"], True)
            ids["after"] = self._encode(texts["after"], False)
        else:
            head, after = prompt.split("{optim_str}")
            ids["before"] = self._encode(head, True)
            ids["after"] = self._encode(after, False)
        ids["target"] = self._encode(target, False)
        self.target_ids = ids["target"]
        self.labels = self.target_ids[0].contiguous()
        self.T = int(self.labels.numel())
        with torch.no_grad():
            self.seg: Dict[str, Tensor] = {k: self.embedding_layer(v).detach() for k, v in ids.items()}
        self.seg_ids = ids
        # the target without its last token: what the candidate forward is fed
        self.seg["target_in"] = self.seg["target"][:, :-1, :].contiguous()

    # ------------------------------------------------------------ random draws
    def _draw(self, width: int, n_opt: int):
        """The two draws of sample_ids_from_grad (:150-160), in the reference's order."""
        cfg, dev = self.config, self.model.device
        if self.opt.rng_device == "cpu":
            rnd = torch.rand((width, n_opt)).to(dev)
            rank = torch.randint(0, cfg.topk, (width, cfg.n_replace, 1)).squeeze(2).to(dev)
        else:
            rnd = torch.rand((width, n_opt), device=dev)
            rank = torch.randint(0, cfg.topk, (width, cfg.n_replace, 1), device=dev).squeeze(2)
        return rnd.contiguous(), rank.contiguous()

    def _rng_state(self, restore=None):
        """The state of the generator ``_draw`` takes from (read, or put back)."""
        on_cpu = self.opt.rng_device == "cpu"
        if restore is None:
            return torch.get_rng_state() if on_cpu else torch.cuda.get_rng_state(self.model.device)
        if on_cpu:
            torch.set_rng_state(restore)
        else:
            torch.cuda.set_rng_state(restore, self.model.device)
        return None

    # ------------------------------------------------------------ gradient pass
    def compute_gradient(self, optim_ids: Tensor, image: Optional[Tensor] = None, tokens_only: bool = False):
        """d(mean target CE)/d(one-hot suffix) and /d(image) (:953-1028).  Shapes never
        change during an attack, so the whole forward+backward is captured into a hipGraph
        on the first call (after one eager run) and replayed afterwards; results are the
        eager ones (same kernels, same order)."""
        if self._tp_active():
            # the pass cut over the ranks, its all-reduces included, CAN be one hipGraph too (RCCL collectives are
            # capturable: the process group's stream fork/join lands in the graph) -- opt-in (tp_gradient="graph") until a
            # run on two or more GPUs has been recorded; a gloo rehearsal's collectives run on the host and cannot be captured
            if self._tp_graph is None and not (self.opt.tp_gradient == "graph" and self.opt.graph_gradient and self.shard.backend() == "nccl"):
                self._tp_graph = False
            if self._tp_graph is False:
                return self._gradient_tp(optim_ids, image)
            if self._tp_graph is None:
                graph, err = None, None
                try:
                    graph = _GradientGraph(self, optim_ids, image, fn=self._gradient_tp)
                except Exception as e:
                    err = e
                    torch.cuda.synchronize(self.model.device)
                # EVERY rank replays the graph or NONE does: a rank that replays while another runs eager collectives
                # would pair a captured all-reduce with an eager one (and the ranks would no longer launch the same
                # kernels in the same order, which is what keeps their gradients bit-identical) -- so the outcome of
                # the capture is agreed on: MIN over the ranks of a success flag
                if not self.shard.all_ok(graph is not None, self.model.device):
                    graph = None
                    self._tp_graph = False
                    self._fallback("graph_gradient_tp", err or RuntimeError("another rank could not capture it"),
                                   "tensor-parallel gradient pass not captured into a graph on every rank; all ranks run it eagerly")
                    return self._gradient_tp(optim_ids, image)
                self._tp_graph = graph
                self.graphs_captured.append("gradient_tp")
            return self._tp_graph(optim_ids, image)
        if image is not None and self._gp_enabled():
            try:
                if self._gp is None:
                    self._gp = _GradPrefix(self)
                return self._gp.gradient(optim_ids, image, tokens_only)
            except Exception as e:            # not workable with this model: the full pass below
                self._fallback("grad_prefix_reuse", e, "gradient pass does not reuse the scoring prefix; running the full pass")
                self._gp = False
                torch.cuda.synchronize(self.model.device)
        if not self.opt.graph_gradient or self._grad_graph is False:
            return self._gradient_eager(optim_ids, image)
        if self._grad_graph is None:           # first call of the attack: warm up eagerly, capture, replay
            try:
                self._grad_graph = _GradientGraph(self, optim_ids, image)
                self.graphs_captured.append("gradient")
            except Exception as e:            # not capturable with this model: stay eager
                self._fallback("graph_gradient", e, "gradient pass not captured into a graph; running eager")
                self._grad_graph = False
                torch.cuda.synchronize(self.model.device)
                return self._gradient_eager(optim_ids, image)
        return self._grad_graph(optim_ids, image)

    _GRAD_ORDER = ["before_img", "image", "before_suffix", "optim", "after", "target"]

    def _tp_active(self) -> bool:
        if not (self.opt.tp_gradient and self.shard.enabled):
            return False
        if self._tp_checked is None:
            self._tp_checked = bool(self.fused.tp_ok(self.shard.world))
            if not self._tp_checked:
                self._fallback("tp_gradient", RuntimeError("projection widths / head counts do not divide over the ranks, "
                               "or the decoder layers' structure is not one the fused forward restates"),
                               "tensor-parallel gradient pass not applicable; every rank runs the whole pass")
        return self._tp_checked

    def _gradient_tp(self, optim_ids: Tensor, image: Optional[Tensor] = None):
        """The gradient pass cut over the ranks (EngineOptions.tp_gradient): every rank runs the replicated parts
        (embeddings, norms, residual stream, vision tower, head, cross-entropy) and ITS rows / columns of the decoder
        projections; the all-reduces inside leave identical token and pixel gradients on every rank."""
        self.fused.tp = (self.shard.rank, self.shard.world, self.shard.group)
        try:
            return self._gradient_eager(optim_ids, image)
        finally:
            self.fused.tp = None

    def _gp_enabled(self) -> bool:
        """PGD + GCG with the image in FRONT of the suffix (LLaVA layout).  Joint mode: the prefix pass candidate
        scoring runs on the image a PGD step has just produced -- vision tower + prompt up to the suffix, batch 1 --
        is the first 599 of the 644 rows of the NEXT step's gradient pass; run with autograd it serves both
        (``_GradPrefix``) and the gradient pass runs 44 rows forward instead of 644.  Without joint_eval the step has
        two gradient passes on the same image (:481): the second only feeds the token gradient, which does not reach
        the prefix at all -- 44 rows forward and backward against the prefix's detached keys/values -- and the first
        pass of the next step back-propagates through the history that prefix pass kept.  Needs the gradient pass
        and the scoring call to be the same function of the same segments (:968, :981-991 against :1142,
        :1150-1163: true for the llava order and an unscaled embedding table)."""
        if self._gp is False or (self.opt.tp_gradient and self.shard.enabled):
            return False
        if self._gp_flag is None:
            cfg, hf, opt = self.config, self.hf, self.opt
            ok = bool(opt.grad_prefix_reuse and cfg.pgd_attack and cfg.gcg_attack and opt.prefix_reuse
                      and opt.shared_prefix_attention and opt.target_rows_only and hf.emb_scale == 1.0
                      and hf.shared_ok is not False and hf.prefix_ok is not False
                      and segment_order("pgd", hf.model_type, single=True) == self._GRAD_ORDER)
            self._gp_flag = bool(ok and hf.shared_prefix_configs())
        return self._gp_flag

    def _gradient_eager(self, optim_ids: Tensor, image: Optional[Tensor] = None):
        """One forward/backward at batch 1.  The one-hot is never built: its gradient is
        (dL/d suffix embeddings) @ E^T, the same product autograd would form."""
        cfg = self.config
        E = self.embedding_layer.weight
        emb = E[optim_ids[0]].unsqueeze(0).detach()
        if cfg.gcg_attack:
            emb.requires_grad_()
        if cfg.pgd_attack:
            feats = self.image_features(image)
            parts = [self.seg["before_img"], feats.to(emb.dtype), self.seg["before_suffix"], emb, self.seg["after"],
                     self.seg["target"]]
        else:
            parts = [self.seg["before"], emb, self.seg["after"], self.seg["target"]]
        x = torch.cat(parts, dim=1)
        with self.fused, self._b1_attention(x.shape[1]):
            if self.opt.target_rows_only:
                logits = self.hf.target_logits(x[:, :-1], self.T, rows_only=True)
            else:
                logits = self.hf.target_logits(x, self.T, rows_only=False)
        loss = ops.TargetCrossEntropy.apply(logits[0], self.labels)
        wanted = ([emb] if cfg.gcg_attack else []) + ([image] if cfg.pgd_attack else [])
        grads = list(torch.autograd.grad(loss, wanted))
        g_tok = None
        if cfg.gcg_attack:
            g_emb = grads.pop(0)[0]                       # (n_opt, D), model dtype
            with torch.no_grad():
                g_tok = (g_emb @ E.t()).unsqueeze(0)      # (1, n_opt, V), model dtype
        g_img = grads.pop(0) if cfg.pgd_attack else None
        return g_tok, g_img, loss.detach()

    def gradient_pass_eager(self, optim_ids: Tensor, image: Optional[Tensor] = None):
        """What ONE gradient pass of this attack computes, run eagerly on fresh leaves -- the work the hipGraphs replay,
        without the graphs (measurement only: bench.py records the library GEMMs of this call; a replay runs no Python
        and the in-library event brackets cannot see inside it)."""
        img = None if image is None else image.detach().clone().requires_grad_()
        if img is not None and self._gp_enabled() and self._gp not in (None, False):
            gp = _GradPrefix(self)                 # scoring prefix with history + the tail forward + the backward through both
            gp.graphs = False
            gp.image, gp.ids = img, optim_ids
            gp._set(*gp._prefix_fn())
            return gp._tail_fn()
        return self._gradient_eager(optim_ids, img)

    def _b1_attention(self, seq_len: int):
        """Context for the batch-1 gradient pass: mask-free causal attention for the text layers of the model
        families the shared-prefix scheme knows (plain causal attention; a sliding window at least as long as
        the sequence), nothing otherwise."""
        import contextlib
        cfgs = self.hf.shared_prefix_configs(seq_len) if (MASKLESS_B1_ATTENTION and seq_len >= self.opt_b1_min) else []
        if not cfgs:
            return contextlib.nullcontext()
        from . import prefix_attention as pa
        return pa.causal_b1(cfgs)

    def perform_pgd_step(self, image: Tensor, eps: float, alpha: float, image_grad: Tensor,
                         image_original: Tensor) -> Tensor:
        """x <- clamp(clamp(x - alpha*eps*sign(g), x0-eps, x0+eps), 0, 1) (:1030-1037)."""
        x = image.detach()
        if x.dtype != torch.float32:
            raise TypeError("the PGD image must be float32 (the reference feeds ToTensor() output)")
        out = ops.linf_step(x.contiguous(), image_grad.to(torch.float32).contiguous(),
                            image_original.detach().contiguous(), eps, alpha)
        return out.requires_grad_()

    # ------------------------------------------------------------ sampling
    def _width(self, step: int) -> int:
        cfg = self.config
        if self.opt.width_override is not None:
            return int(self.opt.width_override(step))
        return dynamic_width(step, cfg.search_width, cfg.num_steps, cfg.min_search_width, cfg.dynamic_search)

    def _draw_ahead(self, step: int, n_opt: int) -> None:
        """early_plan: the random part of step `step`'s sampling (:150-160) one phase early -- positions and top-k
        ranks do not depend on the gradient -- with a copy on its way to the host, in FRONT of the gradient pass the
        caller queues next.  Several GPUs: rank 0's draws overwrite everybody's, as its candidates will."""
        rnd, rank = self._draw(self._width(step), n_opt)
        pos = ops.rand_positions(rnd, self.config.n_replace)
        self.shard.sync_state(pos, rank)
        both = torch.stack([pos, rank])
        if self._early_pin is None or self._early_pin.numel() < both.numel():
            self._early_pin = torch.empty(max(both.numel(), 4096), dtype=both.dtype, pin_memory=True)
        host = self._early_pin[:both.numel()].view(both.shape)
        host.copy_(both, non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        self._early = dict(step=step, pos=pos, rank=rank, host=host, event=done)

    @staticmethod
    def _virtual_ids(early: dict, parent: List[int]):
        """Host stand-ins for the candidates sampled from `parent` with these draws: the parent with -1 - rank at each
        drawn position.  Two stand-ins are equal exactly when the draws are, and then the candidates are; a stand-in
        differs from the parent first at its smallest drawn position, and the candidate no earlier (it may not differ
        at all -- a top-k token can be the one already there -- which costs rows, not correctness)."""
        early["event"].synchronize()
        pos, rank = early["host"][0].numpy(), early["host"][1].numpy()
        par = np.asarray(parent, dtype=np.int64)
        fake = np.repeat(par[None, :], pos.shape[0], axis=0)
        np.put_along_axis(fake, pos, -1 - rank, axis=1)
        return fake, par

    def _stamp(self, label: str) -> None:
        if self.host_stamps is not None:
            self.host_stamps.append((label, time.perf_counter()))

    def candidate_sampling(self, step: int, optim_ids: Tensor, g_tok: Optional[Tensor], image: Optional[Tensor] = None):
        """mask -> top-k -> random position/rank -> scatter (:130-163).  Returns every sampled
        candidate and a FilterJob: the retokenisation filter (:166-186) runs on the host while
        the GPU scores, and is applied to the losses afterwards.  On several GPUs rank 0's
        candidates -- and its PGD image, when there is one -- overwrite every rank's here, in one
        broadcast (the shapes are a function of the step number, so nothing else is exchanged)."""
        cfg = self.config
        if not cfg.gcg_attack:
            if image is not None:
                self.shard.sync_state(image)
            return optim_ids, FilterJob(optim_ids, self.tokenizer, False)
        ids = optim_ids[0].contiguous()
        early, self._early = self._early, None
        if early is not None and early["step"] == step:       # drawn one phase early (_draw_ahead): same generator, same order
            pos, rank = early["pos"], early["rank"]
        else:
            rnd, rank = self._draw(self._width(step), ids.numel())
            pos = ops.rand_positions(rnd, cfg.n_replace)
        topk_idx = ops.mask_topk(g_tok[0], self.mask_bits, cfg.topk)
        sampled = ops.sample_scatter(ids, topk_idx, pos, rank)
        self.shard.sync_state(*([sampled] if image is None else [sampled, image]))
        self._last = dict(topk_idx=topk_idx, pos=pos, rank=rank, sampled=sampled)
        if cfg.filter_ids and (self._filter_pin is None or self._filter_pin.numel() < sampled.numel()):
            self._filter_pin = torch.empty(max(sampled.numel(), cfg.search_width * sampled.shape[1]), dtype=sampled.dtype,
                                           pin_memory=True)
        return sampled, FilterJob(sampled, self.tokenizer, cfg.filter_ids, pinned=self._filter_pin)

    # ------------------------------------------------------------ filter policy
    def _filter_first_now(self) -> bool:
        """Run the retokenisation filter BEFORE scoring at this step?  ``EngineOptions.filter_first``: True / False force
        it; None (default) decides from the survivor rate of the last FILTER_RATE_STEPS steps -- a quantity every rank
        computes alike (all ranks filter the same ids), so sharded runs stay in step without a collective."""
        cfg, mode = self.config, self.opt.filter_first
        if not (cfg.gcg_attack and cfg.filter_ids):
            return False
        if mode is not None:
            return bool(mode)
        recent = self._keep_rates[-FILTER_RATE_STEPS:]
        if not recent:
            return False
        world = self.emulate_world if (self.emulate_world > 1 and not self.shard.enabled) else self.shard.world
        return sum(recent) / len(recent) < 1.0 - world * FILTER_COST_RATIO

    def _filter_now(self, sampled: Tensor, job: FilterJob):
        """The reference's order (:930-941, :166-186): the round trip runs NOW -- the host waits for the sampled ids --
        and only the survivors go on to be scored.  Returns (survivors, a job that has nothing left to drop, the round
        trip's own seconds)."""
        keep = job.result()
        self._keep_rates.append(len(keep) / max(1, sampled.shape[0]))
        if len(keep) < sampled.shape[0]:
            sampled = sampled.index_select(0, self._upload(np.asarray(keep, dtype=np.int64)))
        return sampled, _AlreadyFiltered(sampled.shape[0]), job.seconds

    # ------------------------------------------------------------ scoring
    def _segments(self, order, feats):
        out = []
        for name in order:
            if name == "optim":
                out.append(("gather", None))
            elif name == "image":
                out.append(("shared", feats))
            else:
                out.append(("shared", self.seg[name]))
        return out

    def score_candidates(self, sampled: Tensor, order: List[str], feats: Optional[Tensor],
                         allow_prefix: bool = True, parent: Optional[Tensor] = None, virtual=None) -> Tensor:
        """Per-candidate mean target CE (:1278-1310) of this rank's slice of `sampled`,
        all-gathered to the full vector.  `order` ends in "target".  `parent` (1,n_opt): the ids
        the candidates were sampled from, when the caller knows them (enables ragged scoring).
        `virtual`: (stand-in ids (n,n_opt), parent ids) on the host (``_virtual_ids``) -- the ragged plan is made
        from them and `sampled` is not waited for."""
        try:
            with self.fused:
                return self._score_candidates(sampled, order, feats, allow_prefix, parent, virtual)
        except DeferredNormMissed as e:
            # (fused.py has switched the deferral off: the same call again runs the head norms as their own launches)
            self._fallback("fuse_qk_rope", e, "q/k norm inside the rotary launch not applicable to this model's attention blocks")
            with self.fused:
                return self._score_candidates(sampled, order, feats, allow_prefix, parent, virtual)

    def image_features(self, image: Tensor) -> Tensor:
        """``model.get_image_features`` on the normalised image (reference :526-536, :877-884, :970-979), inside the fused
        context: the projector's RMSNorm (Gemma-3's ``mm_soft_emb_norm``, on a transposed view of the pooled patches)
        then runs ``bma_rmsnorm`` instead of ATen's strided-mean reduction.  That reduction is a multi-block one with
        a scratch buffer and semaphores, and replayed from a hipGraph on this stack it returned NaN rows for some images
        (rows 64-255 of 256, the eager call on the same input finite: tools/nan_bisect.py --probe-feats; rounds 2 and
        3 shipped a Gemma-3 workload with NaN losses because of it)."""
        with self.fused:
            return self.hf.image_features(image)

    def scoring_features(self, image: Tensor) -> Tensor:
        """Image features for scoring (no autograd): the vision tower at batch 1 is launch-bound,
        so it is replayed from a hipGraph after the first call."""
        if self._gp_enabled() and self._gp is not None:
            try:
                return self._gp.features(image)
            except Exception as e:
                self._fallback("grad_prefix_reuse", e, "gradient pass does not reuse the scoring prefix; running the full pass")
                self._gp = False
                torch.cuda.synchronize(self.model.device)
        if not self.opt.graph_scoring or self._feat_graph is False:
            return self.image_features(image)
        if self._feat_graph is None:
            try:
                self._feat_graph = _ReplayGraph(self.model.device, self.image_features, image)
                self.graphs_captured.append("image_features")
            except Exception as e:
                self._fallback("graph_features", e, "image features not captured into a graph; running eager")
                self._feat_graph = False
                torch.cuda.synchronize(self.model.device)
                return self.image_features(image)
        return self._feat_graph(image)

    def _wants_shared(self, P: int, total_len: int = 0) -> bool:
        # same-process A/B on MI355X, 512 candidates: P=21: 231 ms vs 254 ms through the HF cache
        # (KV concat); P=599: 288 ms vs 714 ms
        hf = self.hf
        return bool(self.opt.shared_prefix_attention and hf.shared_ok is not False
                    and P >= SHARED_PREFIX_MIN_TOKENS and hf.shared_prefix_configs(total_len))

    def _prefix(self, prefix_names: List[str], feats: Optional[Tensor], total_len: int = 0):
        """Keys/values of the segments in front of the suffix.  They depend on nothing but
        the prompt -- and on the image when it is part of the prefix -- so a text-only prefix
        is computed once per attack and an image prefix once per call."""
        hf = self.hf
        key = tuple(prefix_names)
        if "image" not in prefix_names and key in self._prefix_cache:
            return self._prefix_cache[key]
        gp = self._gp
        # ... from the gradient pass's recorded prefix only while the shared-prefix attention will consume it (a
        # RecordingKV is not an HF cache: the KV-concat route cannot expand it)
        if gp not in (None, False) and feats is not None and gp.serves(key, feats) and self._wants_shared(gp.P, total_len):
            hf.prefix_ok = True
            return gp.cache(), gp.P
        def cat_prefix(f):
            return torch.cat([f if p == "image" else self.seg[p] for p in prefix_names], dim=1)

        P = sum((feats.shape[1] if p == "image" else self.seg[p].shape[1]) for p in prefix_names)
        cache = None
        if P > 0:
            try:
                if self._wants_shared(P, total_len):
                    try:
                        cache = self._recorded_prefix(key, cat_prefix, feats)
                    except DeferredNormMissed:
                        raise                          # (only the norm deferral is at fault: score_candidates retries without it)
                    except Exception as e:
                        self._fallback("recorded_prefix", e, "recording prefix pass failed; using the HF cache")
                if cache is None:
                    cache = hf.build_prefix(cat_prefix(feats))
            except DeferredNormMissed:
                raise
            except Exception as e:  # a model without cache support: remember, use the full sequence
                self._fallback("prefix_reuse", e, "prefix reuse disabled")
            hf.prefix_ok = cache is not None
        out = (cache, P)
        if "image" not in prefix_names:
            self._prefix_cache[key] = out
        return out

    def _recorded_prefix(self, key: tuple, cat_prefix, feats: Optional[Tensor]):
        """Prefix keys/values through a RecordingKV.  With an image in the prefix this runs every
        step at batch 1 (launch-bound): captured into a hipGraph keyed by the layout."""
        hf = self.hf

        def build(f):
            x = cat_prefix(f)
            with self._b1_attention(x.shape[1]):          # one sequence, nothing cached in front of it: plain causal
                return hf.build_prefix_recording(x)

        if feats is None or not self.opt.graph_scoring or self._prefix_graphs.get(key) is False:
            return build(feats)
        g = self._prefix_graphs.get(key)
        if g is None:
            try:
                with self.fused:
                    g = _ReplayGraph(self.model.device, build, feats)
            except Exception as e:
                self._fallback("graph_prefix", e, "prefix pass not captured into a graph; running eager")
                self._prefix_graphs[key] = False
                torch.cuda.synchronize(self.model.device)
                return build(feats)
            self._prefix_graphs[key] = g
            self.graphs_captured.append("prefix:" + "|".join(key))
        return g(feats)

    def _read_later(self, packed: Tensor):
        """Device vector -> host list without stopping the stream: the copy lands in a pinned block, an event marks
        it, and the returned callable waits for THAT event only -- work queued behind it keeps the GPU busy."""
        n = packed.numel()
        if self._rb is None or self._rb.numel() < n or self._rb.dtype != packed.dtype:
            self._rb = torch.empty(max(n, 64), dtype=packed.dtype).pin_memory()
        dst = self._rb[:n]
        dst.copy_(packed, non_blocking=True)
        done = torch.cuda.Event()
        done.record()

        def read() -> list:
            done.synchronize()
            return dst.tolist()
        return read

    def _upload(self, host: np.ndarray) -> Tensor:
        """Host array -> device without holding the host: through a pinned block of the engine's own ring
        (``_PinnedRing``) and a non-blocking copy.  A `.to(device)` from pageable memory waits for the stream on ROCm."""
        return self._pins.upload(host, self.model.device)

    def _ragged_score(self, host_ids: np.ndarray, host_parent: np.ndarray, segs, L: int, P: int, cache,
                      n_rows: Optional[int] = None, inverse: Optional[np.ndarray] = None, real=None):
        """(loss (m_out,) fp32, early-stop hit or None) of this rank's candidates through the ragged forward, or None
        when this draw does not fit the row count asked for (then the caller scores the padded block).  host_ids: this
        rank's DISTINCT candidates (host copy); inverse: which of them each candidate to report is (None: one each, in
        order).  (Round 3 also kept the whole forward of a row count as one hipGraph; measured to buy nothing -- the
        forward is GPU-bound even at an eighth of the rows -- and removed in round 4: NOTEBOOK.md r3 7.)"""
        hf, dev, cfg = self.hf, self.model.device, self.config
        from .prefix_attention import RaggedMaps, fused_ragged_route
        # the same predicate the attention function evaluates on the tensors: the library route needs the padded-block
        # maps (BMA_FUSED_RAGGED_ATTENTION=0, fp32 models, head sizes the kernel does not take), the kernel route not
        fused = fused_ragged_route(self.model.dtype, hf.head_dim, hf.heads, hf.kv_heads, L)
        self._stamp("dealt")
        plan = ragged_plan(host_ids, host_parent, L, self.T, P, n_rows, dedup=False, padded_maps=not fused,
                           inverse=inverse)
        self._stamp("planned")
        if plan is None:
            return None
        mu = int(plan["m"])           # distinct candidates, in the plan's order: duplicates are computed once
        st = self.score_stats
        st["ragged_calls"] += 1
        st["rows"] += int(plan["N"])
        st["rows_needed"] += int(plan["needed"])
        m_out = int(plan.get("m_out", mu))
        ids = np.concatenate([plan["cand"], host_parent.reshape(1, -1)])
        E = self.embedding_layer.weight

        def forward(maps, blocks):
            # the row list straight from the segments and the table: the padded (blocks, L, D) block is never built
            rows = ops.splice(segs, blocks, E, maps.ids, hf.emb_scale, rows=maps.flat).unsqueeze(0)
            logits = hf.target_logits_ragged(rows, self.T, cache, maps)
            loss, hit, _, _ = ops.ce_target(logits, self.labels, want_match=cfg.early_stop)
            return loss, hit

        if real is not None:
            # host_ids are stand-ins (``_virtual_ids``): the maps go up without ids, the forward reads the device's
            maps = RaggedMaps(plan, dev, ids=None, stage=self._stage)
            maps.ids = torch.cat([real[0], real[1]], dim=0)
            if maps.ids.shape != (mu + 1, int(plan["n_opt"])):
                raise RuntimeError("ragged scoring: gathered ids do not match the plan")
            self._stamp("maps_up")
            return forward(maps, mu + 1)
        maps = RaggedMaps(plan, dev, ids=ids, stage=self._stage)
        return forward(maps, mu + 1)

    @staticmethod
    def _dealt_rows(dealt, world: int, L: int, n_opt: int) -> int:
        by_cost, _, _, first = dealt
        need = max(n_opt + int((L - first[by_cost[r::world]]).sum()) for r in range(world))
        cap = n_opt + len(by_cost[0::world]) * L
        return ragged_rows(need, cap)

    def _score_candidates(self, sampled: Tensor, order: List[str], feats: Optional[Tensor],
                          allow_prefix: bool = True, parent: Optional[Tensor] = None, virtual=None) -> Tensor:
        """Losses of all n sampled candidates, in order (:1278-1310): `_share_out` decides which of them this rank scores and
        plans the ragged forward from one dedup, `_score_share` runs the forward(s) over that share (ragged rows, padded chunks
        behind a shared prefix, or the HF-cache route; OOM halving), `_collect` puts every rank's losses back in the
        candidates' order."""
        sh = self._share_out(sampled, allow_prefix, parent, virtual)
        sc = self._score_share(sh, sampled, order, feats, allow_prefix, parent)
        return self._collect(sh, sc, sampled, feats)

    def _share_out(self, sampled: Tensor, allow_prefix: bool, parent: Optional[Tensor], virtual) -> "_Share":
        """Which candidates this rank scores (`mine`), with the host-side facts the ragged plan needs -- from ONE copy of the
        ids to the host and ONE exact dedup per step (or from the virtual ids planned while the gradient pass ran)."""
        hf = self.hf
        n = sampled.shape[0]
        sh = _Share()
        sh.emulate = emulate = self.emulate_world if (self.emulate_world > 1 and not self.shard.enabled) else 0
        sh.world = world = emulate or self.shard.world
        # `plan_ok` depends on options and the model family only -- never on what one rank learnt at run time --
        # because it also decides HOW candidates are partitioned over ranks, which every rank must decide alike
        plan_ok = bool(parent is not None and allow_prefix and self.opt.ragged_suffix and n > 1 and self.opt.prefix_reuse
                       and self.opt.target_rows_only and self.opt.shared_prefix_attention and hf.shared_prefix_configs())
        sh.want_ragged = bool(plan_ok and hf.ragged_ok is not False and hf.shared_ok is not False)
        if plan_ok and virtual is not None:
            # planned while the gradient pass ran, from the draws alone: nothing here waits for the stream
            host_all, sh.host_par = virtual
            uniq, inv, first_at = unique_rows(host_all, return_first=True)
        elif plan_ok:
            # ONE device-to-host copy (ids + parent) and ONE exact dedup per step feed both the partition over
            # ranks and the ragged plan
            both_h = torch.cat([sampled, parent.reshape(1, -1).to(sampled.device)], dim=0).cpu().numpy()
            host_all, sh.host_par = both_h[:n], both_h[n]
            uniq, inv = unique_rows(host_all)
            first_at = None
        if plan_ok and (self.shard.enabled or emulate) and n > world:
            # Ragged scoring on several GPUs: every rank sees the same ids, so each can drop the
            # duplicates, sort the distinct candidates by first replaced position and take every
            # world-th one -- all ranks then compute (almost) the same number of rows, and the fixed
            # per-rank row budget is the global one divided by the world size instead of a
            # small-sample budget with its own safety margin.
            diff = uniq != sh.host_par[None, :]
            first = np.where(diff.any(1), diff.argmax(1), uniq.shape[1] - 1)
            by_cost = np.argsort(first, kind="stable")
            take = by_cost[0::world] if emulate else self.shard.deal(by_cost)
            sh.host_mine = np.ascontiguousarray(uniq[take])
            if first_at is None:
                sh.mine = self._upload(sh.host_mine)
            else:
                sh.pick = np.ascontiguousarray(first_at[take])
                sh.mine = sampled.index_select(0, self._upload(sh.pick))
            # where each of the n candidates' loss will sit in the gathered buffer: uploaded NOW, while the stream
            # is idle -- behind the forward the same pageable copy would hold the host until the GPU had finished,
            # and the retokenisation filter would run after the forward instead of beside it (dist.dealt_index)
            if emulate:
                slot = np.full((uniq.shape[0],), take.shape[0], dtype=np.int64)      # unscored: the padding slot
                slot[take] = np.arange(take.shape[0])
                sel = slot[inv]
            else:
                sel = self.shard.dealt_index(by_cost, inv)
            sh.dealt = (by_cost, self._upload(sel), uniq.shape[0], first)
        else:
            lo, hi = self.shard.bounds(n)
            sh.mine = sampled[lo:hi].contiguous()
            if plan_ok:
                if (lo, hi) == (0, n):
                    sh.host_mine, sh.inv_mine, sh.pick = uniq, inv, first_at
                elif first_at is None:
                    sh.host_mine, sh.inv_mine = unique_rows(host_all[lo:hi])
                else:
                    sh.host_mine, sh.inv_mine, sh.pick = unique_rows(host_all[lo:hi], return_first=True)
        return sh

    def _score_share(self, sh: "_Share", sampled: Tensor, order: List[str], feats: Optional[Tensor], allow_prefix: bool,
                     parent: Optional[Tensor]) -> "_Scored":
        """The forward(s) over this rank's candidates: one ragged forward when the plan allows it, else padded chunks."""
        cfg, hf = self.config, self.hf
        mine, dealt, pick = sh.mine, sh.dealt, sh.pick
        m = mine.shape[0]
        E = self.embedding_layer.weight
        feats = None if feats is None else feats.to(E.dtype)
        rows_only = self.opt.target_rows_only
        prefix_names, tail_names = split_at_suffix(order)
        use_prefix = bool(allow_prefix and self.opt.prefix_reuse and rows_only and prefix_names
                          and hf.prefix_ok is not False and m > 0)

        cache, P = None, 0
        total_len = sum((mine.shape[1] if nm == "optim" else (feats.shape[1] if nm == "image" else self.seg[nm].shape[1]))
                        for nm in order)
        if use_prefix:
            cache, P = self._prefix(prefix_names, feats, total_len)
            use_prefix = cache is not None
        shared = bool(use_prefix and self._wants_shared(P, total_len))
        names = tail_names if use_prefix else list(order)
        names = [("target_in" if (t == "target" and rows_only) else t) for t in names]
        segs = self._segments(names, feats)
        L = sum((mine.shape[1] if k == "gather" else t.shape[-2]) for k, t in segs)

        free = torch.cuda.mem_get_info(self.model.device)[0] if m > 1 else (1 << 40)   # one candidate always fits
        fixed = cfg.batch_size if cfg.batch_size is not None else self.opt.chunk
        quantum = CHUNK_QUANTUM if (fixed is None and m > CHUNK_QUANTUM > 1) else 1
        chunk = plan_chunk(max(m, 1), L, P if (use_prefix and not shared) else 0, hf.kv_bytes_per_token,
                           hf.act_bytes_per_token, free, fixed, quantum=quantum)
        if self._chunk_cap is not None:
            chunk = min(chunk, self._chunk_cap)

        losses = torch.empty(m, dtype=torch.float32, device=self.model.device)
        match = torch.zeros(m, dtype=torch.float32, device=self.model.device) if cfg.early_stop else None
        ragged = bool(shared and sh.want_ragged and sh.host_mine is not None
                      and chunk >= m > 1 and tail_names[0] == "optim" and L - self.T >= mine.shape[1] - 1)
        s = 0
        while s < m:
            b = min(chunk, m - s)
            try:
                kv, x, logits, scored = None, None, None, None
                if ragged:
                    try:
                        n_rows = None
                        if dealt is not None:
                            # every rank builds the row count of the rank with the most rows (they differ by
                            # a few rows after dealing): one set of GEMM shapes per step on all ranks
                            n_rows = self._dealt_rows(dealt, sh.world, L, mine.shape[1])
                        real = None
                        if pick is not None:
                            # the stand-ins planned it; the forward embeds the real ids, gathered on the device
                            # (dealt: `mine` IS that gather; else the first appearances within this rank's slice)
                            real = (mine if dealt is not None else mine.index_select(0, self._upload(pick)),
                                    parent.reshape(1, -1).to(sampled.device))
                        scored = self._ragged_score(sh.host_mine, sh.host_par, segs, L, P, cache, n_rows, sh.inv_mine, real)
                        hf.ragged_ok = True
                    except DeferredNormMissed:
                        raise                          # (a missed deferral costs the deferral, not the ragged path)
                    except Exception as e:
                        if hf.ragged_ok or is_oom(e):
                            raise
                        self._fallback("ragged_suffix", e, "ragged scoring disabled")
                        hf.ragged_ok, ragged = False, False
                if m > 1:
                    self.score_stats["candidates"] += b
                if logits is None and scored is None:
                    ids_b = mine[s:s + b]
                    pad = (-b) % quantum if not self._chunk_cap else 0
                    if pad:                      # a short last chunk: up to the next multiple, with copies of its last candidate
                        ids_b = torch.cat([ids_b, ids_b[-1:].expand(pad, -1)], dim=0)
                    x = ops.splice(segs, b + pad, E, ids_b.contiguous(), hf.emb_scale)
                    if m > 1:
                        st = self.score_stats
                        st["padded_calls"] += 1
                        st["rows"] += (b + pad) * L
                        st["rows_needed"] += b * L
                if logits is not None or scored is not None:
                    pass
                elif shared:
                    try:
                        logits = hf.target_logits_shared_prefix(x, self.T, cache)
                        hf.shared_ok = True
                    except DeferredNormMissed:
                        raise
                    except Exception as e:
                        if hf.shared_ok or is_oom(e):
                            raise
                        self._fallback("shared_prefix_attention", e, "shared-prefix attention disabled")
                        hf.shared_ok, shared = False, False
                        self._gp, self._gp_flag = False, False      # its recorded prefix only serves the shared-prefix route
                        self._prefix_cache.clear()                  # rebuild the prefix as an HF cache
                        cache, P = self._prefix(prefix_names, feats, total_len)
                        use_prefix = cache is not None
                        if not use_prefix:
                            raise
                if logits is None and scored is None:
                    # the padded chunk's row count, not b: a short last chunk carries copies of its last candidate
                    kv = hf.expand_prefix(cache, x.shape[0]) if use_prefix else None
                    logits = hf.target_logits(x, self.T, rows_only=rows_only, cache=kv)
                if scored is not None:
                    loss, hit = scored
                else:
                    loss, hit, _, _ = ops.ce_target(logits, self.labels, want_match=cfg.early_stop)
                if self.score_log is not None and logits is not None:
                    self.score_log.append(dict(chunk_at=s, b=b, logits_bad=(~torch.isfinite(logits.float())).sum(),
                                                   x_bad=None if x is None else (~torch.isfinite(x.float())).sum()))
                losses[s:s + b] = loss[:b]       # (the padding's losses are dropped)
                if match is not None:
                    match[s:s + b] = hit[:b].to(torch.float32)
                del x, kv, logits
                s += b
            except Exception as e:
                if not is_oom(e) or chunk == 1:
                    raise
                chunk = max(1, chunk // 2)
                self._chunk_cap = chunk
                ragged = False        # the ragged forward scores all m at once: retry through padded chunks
                logger.warning(f"Decreasing batch size to: {chunk}")
                torch.cuda.empty_cache()
        sc = _Scored()
        sc.losses, sc.match = losses, match
        sc.log = dict(m=m, L=L, P=P, chunk=chunk, ragged=bool(ragged), shared=bool(shared), use_prefix=bool(use_prefix))
        sc.cache, sc.segs = cache, segs
        return sc

    def _collect(self, sh: "_Share", sc: "_Scored", sampled: Tensor, feats: Optional[Tensor]) -> Tensor:
        """Every rank's losses (and early-stop hits, left in self._match) back in the order of the n candidates."""
        n = sampled.shape[0]
        losses, match = sc.losses, sc.match
        if sh.dealt is not None:
            by_cost, sel_t, n_u, _ = sh.dealt
            if sh.emulate:     # GEMM tuning only: rank 0's shapes of an `emulate`-rank run, the other ranks' shares unscored
                full = torch.cat([losses.to(torch.float32), losses.new_full((1,), float("inf"), dtype=torch.float32)])[sel_t]
                self._match = None if match is None else torch.cat([match.to(torch.float32), match.new_zeros((1,), dtype=torch.float32)])[sel_t]
            elif match is None:
                full, self._match = self.shard.gather_dealt(losses, by_cost, at=sel_t), None
            else:
                full, self._match = self.shard.gather_dealt(losses, by_cost, extra=match, at=sel_t)
        else:
            full, self._match = self.shard.gather2(losses, match, n)
        if self.opt.loss_in_model_dtype:
            full = full.to(self.model.dtype)        # the reference's CE returns the model dtype
        if self.score_log is not None:
            # debugging aid (tools/nan_bisect.py): which route scored this call and whether every loss is finite -- kept
            # as device scalars, read by the owner of the list after the run, so that nothing here stops the host
            extra = {}
            if feats is not None:
                extra["feats_bad"] = (~torch.isfinite(feats.float())).sum()
            extra["ids_min"], extra["ids_max"] = sampled.min(), sampled.max()
            cache = sc.cache
            if cache is not None and hasattr(cache, "k"):
                extra["prefix_bad"] = sum((~torch.isfinite(t.float())).sum() for t in list(cache.k) + list(cache.v))
            for k_, t in sc.segs:
                if t is not None and k_ == "shared":
                    extra["segs_bad"] = extra.get("segs_bad", 0) + (~torch.isfinite(t.float())).sum()
            self.score_log.append(dict(n=n, **sc.log, **extra, rows=self.score_stats["rows"],
                                           bad=(~torch.isfinite(full.float())).sum(),
                                           first_bad=(~torch.isfinite(full.float())).to(torch.int32).argmax()))
        return full

    def rescore_winner(self, winner: Tensor, order: List[str], feats: Tensor) -> Tensor:
        """Loss of the step winner scored alone, with the image, on every rank (:605-612).
        One candidate shares nothing with anybody: one plain full-sequence forward (no prefix
        cache), fixed shapes, replayed from a hipGraph after the first call."""
        def eager(ids, f):
            keep, self.shard = self.shard, _SOLO
            try:
                loss = self.score_candidates(ids, order, f, allow_prefix=False)
                return loss, self._match          # the early-stop hit belongs to the outputs (None without early_stop)
            finally:
                self.shard = keep

        key = tuple(order)
        if not self.opt.graph_scoring or self._rescore_graphs.get(key) is False:
            return eager(winner, feats)[0]
        g = self._rescore_graphs.get(key)
        if g is None:
            try:
                g = _ReplayGraph(self.model.device, eager, winner, feats)
                self.graphs_captured.append("rescore:" + "|".join(order))
            except Exception as e:
                self._fallback("graph_rescore", e, "winner re-scoring not captured into a graph; running eager")
                self._rescore_graphs[key] = False
                torch.cuda.synchronize(self.model.device)
                return eager(winner, feats)[0]
            self._rescore_graphs[key] = g
        # a replay runs no Python: the hit tensor of THIS forward is the graph's static output, not
        # whatever the last eager scoring call left in self._match
        loss, self._match = g(winner, feats)
        return loss

    def _warm_gemms(self, n_opt: int) -> None:
        """Touch every (row count, projection shape) pair the ragged scoring forwards of this run can meet, once, before
        the first step (EngineOptions.warm_gemms): kernel choice and lazy code-object loading depend on the shape alone,
        so zeros against the first decoder layer's own weights do."""
        cfg, hf = self.config, self.hf
        if not (self.opt.warm_gemms and cfg.gcg_attack and self.opt.ragged_suffix and self.opt.prefix_reuse
                and self.opt.shared_prefix_attention and self.opt.target_rows_only and hf.shared_prefix_configs()
                and self.model.dtype in (torch.bfloat16, torch.float16)):
            return
        order = segment_order("pgd" if (cfg.pgd_attack and cfg.joint_eval) else "gcg", hf.model_type, single=True) \
            if cfg.pgd_attack else segment_order("gcg", hf.model_type, no_joint_eval=True)
        prefix_names, tail = split_at_suffix(order)
        if not tail or tail[0] != "optim" or "image" in tail:
            return                               # (suffix in front of the image: padded chunks, another set of shapes)
        L = n_opt + sum(self.seg[("target_in" if t == "target" else t)].shape[1] for t in tail if t != "optim")
        world = self.emulate_world if (self.emulate_world > 1 and not self.shard.enabled) else self.shard.world
        widths = {self._width(i) for i in range(cfg.num_steps)} if (cfg.dynamic_search or self.opt.width_override) else {cfg.search_width}
        layer = next((l for l, _, _ in self.fused.layers), None)
        if layer is None:
            return
        shapes = set()
        attn, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
        lin = lambda m, n: getattr(m, n) if isinstance(getattr(m, n, None), torch.nn.Linear) else None    # noqa: E731
        qkv = [lin(attn, n) for n in ("q_proj", "k_proj", "v_proj")]
        if all(qkv):
            if self.opt.derived_weight_copies and attn in self.fused.qkv:
                shapes.add((sum(l.out_features for l in qkv), qkv[0].in_features))
            else:
                shapes |= {(l.out_features, l.in_features) for l in qkv}
        gu = [lin(mlp, n) for n in ("gate_proj", "up_proj")]
        if all(gu):
            if self.opt.derived_weight_copies:
                shapes.add((gu[0].out_features + gu[1].out_features, gu[0].in_features))
            else:
                shapes |= {(l.out_features, l.in_features) for l in gu}
        shapes |= {(l.out_features, l.in_features) for l in (lin(attn, "o_proj"), lin(mlp, "down_proj")) if l is not None}
        counts = set()
        for w_ in sorted(widths)[-8:] if len(widths) > 8 else widths:      # (a long width schedule: its widest steps)
            counts |= set(expected_row_counts(w_, n_opt, L, cfg.n_replace, cfg.topk, (world,)))
        dev, dt = self.model.device, self.model.dtype
        with torch.no_grad():
            for out_f, in_f in sorted(shapes):
                w = torch.zeros((out_f, in_f), dtype=dt, device=dev)
                for M in sorted(counts):
                    torch.nn.functional.linear(torch.zeros((1, M, in_f), dtype=dt, device=dev), w)
                del w
        self.warmed = dict(row_counts=sorted(counts), shapes=sorted(shapes))

    # ------------------------------------------------------------ buffer init
    def init_buffer(self, image) -> AttackBuffer:
        cfg, tok, dev = self.config, self.tokenizer, self.model.device
        buffer = AttackBuffer(cfg.buffer_size)
        if isinstance(cfg.optim_str_init, str):
            first = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
            if cfg.buffer_size > 1:
                pool = tok(INIT_CHARS, add_special_tokens=False, return_tensors="pt")["input_ids"].squeeze().to(dev)
                pick = torch.randint(0, pool.shape[0], (cfg.buffer_size - 1, first.shape[1]))   # CPU draw, as :847
                ids = torch.cat([first, pool[pick.to(dev)]], dim=0)
            else:
                ids = first
        else:
            if len(cfg.optim_str_init) != cfg.buffer_size:
                logger.warning(f"Using {len(cfg.optim_str_init)} initializations but buffer size is set to {cfg.buffer_size}")
            ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
        ids = ids.to(torch.int64).contiguous()
        n = max(1, cfg.buffer_size)
        mt = self.hf.model_type
        with torch.no_grad():
            if cfg.pgd_attack:
                feats = self.image_features(image)
                losses = self.score_candidates(ids[:n], segment_order("gcg_pgd", mt, single=True), feats)
            else:
                losses = self.score_candidates(ids[:n], segment_order("gcg", mt, no_joint_eval=True), None)
        if cfg.early_stop and self._match is not None and bool(self._match.any().item()):
            self.stop_flag = True
        self.init_losses = losses.float().clone()
        host = losses.float().cpu().tolist()
        for i in range(n):
            buffer.add(host[i], ids[[i]])
        buffer.log_buffer(tok)
        return buffer

    # ------------------------------------------------------------------- run
    def _sync(self) -> float:
        torch.cuda.synchronize(self.model.device)
        return time.perf_counter()

    def run(self, messages: Union[str, List[dict]], goal: str, target: str,
            image: Optional[Tensor] = None) -> BimodalAttackResult:
        fill_flag = torch.utils.deterministic.fill_uninitialized_memory
        try:
            return self._run(messages, goal, target, image)
        finally:
            torch.utils.deterministic.fill_uninitialized_memory = fill_flag

    def _run(self, messages, goal, target, image) -> BimodalAttackResult:
        """The loop of the reference's run() (:441-794), one call per phase: _step_gradients (A), _step_pgd (B, C),
        _step_sampling (D, device part), _step_scoring (D) and _step_close (times, buffer, logs).  What a run carries
        from step to step lives in a _RunState, what a step's phases hand each other in a _StepState."""
        from transformers import set_seed

        cfg = self.config
        self.initial_prompt = goal
        os.makedirs(cfg.images_folder, exist_ok=True)
        if cfg.seed is not None:
            set_seed(cfg.seed)
            torch.use_deterministic_algorithms(True, warn_only=True)
            # Deterministic mode also NaN-fills every torch.empty() (a debugging aid that costs one
            # HBM write per output tensor, ~3-5 % of a step); no kernel here reads memory it has not
            # written, so the fill is skipped.  Process-wide switch: restored when run() returns.
            torch.utils.deterministic.fill_uninitialized_memory = False
        if cfg.pgd_after_gcg:
            # the reference's pgd_after_gcg branch dies on iteration 0 (:661, current_loss is None)
            raise TypeError("unsupported format string passed to NoneType.__format__")
        if cfg.pgd_attack and image is None:
            raise ValueError("pgd_attack=True needs an image")

        self.n_scored: List[int] = []          # candidates scored at each step
        self.filter_first_steps: List[int] = []    # steps that ran the retokenisation filter BEFORE scoring (filter policy)
        self._keep_rates: List[float] = []     # survivor rate of the filter, step by step
        self._warned_nonfinite = False
        ops.set_graph_owner(id(self))            # (captured graphs replayed from here on are this attack's: ops.note_graph_replay)
        self._prepare_prompt(messages, target)
        rs = _RunState()
        rs.buffer = self.init_buffer(image)
        rs.optim_ids = rs.buffer.get_best_ids()
        self._warm_gemms(rs.optim_ids.shape[1])
        rs.image = image
        rs.trace = self.opt.trace
        rs.writer = _PngWriter() if (cfg.pgd_attack and self.opt.save_images and self.shard.rank == 0) else None

        if cfg.pgd_attack:
            logger.warning(f"Using alpha: {cfg.alpha}, eps: {cfg.eps}")
            image.requires_grad = True            # the caller's tensor, as the reference (:425)
            rs.image_original = image.clone()

        try:
            hook = self.opt.step_hook
            n_done = 0
            self._early, self._parent_host = None, None      # nothing of an earlier run (an early stop leaves unused draws)
            # PGD-only: nothing is sampled, so the loss of the updated image can come out of the
            # next step's gradient pass (fuse_pgd_only); early_stop needs the argmax test of a
            # scoring call, so it keeps the plain loop
            # ... and the gradient pass must compute the very function the reference scores the winner
            # with: it always uses the llava segment order and the unscaled table (:968, :981-991), the
            # re-score the model's own order and embedding scale (:1142, :1150-1163) -- Gemma-3 differs
            # in both, so it keeps the separate re-score
            mt = self.hf.model_type
            rs.fuse_pgd = bool(self.opt.fuse_pgd_only and cfg.pgd_attack and not cfg.gcg_attack and not cfg.early_stop
                               and segment_order("gcg_pgd", mt) == segment_order("gcg_pgd", "llava")
                               and self.hf.emb_scale == 1.0)
            for i in range(cfg.num_steps):
                if hook is not None:
                    hook(i)
                self._stamp("step")
                n_done = i + 1
                sp = _StepState(i)
                if rs.trace is not None:
                    sp.st = {}
                    rs.trace.append(sp.st)
                    sp.st.update(optim_ids_in=rs.optim_ids.cpu().numpy(), n_grad=0, grad_tok=[], grad_img=[], losses=[],
                                 collectives=self.shard.n_collectives)     # (data-path collectives issued so far: tests)
                self._step_gradients(rs, sp)
                self._step_pgd(rs, sp)
                self._step_sampling(rs, sp)
                self._step_scoring(rs, sp)
                if self._step_close(rs, sp):
                    break
            if hook is not None:
                hook(n_done)
            if rs.trace:
                rs.trace[-1]["collectives_end"] = self.shard.n_collectives
        finally:
            if rs.writer is not None:
                rs.writer.close()

        self.final_image = rs.image
        k = rs.losses.index(min(rs.losses))
        return BimodalAttackResult(
            best_loss=rs.losses[k], best_string=rs.strings[k], losses=rs.losses, strings=rs.strings,
            adversarial_suffixes=rs.suffixes, model_outputs=rs.outputs, gradient_times=rs.t_grad, sampling_times=rs.t_samp,
            loss_times=rs.t_loss, pgd_times=rs.t_pgd, total_times=rs.t_total)

    # ---- phase A: gradients ------------------------------------------------------------------------------------------
    @staticmethod
    def _note_gradient(sp: "_StepState", g) -> None:
        st = sp.st
        if st is not None:
            st["n_grad"] += 1
            if g[0] is not None:
                st["grad_tok"].append(g[0][0].float().cpu().numpy())
            if g[1] is not None:
                st["grad_img"].append(g[1].cpu().numpy())

    def _grad_pass(self, rs: "_RunState", sp: "_StepState", record: bool = True, tokens_only: bool = False):
        t0 = self._sync()
        g = self.compute_gradient(rs.optim_ids, rs.image if self.config.pgd_attack else None, tokens_only)
        dt = self._sync() - t0
        rs.t_grad.append(dt)
        if record:
            self._note_gradient(sp, g)
        return g, dt

    def _step_gradients(self, rs: "_RunState", sp: "_StepState") -> None:
        if rs.pending is not None:
            # PGD-only: computed while scoring the previous step.  gradient_ahead: queued behind the previous
            # step's scoring forward and possibly still running; `span` holds its stream events
            (sp.g_tok, sp.g_img, _), sp.grad_time, sp.span = rs.pending
            self._note_gradient(sp, (sp.g_tok, sp.g_img))
            rs.pending = None
        else:
            (sp.g_tok, sp.g_img, _), sp.grad_time = self._grad_pass(rs, sp)

    # ---- phase B: PGD update; phase C: second gradient pass ------------------------------------------------------------
    def _step_pgd(self, rs: "_RunState", sp: "_StepState") -> None:
        cfg = self.config
        if not cfg.pgd_attack:
            return
        if sp.span is not None:           # gradient_ahead: timed on the stream, read after the step
            sp.pgd_span = _Span()
            rs.image = self.perform_pgd_step(rs.image, cfg.eps, cfg.alpha, sp.g_img, rs.image_original)
            sp.pgd_span.stop()
        else:
            t0 = self._sync()
            rs.image = self.perform_pgd_step(rs.image, cfg.eps, cfg.alpha, sp.g_img, rs.image_original)
            sp.pgd_time = self._sync() - t0
            rs.t_pgd.append(sp.pgd_time)
        if sp.st is not None:
            sp.st["image_after_pgd"] = rs.image.detach().cpu().numpy()
        if cfg.gcg_attack and not cfg.joint_eval:
            # several GPUs: rank 0's image overwrites everybody's HERE, before its first consumer -- the
            # pass below caches what it derives from the image by tensor identity (_GradPrefix), so an
            # in-place overwrite behind it (the packed broadcast of the sampling phase) would leave the
            # winner re-score and the next token gradient on the pre-sync pixels
            self.shard.sync_state(rs.image)
            sp.image_synced = True
            # only the token gradient of this pass is used (the image has just been stepped)
            (sp.g_tok, sp.g_img, _), sp.grad_time = self._grad_pass(rs, sp, tokens_only=True)

    # ---- phase D: sampling (device part; the filter runs on the host during scoring) -----------------------------------
    def _step_sampling(self, rs: "_RunState", sp: "_StepState") -> None:
        cfg, i = self.config, sp.i
        image = rs.image
        if sp.span is not None:
            # the gradient pass was queued ahead and may still be running: the sampling kernels go in behind it
            # and the host carries on to the scoring call -- the first thing to wait for the stream is the copy
            # of the sampled ids its plan needs (none at all with early_plan).  The gradient pass is timed by stream
            # events, read after the step.
            early = self._early if (self._early is not None and self._early["step"] == i) else None
            t_s = time.perf_counter()
            sp.sampled_all, sp.job = self.candidate_sampling(i, rs.optim_ids, sp.g_tok, image if cfg.pgd_attack else None)
            # (the sampling kernels' own ~0.3 ms of GPU time are booked with the scoring phase: event pairs
            # around them, recorded while the gradient graph was still running, read 3-18 ms too long)
            sp.flying = (sp.span, time.perf_counter() - t_s, sp.pgd_span)
            self._stamp("sampled")
            if early is not None and self._parent_host is not None:
                sp.virtual = self._virtual_ids(early, self._parent_host)
            self._stamp("virtual")
        else:
            t0 = self._sync()
            sp.sampled_all, sp.job = self.candidate_sampling(i, rs.optim_ids, sp.g_tok,
                                                             image if (cfg.pgd_attack and not sp.image_synced) else None)
            if cfg.gcg_attack:
                sp.samp_time = self._sync() - t0
        if self._filter_first_now():
            # a tokenizer that rejects a real share of the candidates: filter first, score the survivors only.  (In
            # joint mode the image features and the prefix pass -- which need no ids -- are queued in front of the
            # wait, so the GPU has that much to do while the host runs the round trip.)
            if cfg.pgd_attack and cfg.joint_eval and self._gp_enabled() and self._gp not in (None, False):
                with torch.no_grad():
                    self.scoring_features(image)       # (cached: the scoring call below gets the same tensors back)
            sp.sampled_all, sp.job, filt_s = self._filter_now(sp.sampled_all, sp.job)
            self.filter_first_steps.append(i)
            sp.virtual = None              # the plan made from the draws counted candidates that are gone now
            if sp.flying is not None:
                sp.flying = (sp.flying[0], sp.flying[1] + filt_s, sp.flying[2])
            else:
                sp.samp_time += filt_s
        if cfg.gcg_attack:
            st = sp.st
            if st is not None:
                st["sampled"] = self._last["sampled"].cpu().numpy()
                st["topk_idx"] = self._last["topk_idx"].cpu().numpy()
                st["pos"] = self._last["pos"].cpu().numpy()
                st["rank"] = self._last["rank"].cpu().numpy()

    # ---- phase D: scoring ----------------------------------------------------------------------------------------------
    def _survivors(self, sp: "_StepState", loss_all: Tensor, defer_hit: bool = False):
        """Apply the retokenisation filter, computed on the host while the GPU
        scored, to the losses: the reference's filtered vector, in order.  `defer_hit`: the
        early-stop verdict is handed back as a device tensor instead of being read here."""
        cfg, st, job, sampled_all = self.config, sp.st, sp.job, sp.sampled_all
        self._stamp("enqueued")
        keep = job.result()
        self._stamp("filtered")
        if getattr(job, "enabled", False):
            self._keep_rates.append(len(keep) / max(1, loss_all.shape[0]))
        if len(keep) == loss_all.shape[0]:
            idx = None
            out = loss_all, sampled_all
        else:
            idx = self._upload(np.asarray(keep, dtype=np.int64))
            out = loss_all[idx], sampled_all[idx]
        hit = None
        if cfg.early_stop and self._match is not None:
            hit = self._match if idx is None else self._match[idx]
            if not defer_hit and bool(hit.any().item()):
                self.stop_flag = True
        if st is not None:
            if cfg.filter_ids:
                st["filtered"] = out[1].cpu().numpy()
            st["losses"].append(out[0].float().cpu().numpy())
        return (*out, hit) if defer_hit else out

    def _settle(self, sp: "_StepState", loss: Tensor, sampled: Tensor, hit: Optional[Tensor], img: Optional[Tensor]):
        """gradient_ahead: the outcome stays on the device -- argmin -> winner -> the NEXT step's
        gradient pass, queued behind the scoring forward -- and the host gets ONE packed read-back
        (index, loss, early-stop verdict, the winner's ids), marked by an event in FRONT of that pass:
        whatever the host does from the read to the next sampling launch, the GPU is not waiting for
        it.  Returns (winner, host values, host clock at the read, the queued pass or None)."""
        cfg, i = self.config, sp.i
        at = loss.argmin().reshape(1)
        winner = sampled.index_select(0, at)
        f64 = torch.float64
        read = self._read_later(torch.cat([
            at.to(f64), loss.index_select(0, at).to(f64),
            (hit.any().reshape(1) if hit is not None else at.new_zeros(1)).to(f64),
            winner.reshape(-1).to(f64)]))
        queued = None
        rng_before = None
        if i + 1 < cfg.num_steps:
            # the draws of step i+1 come out of the global generator HERE, ahead of whatever else step i
            # still takes from it in the reference's order -- debug_output's generate() under a sampling
            # generation_config (:745-777) -- so that mode keeps the plain order; and a step that turns
            # out to stop the run (early_stop) gives its draws back, as if they had never been made
            if self.opt.early_plan and cfg.gcg_attack and not cfg.debug_output:
                if cfg.early_stop:
                    rng_before = self._rng_state()
                self._draw_ahead(i + 1, winner.shape[1])
            span = _Span()
            with torch.enable_grad():
                g_next = self.compute_gradient(winner, img)
            queued = (g_next, None, span.stop())
        self._stamp("queued_next")
        host = read()
        self._stamp("read")
        if rng_before is not None and host[2] != 0.0:
            self._rng_state(rng_before)
            self._early = None
        return winner, host, time.perf_counter(), queued

    def _step_scoring(self, rs: "_RunState", sp: "_StepState") -> None:
        cfg, mt, i, st = self.config, self.hf.model_type, sp.i, sp.st
        image, sampled_all, virtual = rs.image, sp.sampled_all, sp.virtual
        sp.t0 = time.perf_counter() if sp.flying is not None else self._sync()
        with torch.no_grad():
            ahead_ok = bool(self.opt.gradient_ahead and not (self.opt.tp_gradient and self.shard.enabled))
            host = None
            best_idx = 0
            current_loss = None
            parent = rs.optim_ids if cfg.gcg_attack else None      # what the candidates were sampled from
            if rs.fuse_pgd:
                # the forward of the NEXT gradient pass scores the image just updated.  Behind the LAST step there is
                # no next pass: the same (captured) pass still scores the final image -- its backward is wasted, 28 ms at
                # configs[1]'s size, where the eager scoring forward it replaces met the library with first-sight shapes
                # and took 195 ms, once per attack (profiles/r6_full_pgd100.json) -- and its time is scoring time
                with torch.enable_grad():
                    rs.pending = (*self._grad_pass(rs, sp, record=False), None)
                sp.prefetch_s = rs.pending[1] if i + 1 < cfg.num_steps else 0.0
                full = rs.pending[0][2].reshape(1)
                if i + 1 >= cfg.num_steps:
                    rs.pending = None
                    rs.t_grad.pop()                  # (booked as scoring time: the result keeps one gradient time per step)
                if self.opt.loss_in_model_dtype:
                    full = full.to(self.model.dtype)
                current_loss = full.item()
                best_idx, sampled, winner = 0, sampled_all, sampled_all[0:1].contiguous()
                if st is not None:
                    st["losses"].append(full.float().cpu().numpy())
            elif (cfg.pgd_attack and cfg.gcg_attack and cfg.joint_eval and ahead_ok and self.opt.joint_winner_from_batch
                  and segment_order("pgd", mt, single=True) == segment_order("gcg_pgd", mt)
                  and segment_order("pgd", mt, single=True) == self._GRAD_ORDER):
                # (image-in-front layouts only.  Gemma-3's joint step, suffix in front, was measured 10 % SLOWER
                # this way -- 491.8 -> 539.8 ms, same kernels and launch counts, the library GEMMs of its padded
                # scoring chunks each 10-16 % longer with the whole forward queued far ahead -- and keeps the
                # plain order)
                # joint mode with the winner's loss taken from the batch (see below): nothing but the outcome's
                # read-back needs the host, and the next gradient pass -- the tail rows against the prefix this
                # scoring call records -- needs the winner and the image, both on the device
                feats = self.scoring_features(image)
                loss, sampled, hit = self._survivors(sp, self.score_candidates(
                    sampled_all, segment_order("pgd", mt, single=True), feats, parent=parent, virtual=virtual),
                    defer_hit=True)
                winner, host, sp.t_read, rs.pending = self._settle(sp, loss, sampled, hit, image)
                if st is not None:
                    st["losses"].append(np.asarray([host[1]], dtype=np.float32))
            elif cfg.pgd_attack:
                feats = self.scoring_features(image)
                if cfg.joint_eval:
                    loss, sampled = self._survivors(sp, self.score_candidates(
                        sampled_all, segment_order("pgd", mt, single=True), feats, parent=parent))
                elif cfg.gcg_attack:
                    loss, sampled = self._survivors(sp, self.score_candidates(
                        sampled_all, segment_order("gcg", mt, single=True), None, parent=parent))
                else:
                    loss, sampled = None, sampled_all
                best_idx = int(loss.argmin().item()) if loss is not None else 0
                winner = sampled[best_idx:best_idx + 1].contiguous()
                # re-score the winner with the image (:605-612); on every rank, unsharded.  With
                # joint_eval the candidates WERE scored with the image, in the very segment order of
                # the re-score: the winner's row of that batch is the same function of the same
                # inputs, so it is taken instead of a second forward (joint_winner_from_batch).
                if cfg.joint_eval and loss is not None and self.opt.joint_winner_from_batch and \
                        segment_order("pgd", mt, single=True) == segment_order("gcg_pgd", mt):
                    full = loss[best_idx].reshape(1).clone()
                else:
                    full = self.rescore_winner(winner, segment_order("gcg_pgd", mt), feats)
                    if cfg.early_stop and self._match is not None:
                        hit = self._match.reshape(-1)[:1].to(torch.float32)
                        if self.shard.enabled:
                            # every rank re-scored the winner itself: rank 0's verdict decides, or ranks
                            # could leave the loop at different steps
                            both = torch.cat([full.reshape(-1)[:1].to(torch.float32), hit])
                            self.shard.broadcast_(both)
                            full, hit = both[:1].to(full.dtype), both[1:]
                        if bool(hit.any().item()):
                            self.stop_flag = True
                current_loss = full.item()
                if st is not None:
                    st["losses"].append(full.float().cpu().numpy())
            elif ahead_ok:
                loss, sampled, hit = self._survivors(sp, self.score_candidates(
                    sampled_all, segment_order("gcg", mt, no_joint_eval=True), None, parent=parent,
                    virtual=virtual), defer_hit=True)
                winner, host, sp.t_read, rs.pending = self._settle(sp, loss, sampled, hit, None)
            else:
                loss, sampled = self._survivors(sp, self.score_candidates(
                    sampled_all, segment_order("gcg", mt, no_joint_eval=True), None, parent=parent))
                best_idx = int(loss.argmin().item())
                current_loss = loss[best_idx].item()
                winner = sampled[best_idx:best_idx + 1].contiguous()
            if sp.t_read is not None:
                best_idx, current_loss = int(host[0]), float(host[1])
                if host[2] != 0.0:
                    self.stop_flag = True
                sp.ids_host = [int(v) for v in host[3:]]
                self._parent_host = sp.ids_host
                if sp.flying is not None:
                    # the phases tile the step: what is left of the period between two read-backs after the
                    # gradient pass, the PGD update and the sampling kernels is the scoring phase (host
                    # planning included)
                    sp.grad_time, sp.samp_time = sp.flying[0].seconds(), sp.flying[1]
                    rs.t_grad.append(sp.grad_time)
                    if sp.flying[2] is not None:
                        sp.pgd_time = sp.flying[2].seconds()
                        rs.t_pgd.append(sp.pgd_time)
                    sp.t0 = min(sp.t_read, self._t_read + sp.grad_time + sp.pgd_time + sp.samp_time)
                self._t_read = sp.t_read
            sp.sampled, sp.n = sampled, sampled.shape[0]
            rs.optim_ids = winner                      # greedy: accepted even when worse (:613, :638)
            if current_loss != current_loss or current_loss in (float("inf"), float("-inf")):
                # the reference would carry a NaN on silently (argmin then picks arbitrary winners and
                # sign(NaN) poisons the image for good); so does the engine -- but it says so, once per run
                if not getattr(self, "_warned_nonfinite", False):
                    self._warned_nonfinite = True
                    logger.warning(f"[Iteration {i}] non-finite loss ({current_loss}): the run continues as the reference's "
                                   "would, but its results from here on are meaningless")
            sp.current_loss = current_loss
            rs.losses.append(current_loss)
            rs.strings.append(self.tokenizer.batch_decode(rs.optim_ids if sp.ids_host is None else [sp.ids_host])[0])
            if rs.buffer.size == 0 or current_loss < rs.buffer.get_highest_loss():
                rs.buffer.add(current_loss, rs.optim_ids)
            self.n_scored.append(sp.n)
            if st is not None:
                st.update(best_idx=best_idx, current_loss=current_loss, n_scored=sp.n)

    # ---- the step's times, outputs and logs; True when the run stops here -------------------------------------------------
    def _step_close(self, rs: "_RunState", sp: "_StepState") -> bool:
        cfg, i = self.config, sp.i
        # a prefetched gradient pass is booked as gradient time
        loss_time = max((self._sync() if sp.t_read is None else sp.t_read) - sp.t0 - sp.prefetch_s, 0.0)
        if cfg.gcg_attack:
            # the reference books the filter under "sampling"; here it ran beside the forward, so what it
            # cost this section is the time result() was blocked on it
            # (gradient_ahead: result() is reached while the stream is still on its way to the ids' copy --
            # that wait is the gradient pass's, already booked; the round trip's own duration counts)
            filt = sp.job.seconds if sp.flying is not None else sp.job.waited
            sp.samp_time += filt
            loss_time = max(loss_time - filt, 0.0)
            rs.t_samp.append(sp.samp_time)
        rs.t_loss.append(loss_time)
        logger.info(f"[Iteration {i}] Current loss: {sp.current_loss:.4f} | Best loss: {rs.buffer.get_lowest_loss():.4f} | ")

        if rs.writer is not None:
            rs.writer.submit(rs.image, os.path.join(cfg.images_folder, f"{i}.png"))
        if cfg.debug_output and i % 10 == 0:
            rs.outputs.append(self._debug_generate(sp.sampled, rs.image if cfg.pgd_attack else None, sp.n))
        else:
            rs.outputs.append("")
        rs.suffixes.append(rs.strings[-1])           # the reference decodes the same ids a second time (:782)
        rs.buffer.log_buffer(self.tokenizer)
        if self.stop_flag:
            logger.info("Early stopping due to finding a perfect match.")
            return True
        rs.t_total.append(sp.grad_time + sp.samp_time + sp.pgd_time + loss_time)
        return False

    # ------------------------------------------------------------ debug output
    def _debug_generate(self, sampled: Tensor, image: Optional[Tensor], n: int) -> str:
        """debug_output (:745-777): greedy generation from the first candidate's prompt."""
        mt, E = self.hf.model_type, self.embedding_layer.weight
        with torch.no_grad():
            if image is not None:
                feats = self.image_features(image).to(E.dtype)
                order = segment_order("gcg_pgd", mt, no_target=True)
            else:
                feats, order = None, segment_order("gcg", mt, no_target=True)
            x = ops.splice(self._segments(order, feats), n, E, sampled.contiguous(), self.hf.emb_scale)
            out = self.model.generate(inputs_embeds=x, max_new_tokens=120)
        text = self.tokenizer.decode(out[0], skip_special_tokens=True)
        logger.info(f"Output generated: {text}")
        return text


# hipGraph captures police the capturing THREAD only: the process has other threads that talk to the runtime while a
# capture is open -- RCCL's watchdog polling events of earlier collectives, the PNG writer copying an image to the
# host -- and in the default ("global") mode any such call from any thread invalidates the capture.
_CAPTURE_MODE = os.environ.get("BMA_CAPTURE_MODE", "thread_local")


class _GradPrefix:
    """The prefix pass of joint-mode candidate scoring, run WITH autograd so that it is also the first 599 rows of
    the next gradient pass (``BimodalAttack._gp_enabled``).

    ``features(image)``  -- scoring, step t: vision tower + prompt up to the suffix at batch 1; the recorded keys/
                            values keep their history; scoring gets detached views.
    ``gradient(ids, image)`` -- step t+1: the 44 tokens behind the prefix run forward against those keys/values
                            (prefix_attention.tail_grad_attention), the loss back-propagates through both parts.

    With hipGraphs (options graph_gradient and graph_scoring): two captures sharing one autograd graph, as
    torch.cuda.make_graphed_callables does for a forward and its backward -- G1 = the prefix forward, G2 = the tail
    forward + the whole backward.  Each keeps its own memory pool, so what G1's replay writes (saved activations,
    keys/values) is read by G2's replay and by scoring and touched by nobody else.  The same maths as the full pass up
    to the rounding of different GEMM shapes and of attention over [prefix ; tail] keys in one library call."""

    def __init__(self, attack: "BimodalAttack"):
        import weakref
        # a proxy, not a reference: attack -> _GradPrefix -> attack would be a cycle, and a cycle holding hipGraphs and
        # their memory pools is freed by the garbage collector whenever it happens to run -- e.g. in the middle of
        # some later capture, which aborts the process
        self.a = weakref.proxy(attack)
        self.names = ("before_img", "image", "before_suffix")
        self.image: Optional[Tensor] = None        # leaf the prefix was computed from (static under graphs)
        self.ids: Optional[Tensor] = None
        self.feats = self.rec = self.out = None
        self.current = None                        # the caller's image tensor the prefix state belongs to
        self.g1 = self.g2 = self.g3 = None
        self.ids3 = self.out3 = None
        self._feats_out = None
        self.graphs = bool(attack.opt.graph_gradient and attack.opt.graph_scoring)
        self.P = 0
        self._cache = None

    # -- the two halves ---------------------------------------------------------------------------
    def _prefix_fn(self):
        a = self.a
        with torch.enable_grad():
            feats = a.image_features(self.image)
            x = torch.cat([a.seg["before_img"], feats.to(a.model.dtype), a.seg["before_suffix"]], dim=1)
            with a.fused, a._b1_attention(x.shape[1]):
                rec = a.hf.build_prefix_recording(x)
        return feats, rec

    def _tail_fn(self):
        a = self.a
        E = a.embedding_layer.weight
        emb = E[self.ids[0]].unsqueeze(0).detach().requires_grad_()
        tail = torch.cat([emb, a.seg["after"], a.seg["target_in"]], dim=1)
        with torch.enable_grad(), a.fused:
            logits = a.hf.target_logits_behind_grad_prefix(tail, a.T, self.rec)
            loss = ops.TargetCrossEntropy.apply(logits[0], a.labels)
        g_emb, g_img = torch.autograd.grad(loss, [emb, self.image])
        with torch.no_grad():
            g_tok = (g_emb[0] @ E.t()).unsqueeze(0)
        return g_tok, g_img, loss.detach()

    def _set(self, feats, rec) -> None:
        from .prefix_attention import RecordingKV
        self.feats, self.rec = feats, rec
        self.P = int(rec.k[0].shape[2])
        det = RecordingKV(len(rec.k))
        det.k, det.v = [t.detach() for t in rec.k], [t.detach() for t in rec.v]
        self._cache = det
        self._feats_out = feats.detach()

    # -- what the engine calls ----------------------------------------------------------------------
    def features(self, image: Tensor) -> Tensor:
        """Make the prefix state current for `image`; the image features for scoring (no history)."""
        if self.current is image and self._cache is not None:
            return self._feats_out
        if self.g1 is not None:
            with torch.no_grad():
                self.image.copy_(image)
            ops.note_graph_replay(self.image.device)
            self.g1.replay()
        else:
            self.image = image.detach().clone().requires_grad_()
            self._set(*self._prefix_fn())
        self.current = image
        return self._feats_out

    def serves(self, key: tuple, feats: Tensor) -> bool:
        return key == self.names and feats is self._feats_out and self._cache is not None

    def cache(self):
        return self._cache

    def _tokens_fn(self):
        """Token gradient only: the prefix is a constant of the suffix, so its detached keys/values do."""
        a = self.a
        E = a.embedding_layer.weight
        emb = E[self.ids3[0]].unsqueeze(0).detach().requires_grad_()
        tail = torch.cat([emb, a.seg["after"], a.seg["target_in"]], dim=1)
        with torch.enable_grad(), a.fused:
            logits = a.hf.target_logits_behind_grad_prefix(tail, a.T, self._cache)
            loss = ops.TargetCrossEntropy.apply(logits[0], a.labels)
        (g_emb,) = torch.autograd.grad(loss, [emb])
        with torch.no_grad():
            g_tok = (g_emb[0] @ E.t()).unsqueeze(0)
        return g_tok, None, loss.detach()

    def _tokens(self, optim_ids: Tensor):
        if self.g1 is None:                        # eager: the history stays for the pass that wants the pixels
            self.ids3 = optim_ids
            return self._tokens_fn()
        if self.g3 is None:
            dev = self.a.model.device
            self.ids3 = optim_ids.detach().clone()
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):
                self._tokens_fn()
            torch.cuda.current_stream(dev).wait_stream(side)
            torch.cuda.synchronize(dev)
            g3 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g3, capture_error_mode=_CAPTURE_MODE):
                self.out3 = self._tokens_fn()
            self.g3 = g3
            self.a.graphs_captured.append("grad_tail_tokens")
        with torch.no_grad():
            self.ids3.copy_(optim_ids)
        ops.note_graph_replay(self.ids3.device)
        self.g3.replay()
        return self.out3

    def gradient(self, optim_ids: Tensor, image: Tensor, tokens_only: bool = False):
        if self.graphs and self.g2 is None:
            self._capture(optim_ids, image)
        if self.current is not image:
            self.features(image)
        if tokens_only:
            return self._tokens(optim_ids)
        if self.g2 is not None:
            with torch.no_grad():
                self.ids.copy_(optim_ids)
            ops.note_graph_replay(self.ids.device)
            self.g2.replay()
            return self.out
        self.ids = optim_ids
        out = self._tail_fn()
        self.current = None                        # eager: the history was consumed by this backward
        return out

    def _capture(self, optim_ids: Tensor, image: Tensor) -> None:
        a, dev = self.a, self.a.model.device
        self.image = image.detach().clone().requires_grad_()
        self.ids = optim_ids.detach().clone()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):              # lazy initialisations must not land in a capture
            self._set(*self._prefix_fn())
            self._tail_fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        a.fused.register_gemm_chain()
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        with torch.cuda.graph(g1, capture_error_mode=_CAPTURE_MODE):
            feats, rec = self._prefix_fn()
        self._set(feats, rec)
        with torch.cuda.graph(g2, capture_error_mode=_CAPTURE_MODE):
            self.out = self._tail_fn()
        self.g1, self.g2 = g1, g2
        self.current = None
        a.graphs_captured.append("grad_prefix")
        a.graphs_captured.append("grad_tail")


class _GradientGraph:
    """The batch-1 gradient pass as one hipGraph.  Inputs live in static buffers that are
    overwritten before each replay; outputs are static too and stay valid until the next
    replay (the loop consumes them at once)."""

    def __init__(self, attack: "BimodalAttack", optim_ids: Tensor, image: Optional[Tensor], fn=None):
        dev = attack.model.device
        fn = fn or attack._gradient_eager
        self.ids = optim_ids.detach().clone()
        self.image = None if image is None else image.detach().clone().requires_grad_()
        side = torch.cuda.Stream(dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):          # lazy initialisations must not land in the capture
            fn(self.ids, self.image)
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        attack.fused.register_gemm_chain()     # the derived weight copies exist now: every skinny product learns its successor
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
            self.out = fn(self.ids, self.image)

    def __call__(self, optim_ids: Tensor, image: Optional[Tensor]):
        with torch.no_grad():
            self.ids.copy_(optim_ids)
            if self.image is not None:
                self.image.copy_(image)
        ops.note_graph_replay(self.ids.device)
        self.graph.replay()
        return self.out


class _ReplayGraph:
    """A fixed-shape, sync-free function of device tensors as one hipGraph: static input
    buffers are overwritten before each replay, the (static) outputs stay valid until the
    next one."""

    def __init__(self, device, fn, *inputs: Tensor):
        self.inputs = [t.detach().clone() for t in inputs]
        side = torch.cuda.Stream(device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():      # lazy initialisations stay out of the capture
            fn(*self.inputs)
        torch.cuda.current_stream(device).wait_stream(side)
        torch.cuda.synchronize(device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph, capture_error_mode=_CAPTURE_MODE):
            self.out = fn(*self.inputs)

    def __call__(self, *inputs: Tensor):
        with torch.no_grad():
            for dst, src in zip(self.inputs, inputs):
                dst.copy_(src)
        ops.note_graph_replay(self.inputs[0].device)
        self.graph.replay()
        return self.out


class _AlreadyFiltered:
    """Stands in for the step's FilterJob once the round trip has run (filter-first): every candidate left is a survivor."""
    seconds = waited = 0.0
    enabled = False

    def __init__(self, n: int):
        self.n = n

    def result(self) -> List[int]:
        return list(range(self.n))


class _Solo:
    """A sharder that does nothing: used while every rank re-scores the winner."""
    enabled, world, rank = False, 1, 0

    @staticmethod
    def bounds(n, rank=None):
        return 0, n

    @staticmethod
    def gather2(local, extra, n, pad=float("inf"), pad_extra=0.0):
        return local, extra


_SOLO = _Solo()


def run(model, tokenizer, processor, messages: Union[str, List[dict]], goal: str, target: str,
        image: Optional[Tensor] = None, config: Optional[BimodalAttackConfig] = None, normalize=None,
        **engine_options) -> BimodalAttackResult:
    """Drop-in for the reference's ``bimodalattack.run`` (:1323-1338).  Extra keyword
    arguments are engine options (``EngineOptions``), never config fields."""
    if config is None:
        config = BimodalAttackConfig()
    logger.setLevel(getattr(logging, config.verbosity))
    attack = BimodalAttack(model, tokenizer, processor, config, normalize, EngineOptions.from_env(**engine_options))
    return attack.run(messages, goal, target, image)
