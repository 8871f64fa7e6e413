"""Fused inference path for the elementwise tail of Llama-family decoder layers.

While a :class:`FusedInference` context is active, the RMSNorm modules, SiLU-gated MLPs and
rotary-embedding function of the given HuggingFace model run the one-pass kernels of
``csrc/fused_elementwise.hip`` instead of their eager op chains: directly when autograd is
off (candidate scoring), through autograd Functions with fused backward kernels when a
gradient is being recorded (the gradient pass; weights are treated as constants -- if a
norm weight requires grad the eager module runs).  Outside the context -- or when the input
is not on the GPU or a shape is beyond the kernels' limits -- the original code runs.
Nothing is left on the user's model afterwards: patches are instance attributes removed
on exit.  Under autograd with at most 1024 rows (the batch-1 gradient pass, with or without the image
tokens) the decoder layers'
bias-free projections additionally compute their input gradient through a transposed copy of the
weight (``ops.FrozenLinearFn``; one extra copy of the language model's weights in HBM).  In 16-bit
models the q/k/v projections of an attention block run as one product against their concatenated
weight (one more copy of those three matrices).

What qualifies (checked structurally, not by model name):
  * a module whose class name ends in ``RMSNorm`` with a 1-D ``weight`` and an epsilon
    (``variance_epsilon`` / ``eps``); classes named ``Gemma*`` use the (1 + w) form;
  * a module with ``gate_proj``, ``up_proj``, ``down_proj`` and a SiLU or GELU-tanh ``act_fn``;
  * ``apply_rotary_pos_emb`` of the modelling file of known rotary families (full-head
    rotary, ``rotate_half`` convention): llama, mistral, qwen2, gemma3.
"""

from __future__ import annotations

import sys
from typing import List, Tuple

import torch

from . import ops

import os as _os

_ROPE_FILES = ("modeling_llama", "modeling_mistral", "modeling_qwen2", "modeling_gemma3")
# Gemma-3's per-head q_norm / k_norm inside the rotary launch of the no-grad scoring forward (bma_qknorm_rope2): one pass
# over q and k instead of two.  An A/B switch (an engine option until round 4): the engine passes it to FusedInference.
FUSE_QK_ROPE = _os.environ.get("BMA_FUSE_QK_ROPE", "1") not in ("0", "false", "False")
_DTYPES = (torch.bfloat16, torch.float16, torch.float32)


def _eps_of(m):
    for name in ("variance_epsilon", "eps"):
        v = getattr(m, name, None)
        if isinstance(v, float):
            return v
    return None


SKINNY_ROWS = 1024     # rows up to which the gradient pass takes its input gradients through transposed weight copies
# A no-grad product whose row count sits just above a whole number of 256 x 256 tile rounds of the library's kernel on this
# chip's CUs runs as TWO calls -- the whole rounds, then the rest: 16896 x 4096 is 66 x 16 = 1056 tiles = 4.1 rounds on 256
# CUs and takes five (o_proj 416 us, down_proj 1072 us), 16384 rows + 512 rows take 388 / 1000 us (profiles/r6_round_split.txt;
# q/k/v and gate/up, whose rounds do not end on a row-tile boundary, gain nothing and are left alone).  BMA_ROUND_SPLIT=0: off.
ROUND_SPLIT = _os.environ.get("BMA_ROUND_SPLIT", "1") not in ("0", "false", "False")
_CUS = {}


def round_cut(rows: int, n_out: int, cus: int) -> int:
    """Row count of the first of two calls for a rows x n_out product on `cus` compute units, 0 = one call.  In 256 x 256
    tiles, n_out is C tile columns; cus tiles make a round, and rounds end on a row-tile boundary every cus / gcd(cus, C) row
    tiles (16 for C = 16 or 48, 128 for C = 86).  The cut is the last such boundary, taken when what is left behind it is at
    most 5/16 of a round's tiles (measured: at 6/16 the second call costs what the saved part-round did)."""
    from math import gcd
    col_tiles = (n_out + 255) // 256
    row_tiles = (rows + 255) // 256
    if col_tiles <= 0 or cus <= 0:
        return 0
    period = cus // gcd(cus, col_tiles)
    cut_tiles = row_tiles // period * period
    rest_tiles = (row_tiles - cut_tiles) * col_tiles
    return cut_tiles * 256 if cut_tiles > 0 and 0 < rest_tiles <= cus * 5 // 16 else 0


def col_cut(rows: int, n_out: int, cus: int) -> int:
    """Column count of the first of two calls, 0 = one call: the most tile columns that still fill WHOLE rounds, when the
    product has at most six of them and what is left is at most 5/16 of a round's tiles -- rank 0's gate/up product of
    eight GPUs (2112-2304 rows: 9 x 86 = 774 tiles = 3.02 rounds, run as four): 85 tile columns, then one (323-338 -> 301-304
    us, profiles/r6_round_split.txt).  With many rounds the saved one is worth less than a product 256 columns wide costs
    (16896 rows: 2028 -> 2039 us), with more left behind the cut the second call costs more than the round (2432 rows, 100
    tiles: 340 -> 493 us)."""
    col_tiles = (n_out + 255) // 256
    row_tiles = (rows + 255) // 256
    if col_tiles <= 1 or rows < 1536 or cus <= 0:          # (a few hundred rows: the library's picks are not 256-row tiles)
        return 0
    rounds = row_tiles * col_tiles // cus                 # whole rounds
    if not 1 <= rounds <= 6 or row_tiles * col_tiles == rounds * cus:
        return 0
    cut = rounds * cus // row_tiles                       # tile columns of the first call
    rest_tiles = row_tiles * (col_tiles - cut)
    return cut * 256 if 0 < cut < col_tiles and 0 < rest_tiles <= cus * 5 // 16 else 0


def two_calls(x: torch.Tensor, w: torch.Tensor):
    """x @ w.T for a no-grad 16-bit product as two library calls into slices of one output when round_cut / col_cut say
    so; None = the caller's own single call."""
    if not (ROUND_SPLIT and x.is_cuda and x.dtype == w.dtype and x.dtype in (torch.bfloat16, torch.float16)
            and x.is_contiguous() and w.is_contiguous() and x.dim() >= 2):
        return None
    cus = _CUS.get(x.device)
    if cus is None:
        cus = _CUS[x.device] = torch.cuda.get_device_properties(x.device).multi_processor_count
    rows, n_out = x.numel() // x.shape[-1], w.shape[0]
    cut = round_cut(rows, n_out, cus)
    if cut:
        x2 = x.view(rows, x.shape[-1])
        out = x2.new_empty((rows, n_out))
        torch.mm(x2[:cut], w.t(), out=out[:cut])
        torch.mm(x2[cut:], w.t(), out=out[cut:])
        return out.view(*x.shape[:-1], n_out)
    cut = col_cut(rows, n_out, cus)
    if cut:
        x2 = x.view(rows, x.shape[-1])
        out = x2.new_empty((rows, n_out))
        torch.mm(x2, w[:cut].t(), out=out[:, :cut])
        torch.mm(x2, w[cut:].t(), out=out[:, cut:])
        return out.view(*x.shape[:-1], n_out)
    return None


# Derived weight copies (transposed, concatenated q/k/v, interleaved gate/up) belong to the MODEL, not to one attack
# object: a second attack on the same model (the next prompt of an experiment, bench.py's other workloads) finds them
# instead of building another 30 GB.  Every copy remembers the (data_ptr, _version) of the tensors it was made from
# and is rebuilt when a caller has changed the weights in between (``_CopyCache.get``).
import weakref

_COPY_CACHES: "weakref.WeakKeyDictionary" = weakref.WeakKeyDictionary()


class _CopyCache:
    def __init__(self):
        self.d = {}

    @staticmethod
    def _stamp(srcs):
        return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in srcs)

    def get(self, key, srcs=None):
        """The cached copy under `key`, or None when there is none or its sources changed since it was made."""
        hit = self.d.get(key)
        if hit is None:
            return None
        val, stamp = hit
        if srcs is not None and stamp != self._stamp(srcs):
            del self.d[key]
            return None
        return val

    def put(self, key, val, srcs):
        self.d[key] = (val, self._stamp(srcs))
        return val

    def __len__(self):
        return len(self.d)


_LLAMA_BODY = ["residual = hidden_states", "hidden_states = self.input_layernorm(hidden_states)",
               "hidden_states, _ = self.self_attn(", "hidden_states = residual + hidden_states", "residual = hidden_states",
               "hidden_states = self.post_attention_layernorm(hidden_states)", "hidden_states = self.mlp(hidden_states)",
               "hidden_states = residual + hidden_states", "return hidden_states"]
_GEMMA_BODY = ["residual = hidden_states", "hidden_states = self.input_layernorm(hidden_states)",
               "hidden_states, _ = self.self_attn(", "hidden_states = self.post_attention_layernorm(hidden_states)",
               "hidden_states = residual + hidden_states", "residual = hidden_states",
               "hidden_states = self.pre_feedforward_layernorm(hidden_states)", "hidden_states = self.mlp(hidden_states)",
               "hidden_states = self.post_feedforward_layernorm(hidden_states)", "hidden_states = residual + hidden_states",
               "return hidden_states"]


def _layer_kind(layer):
    """"llama" / "gemma" when the decoder layer's forward is, statement for statement, the residual structure the
    fused forward restates (read from its source: every line that touches `self.`, `residual` or returns), else None."""
    import inspect
    try:
        src = inspect.getsource(type(layer).forward)
    except (OSError, TypeError):
        return None
    lines = []
    for ln in src.splitlines():
        t = ln.strip()
        if t.startswith("#") or t.startswith("def ") or t.startswith("self,") or not t:
            continue
        if "self." in t or t.startswith("residual") or t.startswith("return") or "residual +" in t:
            lines.append(t)
    for kind, want in (("llama", _LLAMA_BODY), ("gemma", _GEMMA_BODY)):
        if len(lines) == len(want) and all(l.startswith(w) if w.endswith("(") else l == w for l, w in zip(lines, want)):
            return kind
    return None


def _decoder_layers(model, norm_info):
    """[(layer, kind, next norm)] for every decoder stack -- a module with a ``layers`` ModuleList and a final ``norm``
    -- whose layers all have a known residual structure and whose norms are ones the fused kernels replace."""
    out = []
    for parent in model.modules():
        layers = getattr(parent, "layers", None)
        final = getattr(parent, "norm", None)
        if not isinstance(layers, torch.nn.ModuleList) or len(layers) == 0 or final is None or id(final) not in norm_info:
            continue
        n_used = getattr(getattr(parent, "config", None), "num_hidden_layers", len(layers))
        if n_used != len(layers):
            continue                                 # the stack runs a prefix of its layers: the hand-over would miss
        kinds = [_layer_kind(l) for l in layers]
        if any(k is None for k in kinds) or len(set(kinds)) != 1:
            continue
        names = ["input_layernorm", "post_attention_layernorm"] + \
            (["pre_feedforward_layernorm", "post_feedforward_layernorm"] if kinds[0] == "gemma" else [])
        if not all(id(getattr(l, n, None)) in norm_info for l in layers for n in names):
            continue
        if not all(hasattr(l, "self_attn") and hasattr(l, "mlp") for l in layers):
            continue
        for i, l in enumerate(layers):
            nxt = layers[i + 1].input_layernorm if i + 1 < len(layers) else final
            out.append((l, kinds[0], nxt))
    return out


def _last_layers(model, admitted) -> set:
    """id() of the last layer of every admitted decoder stack whose own forward does nothing with that layer's output but
    hand it to the final norm (read from its source): there the rows nobody reads may be dropped inside the last layer."""
    import inspect
    ids = {id(l) for l, _, _ in admitted}
    out = set()
    for parent in model.modules():
        layers = getattr(parent, "layers", None)
        if not isinstance(layers, torch.nn.ModuleList) or len(layers) == 0 or id(layers[-1]) not in ids:
            continue
        try:
            lines = [ln.strip() for ln in inspect.getsource(type(parent).forward).splitlines() if ln.strip()]
        except (OSError, TypeError):
            continue
        if "hidden_states = self.norm(hidden_states)" not in lines:
            continue
        tail = [ln for ln in lines[lines.index("hidden_states = self.norm(hidden_states)") + 1:] if not ln.startswith("#")]
        # behind the norm: the optional hidden-state bookkeeping (off in every engine call) and the return -- nothing that
        # would need the dropped rows
        if all(ln.startswith(("if output_hidden_states", "all_hidden_states", "return ", "last_hidden_state=", "past_key_values=",
                              "hidden_states=", "attentions=", ")")) for ln in tail):
            out.add(id(layers[-1]))
    return out


class _TPCopy(torch.autograd.Function):
    """Megatron's f: identity forward, all-reduce of the gradient backward (the input of column-parallel layers)."""

    @staticmethod
    def forward(ctx, x, group):
        ctx.group = group
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        import torch.distributed as dist
        g = g.contiguous().clone()
        dist.all_reduce(g, group=ctx.group)
        return g, None


class _TPReduce(torch.autograd.Function):
    """Megatron's g: all-reduce forward (the partial outputs of a row-parallel layer), identity backward."""

    @staticmethod
    def forward(ctx, x, group):
        import torch.distributed as dist
        y = x.contiguous().clone()
        dist.all_reduce(y, group=group)
        return y

    @staticmethod
    def backward(ctx, g):
        return g, None


_NORM_ROPE_LINES = ("query_states = self.q_norm(query_states)", "key_states = self.k_norm(key_states)",
                    "query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)")


def _norms_then_rope(attn) -> bool:
    """Does the attention block's forward apply q_norm, k_norm and the rotary embedding back to back, in that order,
    with nothing in between (read from its source)?  Only then may the two norms be deferred into the rotary launch."""
    import inspect
    try:
        lines = [ln.strip() for ln in inspect.getsource(type(attn).forward).splitlines() if ln.strip()]
    except (OSError, TypeError):
        return False
    try:
        i = lines.index(_NORM_ROPE_LINES[0])
    except ValueError:
        return False
    rest = [ln for ln in lines[i:i + 6] if not ln.startswith("#") and ln != "cos, sin = position_embeddings"]
    return tuple(rest[:3]) == _NORM_ROPE_LINES


class DeferredNormMissed(RuntimeError):
    """A q/k norm deferred into the rotary launch was not picked up (the attention block did something else with the
    tensor first).  By the time this is raised the deferral is switched off for good on this object: the caller runs
    the forward again and gets the unfused norms."""


_ATTN_SELF_LINES = ["self,", "hidden_shape = (*input_shape, -1, self.head_dim)",
                    "query_states = self.q_proj(hidden_states).view(hidden_shape).transpose(1, 2)",
                    "key_states = self.k_proj(hidden_states).view(hidden_shape).transpose(1, 2)",
                    "value_states = self.v_proj(hidden_states).view(hidden_shape).transpose(1, 2)",
                    "key_states, value_states = past_key_values.update(key_states, value_states, self.layer_idx)",
                    "self.config._attn_implementation, eager_attention_forward", "self,",
                    "dropout=0.0 if not self.training else self.attention_dropout,", "scaling=self.scaling,",
                    "attn_output = self.o_proj(attn_output)", "return attn_output, attn_weights"]


def _plain_rotary_attention(attn) -> bool:
    """Is the attention block's forward, statement for statement, projections -> rotary embedding -> (cache) -> attention
    function -> o_proj with nothing else touching q, k or v (HuggingFace's LlamaAttention, read from its source)?  Only
    then may the span between the projections and o_proj run as ONE launch (ops.B1AttentionFn)."""
    import inspect
    try:
        lines = [ln.strip() for ln in inspect.getsource(type(attn).forward).splitlines()]
    except (OSError, TypeError):
        return False
    if "query_states, key_states = apply_rotary_pos_emb(query_states, key_states, cos, sin)" not in lines or \
            "cos, sin = position_embeddings" not in lines:
        return False
    # every line that touches `self` (the signature's and the attention function's `self,` included) or returns
    seen = [t for t in lines if not t.startswith("#") and not t.startswith("def ") and ("self." in t or t.startswith("return") or t == "self,")]
    return seen == _ATTN_SELF_LINES


_TP_COLUMN = ("q_proj", "k_proj", "v_proj", "gate_proj", "up_proj")
_TP_ROW = ("o_proj", "down_proj")


class FusedInference:
    def __init__(self, model: torch.nn.Module, enabled: bool = True, weight_copies: bool = True,
                 fuse_qkv: bool = True, fuse_gate_up: bool = True, fuse_add_norm: bool = True, fuse_qk_rope: bool = True,
                 fuse_b1_attention: bool = True, long_attention: bool = True):
        self.enabled = enabled
        self.fuse_b1_attention = fuse_b1_attention
        self.long_attention = long_attention          # the same blocks at 81 .. 4096 tokens: ops.RotaryCausalAttentionFn
        self.b1_attn: List[torch.nn.Module] = []     # attention blocks whose rotary + attention run as one launch in the batch-1 gradient pass
        self._qkv_whole = {}                         # id(attention block) -> x -> the fused q/k/v product (made at __enter__)
        self.fuse_qk_rope = fuse_qk_rope
        self._rope_norms = {}                        # id(q_norm / k_norm module) of attention blocks whose forward rotates right behind them
        self.admitted = {}
        self.refused = {}                            # fusion -> why it was NOT admitted on this model (logged at construction)
        self._pending = {}                           # id(tensor) -> (tensor, weight, eps, gemma): a head norm deferred into the rotary launch
        self.weight_copies = weight_copies
        self.fuse_gate_up = fuse_gate_up
        self.fuse_add_norm = fuse_add_norm
        self.layers: List[Tuple[torch.nn.Module, str, torch.nn.Module]] = []   # (decoder layer, kind, the norm that reads its output)
        self._norm_info = {}                         # id(norm module) -> (eps, gemma)
        self._stash = {}                             # id(norm module) -> (sum tensor, its norm): handed over by the layer in front
        # scoring forwards keep the target-predicting rows only (logits_to_keep): with `keep_rows` set (an index over dim 1)
        # the LAST decoder layer gathers them right behind its attention block, so its MLP, the final norm and the head run
        # on those rows alone; `kept` says the layer did (hf_adapter.HFAdapter._logits_of_rows reads it)
        self.keep_rows = None
        self.kept = False
        self._last_layers = set()                    # id(the last decoder layer) of stacks whose forward ends `norm(layers(...))`
        # tensor-parallel gradient pass (EngineOptions.tp_gradient): (rank, world, process group) while a pass runs with
        # every decoder projection cut over the ranks -- q/k/v/gate/up by output rows (whole heads), o/down by input
        # columns, two all-reduces per layer and direction; set before entering the context, None otherwise
        self.tp = None
        self._tp_roles = {}
        self.gemm_probe = None                       # measurement hook for products no nn.Linear module owns
        # products just above a whole number of tile rounds as two library calls (round_cut): measured with the shipped GEMM
        # selection, whose picks for these shapes ARE 256 x 256-tile kernels -- the engine turns it off when that table is not in use
        self.round_split = True
        cache = _COPY_CACHES.get(model)
        if cache is None:
            cache = _COPY_CACHES[model] = _CopyCache()
        self._copies: _CopyCache = cache             # ("wt"|"wqkv"|"wgu"|"gu_t", id(module)) -> derived weight copy
        self.qkv: List[torch.nn.Module] = []        # attention blocks whose q/k/v projections run as one GEMM
        self.norms: List[Tuple[torch.nn.Module, float, bool]] = []
        self.mlps: List[torch.nn.Module] = []
        self.linears: List[torch.nn.Module] = []
        self.rope_modules = []
        self._saved_rope = {}
        self._gu_ok = {}
        self.depth = 0
        self._patches = None                       # (module, replacement forward) pairs, made at the first __enter__
        self._linear_patches = None
        if not enabled:
            return
        files = set()
        norm_rope_blocks = []
        for m in model.modules():
            cls = type(m).__name__
            w = getattr(m, "weight", None)
            if fuse_qkv and all(isinstance(getattr(m, n, None), torch.nn.Linear) and type(getattr(m, n)) is torch.nn.Linear
                                and getattr(m, n).bias is None for n in ("q_proj", "k_proj", "v_proj")) \
                    and m.q_proj.in_features == m.k_proj.in_features == m.v_proj.in_features \
                    and m.q_proj.weight.dtype in (torch.bfloat16, torch.float16) \
                    and not hasattr(m, "q_norm"):      # per-head norms (Gemma-3) want dense projection outputs
                self.qkv.append(m)
            if fuse_qk_rope and hasattr(m, "q_norm") and hasattr(m, "k_norm") and _norms_then_rope(m):
                norm_rope_blocks.append(m)
            if hasattr(m, "q_proj") or hasattr(m, "gate_proj"):
                for name in _TP_COLUMN + _TP_ROW:
                    lin = getattr(m, name, None)
                    if isinstance(lin, torch.nn.Linear) and hasattr(m, "layer_idx" if name in ("q_proj", "k_proj", "v_proj", "o_proj") else "down_proj"):
                        self._tp_roles[id(lin)] = (lin, "column" if name in _TP_COLUMN else "row")
            if weight_copies and (hasattr(m, "q_proj") or hasattr(m, "gate_proj")):
                # bias-free projections of the decoder layers (attention and MLP blocks)
                for name in ("q_proj", "k_proj", "v_proj", "o_proj", "gate_proj", "up_proj", "down_proj"):
                    lin = getattr(m, name, None)
                    if isinstance(lin, torch.nn.Linear) and lin.bias is None and type(lin) is torch.nn.Linear:
                        self.linears.append(lin)
            if cls.endswith("RMSNorm") and torch.is_tensor(w) and w.dim() == 1 and _eps_of(m) is not None:
                self.norms.append((m, _eps_of(m), cls.startswith("Gemma")))
            elif all(hasattr(m, a) for a in ("gate_proj", "up_proj", "down_proj", "act_fn")) and \
                    self._act_code(m.act_fn) is not None:
                self.mlps.append(m)
            files.add(type(m).__module__)
        if next((True for p in model.parameters() if p.is_cuda), False):
            ops.gemm_workspace_for_graphs(next(p for p in model.parameters() if p.is_cuda).device)   # before any capture needs it
        self._norm_info = {id(m): (eps, gemma) for m, eps, gemma in self.norms}
        if fuse_add_norm:
            self.layers = _decoder_layers(model, self._norm_info)
            self._last_layers = _last_layers(model, self.layers)
        for f in files:
            mod = sys.modules.get(f)
            if mod is not None and f.rsplit(".", 1)[-1] in _ROPE_FILES and hasattr(mod, "apply_rotary_pos_emb"):
                self.rope_modules.append(mod)
        # a head norm may only be deferred into the rotary launch of a block whose modelling file's apply_rotary_pos_emb
        # IS patched here: anywhere else nobody would pick the un-normalised tensor up
        for m in norm_rope_blocks:
            if sys.modules.get(type(m).__module__) in self.rope_modules:
                self._rope_norms[id(m.q_norm)] = m.q_norm
                self._rope_norms[id(m.k_norm)] = m.k_norm
        if fuse_b1_attention:
            for m in self.qkv:
                cfg = getattr(m, "config", None)
                if cfg is not None and hasattr(m, "o_proj") and hasattr(m, "scaling") and getattr(m, "head_dim", 0) == 128 \
                        and getattr(cfg, "num_attention_heads", 0) == (getattr(cfg, "num_key_value_heads", None) or cfg.num_attention_heads) \
                        and m.q_proj.out_features == cfg.num_attention_heads * 128 and _plain_rotary_attention(m):
                    self.b1_attn.append(m)
        kinds = sorted({k for _, k, _ in self.layers})
        # what was admitted (the source-text checks above turn a fusion off silently on a transformers upgrade or a
        # .pyc-only install: the engine logs this and bench.py prints it with the engine state)
        self.refused = self._refusals(model, fuse_qkv, fuse_add_norm, fuse_b1_attention, norm_rope_blocks)
        self.admitted = dict(rmsnorms=len(self.norms), gated_mlps=len(self.mlps), fused_qkv_blocks=len(self.qkv),
                             transposed_copy_projections=len(self.linears), add_norm_layers=len(self.layers),
                             layer_kinds=kinds, rotary_files=[m.__name__.rsplit(".", 1)[-1] for m in self.rope_modules],
                             b1_attention_blocks=len(self.b1_attn),
                             qk_norm_in_rotary_blocks=len(self._rope_norms) // 2,
                             qk_norm_blocks_not_admitted=len(norm_rope_blocks) - len(self._rope_norms) // 2,
                             last_layer_keeps_rows=len(self._last_layers))

    def _refusals(self, model, fuse_qkv, fuse_add_norm, fuse_b1_attention, norm_rope_blocks) -> dict:
        """Which fast paths this model did NOT get, and why -- the admission checks read HuggingFace SOURCE TEXT and
        attribute names, so a transformers upgrade or a .pyc-only install turns a fusion off without an error; this is what
        says so (logged once at construction, printed with bench.py's engine state)."""
        import inspect
        out = {}

        def source_of(cls):
            try:
                inspect.getsource(cls.forward)
                return True
            except (OSError, TypeError):
                return False

        stacks = [m for m in model.modules() if isinstance(getattr(m, "layers", None), torch.nn.ModuleList) and len(m.layers)
                  and hasattr(m.layers[0], "self_attn") and hasattr(m.layers[0], "mlp")]
        if fuse_add_norm and stacks and not self.layers:
            cls = type(stacks[0].layers[0])
            out["add_norm_layers"] = (f"{cls.__name__}.forward: source not available (.pyc-only install?)" if not source_of(cls)
                                      else f"{cls.__name__}.forward is not, statement for statement, the llama / gemma3 residual structure "
                                           "the fused layer forward restates (another transformers version or model family)")
        attn_blocks = [m for m in model.modules() if all(hasattr(m, n) for n in ("q_proj", "k_proj", "v_proj", "o_proj")) and hasattr(m, "layer_idx")]
        if fuse_qkv and attn_blocks and not self.qkv:
            a = attn_blocks[0]
            why = ("projections carry a bias" if a.q_proj.bias is not None else
                   "the block has per-head q/k norms (dense projection outputs wanted)" if hasattr(a, "q_norm") else
                   f"weights are {a.q_proj.weight.dtype} (16-bit only)" if a.q_proj.weight.dtype not in (torch.bfloat16, torch.float16) else
                   "q/k/v are not plain nn.Linear modules over one input width")
            out["fused_qkv_blocks"] = why
        if fuse_b1_attention and self.qkv and len(self.b1_attn) < len(self.qkv):
            a = next(m for m in self.qkv if m not in self.b1_attn)
            cfg = getattr(a, "config", None)
            heads = getattr(cfg, "num_attention_heads", 0)
            kv = getattr(cfg, "num_key_value_heads", None) or heads
            why = (f"{type(a).__name__}.forward: source not available (.pyc-only install?)" if not source_of(type(a)) else
                   f"head width {getattr(a, 'head_dim', None)} (128 only)" if getattr(a, "head_dim", 0) != 128 else
                   f"grouped key/value heads ({heads} on {kv})" if heads != kv else
                   f"{type(a).__name__}.forward is not projections -> rotary -> attention function -> o_proj with nothing else touching q, k, v")
            out["b1_attention_blocks"] = f"{len(self.qkv) - len(self.b1_attn)} of {len(self.qkv)} blocks: {why}"
        elif fuse_b1_attention and attn_blocks and not self.qkv:
            out["b1_attention_blocks"] = "needs the fused q/k/v product (see fused_qkv_blocks)"
        missed = len(norm_rope_blocks) - len(self._rope_norms) // 2
        if missed:
            out["qk_norm_in_rotary_blocks"] = f"{missed} blocks: their modelling file's apply_rotary_pos_emb is not one this module patches"
        files = {type(m).__module__.rsplit(".", 1)[-1] for m in attn_blocks}
        unknown = sorted(f for f in files if f not in _ROPE_FILES and hasattr(sys.modules.get(type(attn_blocks[0]).__module__), "apply_rotary_pos_emb"))
        if unknown:
            out["rotary_files"] = f"{unknown}: not a modelling file whose rotary convention is known (full-head, rotate_half)"
        return out

    @staticmethod
    def _act_code(act_fn):
        """0 for SiLU, 1 for gelu(approximate="tanh") through its C implementation, None otherwise."""
        name = type(act_fn).__name__
        if name in ("SiLU", "SiLUActivation"):
            return ops.ACT_SILU
        if name in ("GELUTanh", "PytorchGELUTanh"):
            inner = getattr(act_fn, "act", None)          # HF: functools.partial(F.gelu, approximate="tanh")
            if inner is None or getattr(inner, "func", None) is torch.nn.functional.gelu:
                return ops.ACT_GELU_TANH
        if isinstance(act_fn, torch.nn.GELU) and getattr(act_fn, "approximate", "none") == "tanh":
            return ops.ACT_GELU_TANH
        return None

    # -- the replacements ----------------------------------------------------------------
    @staticmethod
    def _usable(x: torch.Tensor) -> bool:
        return x.is_cuda and x.dtype in _DTYPES

    @staticmethod
    def _tracking(*ts) -> bool:
        return torch.is_grad_enabled() and any(t.requires_grad for t in ts)

    # NOTE on weights: inside this context they are constants.  The engine borrows the model
    # read-only and calls torch.autograd.grad w.r.t. inputs only, so a norm weight that happens
    # to have requires_grad=True (HF's default after from_pretrained) gets no gradient here.

    def _norm_forward(self, m, eps, gemma, orig):
        def forward(x):
            hit = self._stash.pop(id(m), None)
            if hit is not None and hit[0] is x:
                return hit[1]                        # computed by the fused add + norm of the layer in front
            D = x.shape[-1]
            if not self._usable(x) or (D * x.element_size()) % 16 or D * x.element_size() > 16384 \
                    or m.weight.dtype != x.dtype:
                return orig(x)
            if x.dim() == 4 and not x.is_contiguous() and x.transpose(1, 2).is_contiguous():
                if id(m) in self._rope_norms and not self._tracking(x) and ops.qknorm_rope_ok(x):
                    # Gemma-3's q_norm / k_norm in the no-grad scoring forward: the rotation is the very next thing the
                    # attention block does with this tensor (checked from its source), so the norm rides in the rotary
                    # launch (bma_qknorm_rope2) -- x goes back as it is, with a note for apply_rotary_pos_emb
                    self._pending[id(x)] = (x, m.weight, eps, gemma)
                    return x
                # per-head q/k norm on a (B,H,L,Dh) VIEW of the projection's (B,L,H,Dh) output: rows
                # are rows in either order, so normalise in place of layout instead of copying
                return forward(x.transpose(1, 2)).transpose(1, 2)
            if self._tracking(x):
                # the weight is a constant to the fused backward (the engine only ever asks for
                # gradients w.r.t. inputs)
                return ops.RMSNormFn.apply(x, m.weight, eps, gemma)
            return ops.rmsnorm(x, m.weight, eps, gemma)
        return forward

    def _product(self, x, w):
        """A no-grad x @ w.T: two library calls when the row count sits just above whole tile rounds (two_calls), else one."""
        if self.round_split and self.tp is None:
            y = two_calls(x, w)
            if y is not None:
                return y
        return torch.nn.functional.linear(x, w)

    def _linear_forward(self, m, orig):
        """Gradient pass with a handful of rows: the backward product goes through a transposed
        copy of the (constant) weight so that it, too, streams weight rows along the reduction."""
        def forward(x):
            w = m.weight
            if self.round_split and self.tp is None and not self._tracking(x):
                y = two_calls(x, w)
                if y is not None:
                    return y
            if not (self._tracking(x) and x.is_cuda and x.dtype == w.dtype and x.dtype in (torch.bfloat16, torch.float16)
                    and x.numel() // x.shape[-1] <= SKINNY_ROWS):
                return orig(x)
            wt = self._copies.get(("wt", id(m)), (w,))
            if wt is None:
                if torch.cuda.is_current_stream_capturing():
                    return orig(x)
                with torch.no_grad():
                    wt = self._copies.put(("wt", id(m)), w.detach().t().contiguous(), (w,))
            return ops.FrozenLinearFn.apply(x, w, wt)
        return forward

    def _tp_linear(self, m, kind):
        """This rank's part of a decoder projection: `column` = its block of output rows (whole heads / its share of the
        MLP width), input replicated; `row` = its block of input columns, partial outputs summed over the ranks."""
        def forward(x):
            rank, world, group = self.tp
            w = m.weight
            if kind == "column":
                n = w.shape[0] // world
                return torch.nn.functional.linear(x, w[rank * n:(rank + 1) * n], None if m.bias is None else m.bias[rank * n:(rank + 1) * n])
            k = w.shape[1] // world
            y = torch.nn.functional.linear(x, w[:, rank * k:(rank + 1) * k])
            y = _TPReduce.apply(y, group)
            return y if m.bias is None else y + m.bias
        return forward

    def register_gemm_chain(self) -> int:
        """Tell ops.gemm_nt which weight FOLLOWS which in the batch-1 gradient pass (cross-product weight prefetch,
        bma_gemm_nt_next): forward, per decoder layer, fused q/k/v -> fused gate/up -> down -> the next layer's q/k/v;
        backward the transposed copies in reverse (down -> gate/up -> q/k/v -> the layer below's down).  o_proj sits
        between q/k/v and gate/up on the library and is stepped over.  Called once the derived copies exist (after the
        first eager pass, before a capture: a captured launch carries its successor's address).  Returns the links made."""
        if not (self.enabled and self.layers and self.weight_copies and ops.GEMM_NT_PREFETCH):
            return 0
        fwd, bwd = [], []
        for layer, _, _ in self.layers:
            attn, mlp = getattr(layer, "self_attn", None), getattr(layer, "mlp", None)
            down = getattr(getattr(mlp, "down_proj", None), "weight", None)
            down = down if (torch.is_tensor(down) and down.is_cuda) else None
            fwd += [self._copies.get(("wqkv", id(attn))), self._copies.get(("wgu", id(mlp))), down]
            bwd.append([self._copies.get(("wt", id(getattr(mlp, "down_proj", None)))), self._copies.get(("wgu_t", id(mlp))),
                        self._copies.get(("wqkv_t", id(attn)))])
        back = [w for trio in reversed(bwd) for w in trio]
        return ops.gemm_nt_chain(fwd) + ops.gemm_nt_chain(back)

    def tp_ok(self, world: int) -> bool:
        """Can the gradient pass be cut `world` ways?  Known layer structure (the fused layer forward carries the f
        operator), every projection width and head count divisible."""
        if not (self.enabled and self.layers and self._tp_roles):
            return False
        for lin, kind in self._tp_roles.values():
            if (lin.weight.shape[0] if kind == "column" else lin.weight.shape[1]) % world:
                return False
        for layer, _, _ in self.layers:
            cfg = getattr(layer.self_attn, "config", None)
            heads = getattr(cfg, "num_attention_heads", 0)
            kv = getattr(cfg, "num_key_value_heads", heads) or heads
            if heads % world or kv % world:
                return False
        return True

    def _qkv_forwards(self, attn):
        """q_proj / k_proj / v_proj of one attention block as ONE product against the concatenated
        weight (16-bit models): three N = 4096 GEMMs fill 3 x 4.4 of 15 tile rounds on 256 CUs, one
        N = 12288 GEMM 13.1 of 14; in the ~70-row gradient pass it is one weight stream instead of three
        and one input-gradient product instead of three plus two adds.  The first of the three calls does
        the product; the other two hand out their column slices of it (same input tensor, checked by
        identity), so HF's attention code is untouched."""
        mods = (attn.q_proj, attn.k_proj, attn.v_proj)
        origs = [type(m).forward.__get__(m) for m in mods]
        sizes = [m.out_features for m in mods]
        slot = {}

        srcs = tuple(m.weight for m in mods)

        def fused_weight():
            w = self._copies.get(("wqkv", id(attn)), srcs)
            if w is None:
                if torch.cuda.is_current_stream_capturing():
                    return None
                with torch.no_grad():
                    w = self._copies.put(("wqkv", id(attn)), torch.cat([m.weight.detach() for m in mods], dim=0).contiguous(), srcs)
            return w

        def whole(x):
            """The fused product itself, (.., q + k + v columns); None when it cannot be formed here."""
            w0 = mods[0].weight
            if not (x.is_cuda and x.dtype == w0.dtype and x.dim() >= 2):
                return None
            w = fused_weight()
            if w is None:
                return None
            if self._tracking(x):
                if not self.weight_copies or x.numel() // x.shape[-1] > SKINNY_ROWS:
                    return torch.nn.functional.linear(x, w)
                wt = self._copies.get(("wqkv_t", id(attn)), srcs)
                if wt is None:
                    if torch.cuda.is_current_stream_capturing():
                        return None
                    with torch.no_grad():
                        wt = self._copies.put(("wqkv_t", id(attn)), w.t().contiguous(), srcs)
                return ops.FrozenLinearFn.apply(x, w, wt)
            return self._product(x, w)

        def first(x):
            slot.clear()
            y = whole(x)
            if y is None:
                return origs[0](x)
            # one split node: its backward is a single concatenation of the three gradients (three
            # independent slices would each zero-fill a full-width buffer and add)
            parts = torch.split(y, sizes, dim=-1)
            slot["x"], slot["y"] = x, parts
            return parts[0]

        def later(i):
            def forward(x):
                y = slot.get("y")
                if y is None or slot.get("x") is not x:
                    return origs[i](x)
                out = y[i]
                if i == 2:
                    slot.clear()
                return out
            return forward

        return first, later(1), later(2), whole

    def _attn_forward(self, attn, orig):
        """The attention block between its (fused) q/k/v projection and o_proj as ONE launch forward and ONE backward --
        rotary embedding, causal attention, the head transposes either side -- for the batch-1 gradient pass over a
        short sequence (ops.B1AttentionFn; <= 80 tokens, 128-wide heads, no grouped heads).  Taken only when the engine
        has switched the block to its mask-free causal attention for this forward (prefix_attention.causal_b1: plain
        causal, no cache), autograd is recording, and the tensors qualify; HuggingFace's own forward otherwise."""
        cfg = attn.config
        heads = cfg.num_attention_heads

        def forward(hidden_states, position_embeddings=None, attention_mask=None, past_key_values=None, **kwargs):
            whole = self._qkv_whole.get(id(attn))
            if whole is not None and position_embeddings is not None and attention_mask is None and past_key_values is None \
                    and self.tp is None and getattr(cfg, "_attn_implementation", None) == "bma_causal_b1" \
                    and hidden_states.dim() == 3 and hidden_states.shape[0] == 1 and self._tracking(hidden_states):
                cos, sin = position_embeddings
                # (everything that can be known without the product is checked BEFORE it is formed: a refusal behind it
                # would run the projection twice -- the 643-row image pass did, for one round-4 measurement)
                if not (cos.requires_grad or sin.requires_grad) and cos.shape == (1, hidden_states.shape[1], 128) \
                        and hidden_states.shape[1] <= ops.B1_ATTENTION_MAX_TOKENS and cos.dtype == hidden_states.dtype \
                        and hidden_states.dtype in (torch.bfloat16, torch.float16) and hidden_states.is_cuda:
                    y = whole(hidden_states)
                    if y is not None and ops.b1_attention_ok(y, cos, heads, heads, 128):
                        out = ops.B1AttentionFn.apply(y, cos[0], sin[0], heads, float(attn.scaling))
                        return attn.o_proj(out), None
                # the long sequence (the image prompt): rotary + the hand-written causal attention pair, gradients written
                # straight into the fused projection's (ops.RotaryCausalAttentionFn)
                elif self.long_attention and not (cos.requires_grad or sin.requires_grad) \
                        and ops.B1_ATTENTION_MAX_TOKENS < hidden_states.shape[1] <= ops.CAUSAL_ATTENTION_MAX_TOKENS \
                        and cos.shape == (1, hidden_states.shape[1], 128) and cos.dtype == hidden_states.dtype \
                        and hidden_states.dtype in (torch.bfloat16, torch.float16) and hidden_states.is_cuda and ops.CAUSAL_ATTENTION:
                    y = whole(hidden_states)
                    if y is not None and ops.rotary_causal_attention_ok(y, cos, heads):
                        out = ops.RotaryCausalAttentionFn.apply(y, cos[0], sin[0], heads, float(attn.scaling))
                        return attn.o_proj(out), None
            return orig(hidden_states, position_embeddings=position_embeddings, attention_mask=attention_mask,
                        past_key_values=past_key_values, **kwargs)
        return forward

    def _gate_up_weight(self, m):
        """The chunk-interleaved [gate_proj; up_proj] weight of a 16-bit MLP (ops.interleave_gate_up), or None when
        the block does not qualify (or a capture is running and the copy does not exist yet)."""
        if not self.fuse_gate_up or self.tp is not None:
            return None
        g, u = m.gate_proj, m.up_proj
        ok = self._gu_ok.get(id(m))
        if ok is None:
            ok = all(type(l) is torch.nn.Linear and l.bias is None for l in (g, u)) \
                and g.weight.shape == u.weight.shape and g.weight.dtype == u.weight.dtype \
                and g.weight.dtype in (torch.bfloat16, torch.float16) and g.out_features % 8 == 0
            self._gu_ok[id(m)] = ok
        if not ok:
            return None
        srcs = (g.weight, u.weight)
        w = self._copies.get(("wgu", id(m)), srcs)
        if w is None:
            if torch.cuda.is_current_stream_capturing():
                return None
            with torch.no_grad():
                w = self._copies.put(("wgu", id(m)), ops.interleave_gate_up(g.weight.detach(), u.weight.detach()), srcs)
        return w

    def _mlp_forward(self, m, orig):
        act = self._act_code(m.act_fn)

        def forward(x):
            if not self._usable(x):
                return orig(x)
            w = self._gate_up_weight(m) if x.dtype == m.gate_proj.weight.dtype else None
            if w is not None:
                # gate_proj and up_proj as ONE product against their chunk-interleaved weights: N = 2I fills the
                # 256-CU tile rounds better than two N = I products (22.5 of 23 rounds instead of 2 x 11.25 of 12
                # at 17k rows; 3.0 of 4 instead of 2 x 1.5 of 2 at an eighth of them), one weight stream and one
                # input-gradient product in the ~70-row gradient pass; the gate kernel reads the alternating chunks
                if self._tracking(x):
                    if self.weight_copies and x.numel() // x.shape[-1] <= SKINNY_ROWS:
                        srcs = (m.gate_proj.weight, m.up_proj.weight)
                        wt = self._copies.get(("wgu_t", id(m)), srcs)
                        if wt is None and not torch.cuda.is_current_stream_capturing():
                            with torch.no_grad():
                                wt = self._copies.put(("wgu_t", id(m)), w.t().contiguous(), srcs)
                        y = ops.FrozenLinearFn.apply(x, w, wt) if wt is not None else torch.nn.functional.linear(x, w)
                    else:
                        y = torch.nn.functional.linear(x, w)
                    return m.down_proj(ops.SwiGLUInterleavedFn.apply(y, act))
                probe = self.gemm_probe
                if probe is not None and probe.on:       # bench.py's HIP-event bracket (profiled steps only)
                    t0 = probe.begin(x)
                    y = self._product(x, w)
                    probe.end(t0, "gate_up_proj", x, w.shape[0], w.shape[1])
                else:
                    y = self._product(x, w)
                return m.down_proj(ops.swiglu_il(y, act))
            g, u = m.gate_proj(x), m.up_proj(x)
            if (g.numel() * g.element_size()) % 16:
                return m.down_proj(m.act_fn(g) * u)
            if self._tracking(g, u):
                return m.down_proj(ops.SwiGLUFn.apply(g, u, act))
            return m.down_proj(ops.swiglu(g, u, act))
        return forward

    def _add_norm(self, residual, h, norm, pre=None):
        """(residual + h', norm(residual + h')) with h' = h, or pre(h) for Gemma-3's sandwich norm -- one launch when
        the tensors qualify, the eager add and the (patched) norm modules otherwise."""
        eps, gemma = self._norm_info[id(norm)]
        w = norm.weight
        ok = (h.shape == residual.shape and h.dtype == residual.dtype and ops.add_rmsnorm_ok(h, w)
              and h.is_contiguous() and residual.is_contiguous())
        if ok and self._tracking(residual, h):
            if pre is not None:
                h = pre(h)                           # (its own fused forward/backward; the add + norm pair below)
            return ops.AddRMSNormFn.apply(residual, h, w, eps, gemma)
        if ok and not torch.is_grad_enabled():
            if pre is None:
                return ops.add_rmsnorm(residual, h, w, eps, gemma)
            peps, pgemma = self._norm_info[id(pre)]
            # (fp16: the three-in-one variant compiles to a sum of squares that can differ from bma_rmsnorm's in the
            # last place -- 1 ulp on ~0.01 % of outputs; bf16 and fp32 are bit-identical -- so fp16 keeps two launches)
            if pgemma == gemma and pre.weight.dtype == h.dtype and h.dtype != torch.float16:
                return ops.add_rmsnorm(residual, h, w, eps, gemma, pre_weight=pre.weight, pre_eps=peps)
            return ops.add_rmsnorm(residual, pre(h), w, eps, gemma)
        s = residual + (h if pre is None else pre(h))
        return s, norm(s)

    def _layer_forward(self, layer, kind, next_norm):
        """The decoder layer's forward with each residual add fused into the norm that follows it -- the layer's own
        post-attention norm, and across the layer boundary the NEXT layer's input norm (or the stack's final norm),
        whose result is handed over through ``_stash`` and picked up by that norm's patched forward when it is called
        on the very tensor this layer returned.  Same statements as HuggingFace's forward (checked structurally when
        the layer was admitted, ``_decoder_layers``), same rounding points."""
        gem = kind == "gemma"
        last = id(layer) in self._last_layers

        def forward(hidden_states, *args, **kwargs):
            if args:                                  # HF calls its layers with keywords; anything else: their code
                return type(layer).forward(layer, hidden_states, *args, **kwargs)
            residual = hidden_states
            h = layer.input_layernorm(hidden_states)
            tp = self.tp
            if tp is not None:
                h = _TPCopy.apply(h, tp[2])          # replicated input of the column-parallel q/k/v
            h, _ = layer.self_attn(hidden_states=h, **kwargs)
            if self._pending:
                self._missed()
            if last and self.keep_rows is not None and not torch.is_grad_enabled():
                # everything behind the last attention block is row-wise: only the rows whose logits are asked for go on
                # (39 % of a 7B scoring forward's rows are suffix and template rows nobody reads: their share of this
                # layer's MLP -- the stack's largest products -- of the final norm and of nothing else)
                residual = residual.index_select(1, self.keep_rows)
                h = h.index_select(1, self.keep_rows)
                self.kept = True
            if gem:
                residual, h = self._add_norm(residual, h, layer.pre_feedforward_layernorm, pre=layer.post_attention_layernorm)
            else:
                residual, h = self._add_norm(residual, h, layer.post_attention_layernorm)
            if tp is not None:
                h = _TPCopy.apply(h, tp[2])          # ... and of gate/up
            h = layer.mlp(h)
            if next_norm is None:
                return residual + (layer.post_feedforward_layernorm(h) if gem else h)
            out, normed = self._add_norm(residual, h, next_norm, pre=layer.post_feedforward_layernorm if gem else None)
            self._stash[id(next_norm)] = (out, normed)
            return out
        return forward

    def _missed(self):
        """A deferred head norm nobody picked up: the result of this forward is wrong, so it must not be used -- but
        only this once: the deferral is off from here on and the same call goes through unfused."""
        self._pending.clear()
        n = len(self._rope_norms) // 2
        self._rope_norms.clear()
        self.admitted["qk_norm_in_rotary_blocks"] = 0
        raise DeferredNormMissed(f"bimodalattack_amd.fused: a deferred q/k norm was not picked up by the rotary embedding "
                                 f"({n} blocks had been admitted; the deferral is now off for this model)")

    def _rope(self, orig):
        def apply_rotary_pos_emb(q, k, cos, sin, *args, unsqueeze_dim=1, **kw):
            pq, pk = self._pending.pop(id(q), None), self._pending.pop(id(k), None)
            if pq is not None or pk is not None:
                fused_ok = (pq is not None and pk is not None and pq[0] is q and pk[0] is k and pq[2:] == pk[2:]
                            and not args and not kw and unsqueeze_dim == 1 and cos.dim() == 3 and cos.dtype == q.dtype
                            and cos.shape[-1] == q.shape[-1] and q.shape[0] == k.shape[0] and q.shape[2:] == k.shape[2:]
                            and not (cos.requires_grad or sin.requires_grad))
                if fused_ok:
                    return ops.qknorm_rope2(q, k, pq[1], pk[1], pq[2], pq[3], cos, sin, inplace=True)
                # not the shapes the fused launch takes after all: the deferred norms now, then the usual route
                if pq is not None:
                    q = ops.rmsnorm(q.transpose(1, 2), pq[1], pq[2], pq[3]).transpose(1, 2)
                if pk is not None:
                    k = ops.rmsnorm(k.transpose(1, 2), pk[1], pk[2], pk[3]).transpose(1, 2)
            ok = (self._usable(q) and not args and not kw and unsqueeze_dim == 1 and q.dim() == 4 and k.dim() == 4
                  and cos.dim() == 3 and cos.dtype == q.dtype and k.dtype == q.dtype and q.stride(3) == 1
                  and k.stride(3) == 1 and cos.shape[-1] == q.shape[-1]
                  and q.shape[-1] % (32 // q.element_size()) == 0
                  and (q.shape[-1] * q.element_size() // 16) <= 64
                  and ((q.shape[-1] * q.element_size() // 16) & (q.shape[-1] * q.element_size() // 16 - 1)) == 0
                  and all((s * q.element_size()) % 16 == 0 for s in q.stride()[:3] + k.stride()[:3]))
            if not ok:
                return orig(q, k, cos, sin, *args, unsqueeze_dim=unsqueeze_dim, **kw)
            same = q.shape[0] == k.shape[0] and q.shape[2] == k.shape[2] and q.shape[3] == k.shape[3]
            if self._tracking(q, k):
                if cos.requires_grad or sin.requires_grad:
                    return orig(q, k, cos, sin, *args, unsqueeze_dim=unsqueeze_dim, **kw)
                if same and self.fuse_add_norm:
                    return ops.RoPE2Fn.apply(q, k, cos, sin)          # q and k in one launch, forward and backward
                return ops.RoPEFn.apply(q, cos, sin), ops.RoPEFn.apply(k, cos, sin)
            if same and self.fuse_add_norm:
                return ops.rope2(q, k, cos, sin, inplace=True)
            ops.rope_(q, cos, sin)
            ops.rope_(k, cos, sin)
            return q, k
        return apply_rotary_pos_emb

    # -- install / remove ------------------------------------------------------------------
    def __enter__(self):
        self.depth += 1
        if not self.enabled or self.depth > 1:
            return self
        # (instance attributes written straight into __dict__: nn.Module.__setattr__ checks every value against its
        # parameter / buffer / submodule tables, ~2 us a time over ~450 modules, in front of an idle GPU)
        # the replacement forwards of norms, MLPs, plain projections and layers keep no tensors between calls (their
        # state lives on this object and is cleared on the way out), so they are made once; the fused q/k/v triples
        # hold the product they share and are made afresh
        if self._patches is None:
            self._patches = (
                [(m, self._norm_forward(m, eps, gemma, type(m).forward.__get__(m))) for m, eps, gemma in self.norms]
                + [(m, self._mlp_forward(m, type(m).forward.__get__(m))) for m in self.mlps]
                + [(layer, self._layer_forward(layer, kind, nxt)) for layer, kind, nxt in self.layers])
            self._linear_patches = [(m, self._linear_forward(m, type(m).forward.__get__(m))) for m in self.linears]
        for m, fn in self._patches:
            m.__dict__["forward"] = fn
        if self.tp is not None:
            for lin, kind in self._tp_roles.values():
                lin.__dict__["forward"] = self._tp_linear(lin, kind)
        else:
            for m, fn in self._linear_patches:
                m.__dict__["forward"] = fn
            for attn in self.qkv:                      # after the per-projection patches: these win for q/k/v
                fq, fk, fv, whole = self._qkv_forwards(attn)
                attn.q_proj.__dict__["forward"], attn.k_proj.__dict__["forward"], attn.v_proj.__dict__["forward"] = fq, fk, fv
                self._qkv_whole[id(attn)] = whole
            for attn in self.b1_attn:
                attn.__dict__["forward"] = self._attn_forward(attn, type(attn).forward.__get__(attn))
        for mod in self.rope_modules:
            self._saved_rope[mod] = mod.apply_rotary_pos_emb
            mod.apply_rotary_pos_emb = self._rope(mod.apply_rotary_pos_emb)
        return self

    def __exit__(self, *exc):
        self.depth -= 1
        if not self.enabled or self.depth > 0:
            return False
        for m, _, _ in self.norms:
            m.__dict__.pop("forward", None)
        for m in self.mlps:
            m.__dict__.pop("forward", None)
        for m in self.linears:
            m.__dict__.pop("forward", None)
        for attn in self.qkv:
            for m in (attn.q_proj, attn.k_proj, attn.v_proj):
                m.__dict__.pop("forward", None)
        for attn in self.b1_attn:
            attn.__dict__.pop("forward", None)
        self._qkv_whole.clear()
        for layer, _, _ in self.layers:
            layer.__dict__.pop("forward", None)
        for lin, _ in self._tp_roles.values():
            lin.__dict__.pop("forward", None)
        self._stash.clear()
        for mod, fn in self._saved_rope.items():
            mod.apply_rotary_pos_emb = fn
        self._saved_rope.clear()
        if self._pending:                          # (not when an exception is already on its way out)
            if exc[0] is None:
                self._missed()
            self._pending.clear()
        return False
