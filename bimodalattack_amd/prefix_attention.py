"""Shared-prefix attention for candidate scoring.

Every candidate of a step shares the prompt in front of the suffix (and, in joint mode,
the 576 image tokens): under causal attention those positions' keys/values are identical
in all candidates.  ``hf_adapter.build_prefix`` computes them once.  The stock HuggingFace
cache then COPIES them into every candidate's key/value tensors (``torch.cat`` per layer:
5.4 GB per layer at 512 candidates x 644 tokens) -- which is what this module removes.

A candidate's new tokens attend to
  (1) the shared prefix: ONE flash-attention launch with batch 1 and B*L queries against
      the (1,H,P,Dh) prefix keys/values (no mask: every prefix position is visible), and
  (2) themselves, causally: one flash launch with batch B,
and the two partial softmaxes are merged from their log-sum-exps by ``bma_attn_merge``.
Identical maths to attention over the concatenated sequence; nothing is copied per
candidate, so joint-mode scoring no longer needs small chunks.

Plumbing (transformers >= 4.48 attention interface): a registered attention function
``bma_shared_prefix`` + a no-op mask function, switched on for the duration of one forward;
a cache object that hands the new keys/values back untouched and reports the prefix length
so rotary positions are right.  Used only for model families whose text layers are plain
causal attention (llama / mistral / qwen2 / gemma3 modelling files) -- a sliding-window
layer IS plain causal attention while the whole sequence fits its window, which the engine
checks per call (``min_sliding_window``); anything else keeps the generic path.

Where the head size and dtype allow (16-bit, Dh <= 256, prefix <= FUSED_PREFIX_MAX keys) the
two library launches + merge collapse into ONE launch of ``bma_ragged_attention``: a padded
candidate block is the trivially ragged row list (first = 0, len = L), grouped key/value
heads are read in place (no ``repeat_kv`` copies: Gemma-3 has 8 query heads on 4).
"""

from __future__ import annotations

import contextlib
import os
from typing import List, Optional

import torch

from . import ops

# ragged scoring: attention in one MFMA launch (csrc/ragged_attention.hip) instead of the five-launch
# library route; prefixes longer than FUSED_PREFIX_MAX keys (the image of joint mode) are a separate partial
FUSED_RAGGED_ATTENTION = os.environ.get("BMA_FUSED_RAGGED_ATTENTION", "1") not in ("0", "false", "False")
FUSED_PREFIX_MAX = int(os.environ.get("BMA_FUSED_PREFIX_MAX", "128"))
# ... whose part is the hand-written flash kernel csrc/prefix_attention.hip (BMA_PREFIX_ATTENTION=0: the library's)
PREFIX_ATTENTION_KERNEL = os.environ.get("BMA_PREFIX_ATTENTION", "1") not in ("0", "false", "False")
NAME = "bma_shared_prefix"
_FAMILIES = ("modeling_llama", "modeling_mistral", "modeling_qwen2", "modeling_gemma3")
_ACTIVE: List["SharedPrefixKV"] = []
_REGISTERED = {"done": False}

try:
    from transformers.cache_utils import Cache as _CacheBase
except Exception:  # pragma: no cover
    _CacheBase = object


def fused_ragged_route(dtype, head_dim: int, heads: int, kv_heads: int, max_len: int) -> bool:
    """Will ``_ragged_attention`` take the one-launch MFMA kernel for these text layers?  The planner asks this BEFORE
    the forward (the library route needs the padded-block maps, the kernel route does not), with the very conditions
    ``ops.ragged_attention_ok`` checks on the tensors at run time plus the module switch."""
    return bool(FUSED_RAGGED_ATTENTION and dtype in (torch.bfloat16, torch.float16) and head_dim in (32, 64, 128, 256)
                and 0 < max_len <= ops.RAGGED_ATTN_MAX_LEN and kv_heads > 0 and heads % kv_heads == 0)


class RecordingKV(_CacheBase):
    """The cache handed to the model for the PREFIX pass: it keeps a reference to every
    layer's keys/values (after rotary embedding) and hands them straight back -- nothing is
    concatenated, nothing is allocated, no host value reaches the device, so the pass can
    be captured into a hipGraph (HF's DynamicCache cannot: replaying a capture that went
    through it faulted)."""

    def __init__(self, n_layers: int):
        try:
            super().__init__(layers=[])
        except Exception:
            pass
        self.k: List[Optional[torch.Tensor]] = [None] * n_layers
        self.v: List[Optional[torch.Tensor]] = [None] * n_layers
        self._sliding = [False] * n_layers

    @property
    def is_sliding(self):
        return self._sliding

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        self.k[layer_idx], self.v[layer_idx] = key_states, value_states
        return key_states, value_states

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return 0

    def get_mask_sizes(self, q, layer_idx: int = 0):
        return (int(q) if isinstance(q, int) else int(q.shape[0])), 0

    def get_max_cache_shape(self, layer_idx: int = 0) -> int:
        return -1

    def __len__(self):
        return len(self.k)


class SharedPrefixKV(_CacheBase):
    """Duck-typed HF cache holding the prefix keys/values of every layer, batch 1."""

    def __init__(self, base_cache):
        try:
            super().__init__(layers=[])
        except Exception:
            pass
        if isinstance(base_cache, RecordingKV):
            self.k, self.v = list(base_cache.k), list(base_cache.v)
            layers = self.k
        else:
            layers = base_cache.layers
            self.k = [l.keys for l in layers]
            self.v = [l.values for l in layers]
        self.P = int(self.k[0].shape[2])
        self._sliding = [False] * len(layers)
        self._rep = {}
        self.ragged: Optional["RaggedMaps"] = None     # set for a ragged scoring forward
        self.fused_ok = True                           # padded blocks through the one-launch MFMA kernel
        self.prefix_kernel_ok = True                   # long prefixes through csrc/prefix_attention.hip

    @property
    def is_sliding(self):                          # read-only property on the HF base class
        return self._sliding

    # -- what the HF forward asks of a cache --------------------------------------------
    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        return key_states, value_states            # the new tokens only: nothing is concatenated

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.P

    def get_mask_sizes(self, q, layer_idx: int = 0):
        q_len = int(q) if isinstance(q, int) else int(q.shape[0])
        return self.P + q_len, 0

    def get_max_cache_shape(self, layer_idx: int = 0) -> int:
        return -1

    def __len__(self):
        return len(self.k)

    def prefix(self, layer_idx: int, n_rep: int):
        """(K, V) of one layer, with kv heads repeated to the query heads when grouped."""
        if n_rep == 1:
            return self.k[layer_idx], self.v[layer_idx]
        hit = self._rep.get(layer_idx)
        if hit is None:
            hit = (self.k[layer_idx].repeat_interleave(n_rep, dim=1), self.v[layer_idx].repeat_interleave(n_rep, dim=1))
            self._rep[layer_idx] = hit
        return hit


class RaggedMaps:
    """Device copies of ``layout.ragged_plan``'s index maps for one scoring forward, plus the ids the
    forward embeds (the distinct candidates, then the parent): everything the host planned goes to the
    device in ONE copy out of a pinned staging buffer (`stage`: a dict the owner keeps between steps)."""

    _INT32 = ("flat", "cstart", "cfirst", "clen", "q_src", "kv_src")
    _INT64 = ("pos", "keep", "ids")

    def __init__(self, plan: dict, device, ids=None, stage: Optional[dict] = None):
        import numpy as np
        self.N, self.L, self.B2 = int(plan["N"]), int(plan["L"]), int(plan["m"]) + 1
        self.m_out = int(plan.get("m_out", plan["m"]))
        self.m_real = self.m_out
        arrays = {k: plan[k] for k in self._INT32 + ("pos", "keep") if plan.get(k) is not None}
        if ids is not None:
            arrays["ids"] = np.ascontiguousarray(ids, dtype=np.int64).reshape(-1)
        # int64 arrays first: every piece stays aligned to its element size
        order = [k for k in self._INT64 if k in arrays] + [k for k in self._INT32 if k in arrays]
        sizes = [arrays[k].size * (8 if k in self._INT64 else 4) for k in order]
        total = (sum(sizes) + 7) // 8 * 8
        pin = None if stage is None else stage.get("pin")
        if stage is not None and stage.get("event") is not None:
            stage["event"].synchronize()          # last step's copy has left the staging buffer (long since)
        if pin is None or pin.numel() < total:
            pin = torch.empty(max(total, 1 << 16), dtype=torch.uint8, pin_memory=torch.cuda.is_available())
            if stage is not None:
                stage["pin"] = pin
        host = pin.numpy()
        at, where = 0, {}
        for k, nb in zip(order, sizes):
            dt = np.int64 if k in self._INT64 else np.int32
            host[at:at + nb].view(dt)[:] = arrays[k].astype(dt, copy=False).reshape(-1)
            where[k] = (at, nb)
            at += nb
        self.nbytes = total
        dev = torch.empty(total, dtype=torch.uint8, device=device)
        dev.copy_(pin[:total], non_blocking=True)
        if stage is not None:
            # the staging buffer is rewritten next step: that copy must have left it by then
            stage["event"] = torch.cuda.Event()
            stage["event"].record(torch.cuda.current_stream(device))

        def view(k):
            if k not in where:
                return None
            a, nb = where[k]
            return dev[a:a + nb].view(torch.int64 if k in self._INT64 else torch.int32)

        self.flat, self.q_src, self.kv_src = view("flat"), view("q_src"), view("kv_src")
        self.pos = view("pos").unsqueeze(0)
        self.keep = view("keep")
        self.cstart, self.cfirst, self.clen = view("cstart"), view("cfirst"), view("clen")
        self.ids = None if ids is None else view("ids").view(-1, int(plan["n_opt"]))
        self.fused_ok = True           # a caller may clear it to force the library route (needs the padded-block maps)


_BIAS = {}


def _causal_bias(L: int, dtype, device) -> torch.Tensor:
    """(1,1,L,L) additive mask, -inf above the diagonal (rows padded to a multiple of 8)."""
    key = (L, dtype, str(device))
    b = _BIAS.get(key)
    if b is None:
        Lp = (L + 7) // 8 * 8
        full = torch.zeros((L, Lp), dtype=dtype, device=device)
        full[:, :L].masked_fill_(~torch.tril(torch.ones((L, L), dtype=torch.bool, device=device)), float("-inf"))
        b = full[:, :L].view(1, 1, L, L)
        _BIAS[key] = b
    return b


def _partial_attention(q, k, v, causal: bool, scale: float):
    """(out (B,H,L,Dh), lse (B,H,L) fp32) of softmax(q k^T * scale) v.

    The causal part goes through an explicit additive bias: on this ROCm build the is_causal
    variant of the attention kernel takes ~800 us at (512,32,45,45) where the biased one takes
    ~205 us (same outputs)."""
    if causal:
        B, H, L, _ = q.shape
        bias = _causal_bias(L, q.dtype, q.device).expand(B, H, L, L)
        out, lse = torch.ops.aten._scaled_dot_product_efficient_attention(q, k, v, bias, True, 0.0, False, scale=scale)[:2]
    else:
        # the unmasked part (every prefix key visible) through the same op: at (1, 32, 17152, 128) x 599 keys it
        # takes 338 us where the flash op takes 379 us, with identical outputs
        out, lse = torch.ops.aten._scaled_dot_product_efficient_attention(q, k, v, None, True, 0.0, False, scale=scale)[:2]
    return out, lse[..., : q.shape[2]]


def shared_prefix_attention(module, query, key, value, attention_mask=None, dropout: float = 0.0,
                            scaling: Optional[float] = None, **kwargs):
    """HF attention-interface function: returns (attn_output (B,L,H,Dh), None)."""
    kv = _ACTIVE[-1]
    B, H, L, Dh = query.shape
    n_rep = H // key.shape[1]
    scale = float(scaling) if scaling is not None else Dh ** -0.5
    if kv.ragged is not None:
        return _ragged_attention(kv, module.layer_idx, query, key, value, n_rep, scale), None
    if FUSED_RAGGED_ATTENTION and kv.fused_ok and kv.P <= FUSED_PREFIX_MAX and ops.ragged_attention_ok(query, key, L):
        out = _fused_block_attention(kv, module.layer_idx, query, key, value, scale)
        if out is not None:
            return out, None
    n_rep_prefix = n_rep
    if n_rep > 1:
        key, value = key.repeat_interleave(n_rep, dim=1), value.repeat_interleave(n_rep, dim=1)
    qm = query.transpose(1, 2)                      # (B,L,H,Dh): the projection's own memory order
    if not qm.is_contiguous():
        qm = qm.contiguous()
    q1 = qm.view(1, B * L, H, Dh).transpose(1, 2)   # (1,H,B*L,Dh) view, no copy
    o1, l1 = _prefix_partial(kv, module.layer_idx, q1, n_rep_prefix, scale)
    o2, l2 = _partial_attention(qm.transpose(1, 2), key, value, True, scale)
    o1 = o1.view(B, L, H, Dh)
    o2 = o2.transpose(1, 2).contiguous()
    out = ops.attn_merge(o1, o2, l1, l2.contiguous())
    return out, None


_BLOCKS = {}


def _block_maps(B: int, L: int, device):
    """start / first / len of B padded blocks of L tokens as a (trivially) ragged row list.  Entries are
    never evicted (a captured graph may hold their pointers)."""
    key = (B, L, str(device))
    m = _BLOCKS.get(key)
    if m is None:
        m = (torch.arange(B, dtype=torch.int32, device=device) * L, torch.zeros(B, dtype=torch.int32, device=device),
             torch.full((B,), L, dtype=torch.int32, device=device))
        _BLOCKS[key] = m
    return m


def _fused_block_attention(kv: "SharedPrefixKV", layer_idx: int, query, key, value, scale: float):
    """Shared prefix + causal self attention of B padded candidate blocks in one launch of the MFMA kernel
    (csrc/ragged_attention.hip); (B,L,H,Dh) output, or None when the tensors are not views of the projections'
    (B,L,heads,Dh) memory (then the library route runs)."""
    B, H, L, Dh = query.shape
    Hk = key.shape[1]

    def rows(t, heads):
        m = t.transpose(1, 2)                       # (B,L,heads,Dh)
        if not m.is_contiguous():
            return None
        return m.view(1, B * L, heads, Dh).transpose(1, 2)     # (1,heads,B*L,Dh) view, no copy

    q, k, v = rows(query, H), rows(key, Hk), rows(value, Hk)
    if q is None or k is None or v is None:
        return None
    start, first, length = _block_maps(B, L, query.device)
    if kv.P:
        Kp, Vp = kv.prefix(layer_idx, 1)
    else:
        Kp = Vp = None
    out = ops.ragged_attention(q, k, v, Kp, Vp, start, first, length, L, scale)
    return out.view(B, L, H, Dh)


def _prefix_partial(kv: "SharedPrefixKV", layer_idx: int, query, n_rep: int, scale: float):
    """(o1 (N,H,Dh), lse1 (H,N)) of every row of `query` (1,H,N,Dh) against the shared prefix, no mask: the
    hand-written flash kernel (grouped heads in place, output already in row-list order), else the library's
    efficient attention + a transpose copy."""
    _, H, N, Dh = query.shape
    if PREFIX_ATTENTION_KERNEL and kv.prefix_kernel_ok:
        Kp, Vp = kv.prefix(layer_idx, 1)
        if ops.prefix_attention_ok(query, Kp):
            return ops.prefix_attention(query, Kp, Vp, scale)
    Kp, Vp = kv.prefix(layer_idx, n_rep)
    o1, l1 = _partial_attention(query, Kp, Vp, False, scale)
    return o1.transpose(1, 2).reshape(N, H, Dh).contiguous(), l1.reshape(H, N).contiguous()


def _rows(t: torch.Tensor) -> torch.Tensor:
    """(1,H,N,Dh) view of a projection's (1,N,H,Dh) output -> the contiguous (N,H,Dh) row list."""
    r = t.transpose(1, 2)
    return (r if r.is_contiguous() else r.contiguous())[0]


def _ragged_attention(kv: SharedPrefixKV, layer_idx: int, query, key, value, n_rep: int, scale: float):
    """Attention of the ragged row list (layout.ragged_plan): every computed token against the
    shared prefix (one batch-1 launch over all N rows), plus causal attention inside the padded
    (B2,L) block whose slots are filled from the row list -- a candidate's slots in front of its
    first replaced position hold its PARENT's keys/values, which are rows of the same forward."""
    rg = kv.ragged
    _, H, N, Dh = query.shape
    Hk = key.shape[1]
    if FUSED_RAGGED_ATTENTION and rg.fused_ok and ops.ragged_attention_ok(query, key, rg.L):
        # one launch: prefix + parent + own keys per (candidate, head) on the matrix cores
        # (csrc/ragged_attention.hip).  A long prefix (the image in joint mode) keeps the library
        # flash kernel for its part and is merged in the kernel's epilogue.
        if kv.P <= FUSED_PREFIX_MAX:
            Kp, Vp = kv.prefix(layer_idx, 1)
            out = ops.ragged_attention(query, key, value, Kp, Vp, rg.cstart, rg.cfirst, rg.clen, rg.L, scale)
        else:
            o1, l1 = _prefix_partial(kv, layer_idx, query, n_rep, scale)
            out = ops.ragged_attention(query, key, value, None, None, rg.cstart, rg.cfirst, rg.clen, rg.L, scale,
                                       o1=o1, lse1=l1)
        return out.unsqueeze(0)
    if rg.q_src is None:
        raise RuntimeError("ragged maps were built without the padded-block maps the library attention route needs")
    q_rows, k_rows, v_rows = _rows(query), _rows(key), _rows(value)
    Kp, Vp = kv.prefix(layer_idx, n_rep)
    o1, l1 = _partial_attention(q_rows.unsqueeze(0).transpose(1, 2), Kp, Vp, False, scale)
    o1 = o1.transpose(1, 2).reshape(N, H, Dh).contiguous()
    qp = ops.gather_rows(q_rows, rg.q_src).view(rg.B2, rg.L, H, Dh).transpose(1, 2)
    kp = ops.gather_rows(k_rows, rg.kv_src).view(rg.B2, rg.L, Hk, Dh).transpose(1, 2)
    vp = ops.gather_rows(v_rows, rg.kv_src).view(rg.B2, rg.L, Hk, Dh).transpose(1, 2)
    if n_rep > 1:
        kp, vp = kp.repeat_interleave(n_rep, dim=1), vp.repeat_interleave(n_rep, dim=1)
    o2, l2 = _partial_attention(qp, kp, vp, True, scale)
    o2 = o2.transpose(1, 2).contiguous()
    out = ops.attn_merge_rows(o1, o2, l1.reshape(H, N).contiguous(), l2.contiguous(), rg.flat)
    return out.unsqueeze(0)                         # (1,N,H,Dh)


NAME_B1 = "bma_causal_b1"
B1_BACKENDS: list = []          # tests / A-B runs: an explicit backend priority list for the batch-1 causal attention
if os.environ.get("BMA_B1_FLASH_FIRST") == "1":
    from torch.nn.attention import SDPBackend as _B
    B1_BACKENDS = [_B.FLASH_ATTENTION, _B.EFFICIENT_ATTENTION, _B.MATH]


def causal_b1_attention(module, query, key, value, attention_mask=None, dropout: float = 0.0,
                        scaling: Optional[float] = None, **kwargs):
    """HF attention-interface function for the batch-1 GRADIENT pass (no cache, plain causal): the library's
    attention asked for `is_causal` instead of being handed a mask tensor.  HuggingFace's sdpa path builds a
    (1,1,S,S) mask for `inputs_embeds` calls, which routes the 643-token image prompt to the masked
    efficient-attention kernels; the causal flash variant does the same maths 3.1 ms per pass faster (forward
    + backward, 32 layers).  Measured against it and dropped: explicit scores (library batched products with
    fp32 scores + hand-written causal-softmax row kernels), 1.8 ms slower than this per pass."""
    B, H, S, Dh = query.shape
    scale = float(scaling) if scaling is not None else Dh ** -0.5
    out = _own_causal(query, key, value, scale, dropout)
    if out is not None:
        return out, None
    n_rep = H // key.shape[1]
    if n_rep > 1:
        key, value = key.repeat_interleave(n_rep, dim=1), value.repeat_interleave(n_rep, dim=1)
    # backend order: at the 643 tokens of the image prompt the efficient kernels' forward + backward pair measures
    # 154 us against 184 us for the flash pair (MI355X, 32 heads x 128); the others are within 5 % either way
    from torch.nn.attention import SDPBackend, sdpa_kernel
    with sdpa_kernel(B1_BACKENDS or [SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH],
                     set_priority=True):
        out = torch.nn.functional.scaled_dot_product_attention(query, key, value, is_causal=S > 1, scale=scale)
    return out.transpose(1, 2).contiguous(), None


# 256-wide and grouped-query heads on the hand-written pair (round 5; BMA_OWN_WIDE_HEADS=0: the library's flash kernels behind
# repeated copies of k / v, as before)
OWN_WIDE_HEADS = os.environ.get("BMA_OWN_WIDE_HEADS", "1") not in ("0", "false", "False")


def _own_causal(query, key, value, scale: float, dropout: float, causal: bool = True):
    """(1, Lq, H, Dh) through the hand-written attention pair (csrc/causal_attention.hip: forward 18 us and backward 53 us
    at 643 tokens x 32 heads x 128, where the library's pair and its helper launches take ~150), or None when the call is not
    its shape: batch 1, heads of 64, 72, 128 or 256 (grouped queries included: Gemma-3's decoder, 8 heads over 4 of 256 -- no
    repeated copies of k / v, the group's sum inside the backward launch), 16-bit, no dropout; causal with the queries as the
    last Lq of the Lk key positions, or every key visible (a vision tower)."""
    from . import ops
    if not ops.CAUSAL_ATTENTION or dropout or query.shape[0] != 1 or query.shape[1] % key.shape[1] \
            or query.shape[3] not in ((64, 72, 128, 256) if OWN_WIDE_HEADS else (64, 72, 128)) \
            or (query.shape[1] != key.shape[1] and not OWN_WIDE_HEADS):
        return None
    # (squeeze, not [0]: the backward of an index is a zero fill plus a copy per operand, of a squeeze nothing)
    q3, k3, v3 = query.squeeze(0).transpose(0, 1), key.squeeze(0).transpose(0, 1), value.squeeze(0).transpose(0, 1)
    if not ops.causal_attention_ok(q3, k3, v3):
        return None
    if torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad):
        return ops.CausalAttentionFn.apply(q3, k3, v3, scale, causal).unsqueeze(0)
    return ops.causal_attention(q3, k3, v3, scale, causal)[0].unsqueeze(0)


NAME_TAIL = "bma_tail_grad"


class GradPrefixKV(_CacheBase):
    """The prefix keys/values of every layer WITH their autograd history (a RecordingKV filled under
    `torch.enable_grad()`), handed to the forward of the tokens behind the prefix in the gradient pass: the
    loss then back-propagates through them into whatever the prefix was computed from (the image)."""

    def __init__(self, rec: RecordingKV):
        try:
            super().__init__(layers=[])
        except Exception:
            pass
        self.k, self.v = list(rec.k), list(rec.v)
        self.P = int(self.k[0].shape[2])
        self._sliding = [False] * len(self.k)
        self._bias = {}

    @property
    def is_sliding(self):
        return self._sliding

    def update(self, key_states, value_states, layer_idx, cache_kwargs=None):
        return key_states, value_states            # the new tokens only: the attention function concatenates

    def get_seq_length(self, layer_idx: int = 0) -> int:
        return self.P

    def get_mask_sizes(self, q, layer_idx: int = 0):
        q_len = int(q) if isinstance(q, int) else int(q.shape[0])
        return self.P + q_len, 0

    def get_max_cache_shape(self, layer_idx: int = 0) -> int:
        return -1

    def __len__(self):
        return len(self.k)

    def bias(self, L: int, dtype, device) -> torch.Tensor:
        """(1,1,L,P+L) additive mask: every prefix key, causal among the L new tokens."""
        key = (L, dtype, str(device))
        if key not in self._bias:
            b = torch.zeros((L, self.P + L), dtype=dtype, device=device)
            b[:, self.P:] = torch.full((L, L), float("-inf"), dtype=dtype, device=device).triu(1)
            self._bias[key] = b.view(1, 1, L, self.P + L)
        return self._bias[key]


def tail_grad_attention(module, query, key, value, attention_mask=None, dropout: float = 0.0,
                        scaling: Optional[float] = None, **kwargs):
    """HF attention-interface function for the tokens behind a prefix whose keys/values carry autograd history
    (gradient pass with the scoring prefix reused, attack.py): library attention over [prefix ; new] keys, every
    operand differentiable.  45 queries against 644 keys: the cost is in the launches, not the arithmetic."""
    kv: GradPrefixKV = _ACTIVE[-1]
    B, H, L, Dh = query.shape
    scale = float(scaling) if scaling is not None else Dh ** -0.5
    pk, pv = kv.k[module.layer_idx], kv.v[module.layer_idx]
    K = torch.cat([pk.expand(B, -1, -1, -1), key], dim=2)
    V = torch.cat([pv.expand(B, -1, -1, -1), value], dim=2)
    out = _own_causal(query, K, V, scale, dropout)                # the queries are the last L of the P + L positions
    if out is not None:
        return out, None
    n_rep = H // K.shape[1]
    if n_rep > 1:
        K, V = K.repeat_interleave(n_rep, dim=1), V.repeat_interleave(n_rep, dim=1)
    out = torch.nn.functional.scaled_dot_product_attention(query, K, V, attn_mask=kv.bias(L, query.dtype, query.device),
                                                           scale=scale)
    return out.transpose(1, 2).contiguous(), None


NAME_VIS = "bma_padded_heads"
PAD_HEADS_MIN_TOKENS = int(os.environ.get("BMA_PAD_HEADS_MIN_TOKENS", "1024"))
TOWER_EFFICIENT_FIRST = os.environ.get("BMA_TOWER_EFFICIENT_FIRST", "1") not in ("0", "false", "False")
OWN_TOWER_MAX_TOKENS = int(os.environ.get("BMA_OWN_TOWER_MAX_TOKENS", "1024"))
# ... and SigLIP's 72-wide heads (Gemma-3: 4096 tokens x 16 heads) on the same kernels, which take the real width and pad it
# to 96 in their LDS images only: no zero-padded copies of q / k / v, no slice of the output, and a backward of two launches
# where the library's padded pair took 719 us + the pad / slice copies per layer (profiles/r4_bench_gemma_joint_kernel_by_grid.txt)
OWN_TOWER_72 = os.environ.get("BMA_OWN_TOWER_72", "1") not in ("0", "false", "False")


def padded_width(head_dim: int, grad: bool) -> int:
    """Head width the vision attention is padded to (zero columns) before the library call; `head_dim` itself
    when padding does not pay.  Measured on MI355X at SigLIP's (1,16,4096,72), bf16, forward / backward µs:
    72 as is 289 / 1220 (the library pads to 80 inside); 96: 180 / 972; 128 (efficient backend): 214 / 856."""
    if head_dim % 32 == 0 or head_dim > 128:
        return head_dim
    return 128 if grad else -(-head_dim // 32) * 32


def padded_heads_attention(module, query, key, value, attention_mask=None, dropout: float = 0.0,
                           scaling: Optional[float] = None, is_causal: bool = False, **kwargs):
    """HF attention-interface function for a vision tower whose head width is not a multiple of 32 (SigLIP in
    Gemma-3: 72): q, k, v get zero columns up to a width the library's kernels are built for, the extra output
    columns are dropped.  Zero columns add nothing to q.k and produce zero outputs, so the result is the same
    attention; what changes is the kernel the library picks (see ``padded_width``)."""
    B, H, S, Dh = query.shape
    scale = float(scaling) if scaling is not None else Dh ** -0.5
    F = torch.nn.functional
    grad = torch.is_grad_enabled() and (query.requires_grad or key.requires_grad or value.requires_grad)
    W = padded_width(Dh, grad) if S >= PAD_HEADS_MIN_TOKENS else Dh
    causal = bool(is_causal) and attention_mask is None and S > 1
    if W == Dh:
        if attention_mask is None and B == 1 and S <= OWN_TOWER_MAX_TOKENS:
            # CLIP's 577 tokens x 16 heads of 64 at batch 1: the hand-written pair (latency-bound shapes; a tower of
            # thousands of tokens is arithmetic and stays with the library)
            own = _own_causal(query, key, value, scale, dropout, causal=causal)
            if own is not None:
                return own, None
        if TOWER_EFFICIENT_FIRST and grad and attention_mask is None and query.dtype in (torch.bfloat16, torch.float16):
            # CLIP's 577 tokens x 16 heads of 64 with autograd: the efficient backend's forward + backward pair measures
            # 89 us against 101 us for the flash pair (MI355X)
            from torch.nn.attention import SDPBackend, sdpa_kernel
            with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH], set_priority=True):
                out = F.scaled_dot_product_attention(query, key, value, dropout_p=dropout, is_causal=causal, scale=scale)
            return out.transpose(1, 2).contiguous(), None
        out = F.scaled_dot_product_attention(query, key, value, attn_mask=attention_mask, dropout_p=dropout,
                                             is_causal=causal, scale=scale)
        return out.transpose(1, 2).contiguous(), None
    if OWN_TOWER_72 and Dh == 72 and attention_mask is None and B == 1 and S <= ops.CAUSAL_ATTENTION_MAX_TOKENS:
        own = _own_causal(query, key, value, scale, dropout, causal=causal)
        if own is not None:
            return own, None
    q, k, v = (F.pad(t, (0, W - Dh)) for t in (query, key, value))
    from torch.nn.attention import SDPBackend, sdpa_kernel
    order = [SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH]
    with sdpa_kernel(order, set_priority=True):
        out = F.scaled_dot_product_attention(q, k, v, attn_mask=attention_mask, dropout_p=dropout,
                                             is_causal=causal, scale=scale)
    return out[..., :Dh].transpose(1, 2).contiguous(), None


def vision_configs(model) -> list:
    """Config objects of the vision tower's attention layers when their head width is one the padded route
    helps ([] otherwise): non-causal attention modules of the SigLIP / CLIP modelling files."""
    cfgs = {}
    for m in model.modules():
        if not (type(m).__name__.endswith("Attention") and hasattr(m, "q_proj") and hasattr(m, "head_dim")):
            continue
        if type(m).__module__.rsplit(".", 1)[-1] not in ("modeling_siglip", "modeling_clip"):
            continue
        if getattr(m, "is_causal", False) or hasattr(m, "layer_idx"):
            return []
        if padded_width(int(m.head_dim), True) != int(m.head_dim) or \
                (TOWER_EFFICIENT_FIRST and m.q_proj.weight.dtype in (torch.bfloat16, torch.float16)):
            cfgs[id(m.config)] = m.config          # (16-bit towers of any head width: the backend order under autograd)
    return list(cfgs.values())


def _no_mask(*args, **kwargs):
    return None


def register() -> bool:
    if _REGISTERED["done"]:
        return True
    try:
        from transformers.masking_utils import AttentionMaskInterface
        from transformers.modeling_utils import AttentionInterface
        AttentionInterface.register(NAME, shared_prefix_attention)
        AttentionMaskInterface.register(NAME, _no_mask)
        AttentionInterface.register(NAME_B1, causal_b1_attention)
        AttentionMaskInterface.register(NAME_B1, _no_mask)
        AttentionInterface.register(NAME_VIS, padded_heads_attention)
        AttentionInterface.register(NAME_TAIL, tail_grad_attention)
        AttentionMaskInterface.register(NAME_TAIL, _no_mask)
        _REGISTERED["done"] = True
    except Exception:
        return False
    return True


def _text_attention_layers(model):
    return [m for m in model.modules()
            if type(m).__name__.endswith("Attention") and hasattr(m, "layer_idx") and hasattr(m, "q_proj")]


def eligible_configs(model) -> list:
    """Config objects of the text attention layers when EVERY text layer is causal attention (full, or
    sliding-window -- see ``min_sliding_window``) from a known modelling file; [] otherwise."""
    cfgs = {}
    layers = _text_attention_layers(model)
    for m in layers:
        if type(m).__module__.rsplit(".", 1)[-1] not in _FAMILIES:
            return []
        cfgs[id(m.config)] = m.config
    if not layers:
        return []
    for c in cfgs.values():
        types = getattr(c, "layer_types", None)
        if types and any(t not in ("full_attention", "sliding_attention") for t in types):
            return []
        if getattr(c, "use_bidirectional_attention", False) or getattr(c, "attn_logit_softcapping", None):
            return []
    # the decoder stack's own config object drives mask creation: same object as the layers'
    return list(cfgs.values())


def min_sliding_window(model) -> Optional[int]:
    """Smallest sliding window among the text attention layers, None when there is none.  A sliding layer is
    plain causal attention as long as prefix + new tokens <= window; beyond that the generic path must run."""
    w = [int(m.sliding_window) for m in _text_attention_layers(model) if getattr(m, "sliding_window", None)]
    return min(w) if w else None


@contextlib.contextmanager
def active(configs: list, kv, name: str = NAME):
    """Switch the text layers to the shared-prefix attention (or, with ``name=NAME_TAIL`` and a GradPrefixKV, to the
    differentiable prefix + new-token attention of the gradient pass) for one forward."""
    old = [getattr(c, "_attn_implementation", None) for c in configs]
    _ACTIVE.append(kv)
    try:
        for c in configs:
            c._attn_implementation = name
        yield
    finally:
        _ACTIVE.pop()
        for c, o in zip(configs, old):
            c._attn_implementation = o


@contextlib.contextmanager
def causal_b1(configs: list, name: str = NAME_B1):
    """Switch the text layers to the mask-free causal attention for one batch-1 forward (no cache) -- or, with
    ``name=NAME_VIS``, the vision tower's layers to the padded-head attention."""
    old = [getattr(c, "_attn_implementation", None) for c in configs]
    try:
        for c in configs:
            c._attn_implementation = name
        yield
    finally:
        for c, o in zip(configs, old):
            c._attn_implementation = o
