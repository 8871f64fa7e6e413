"""Torch-facing wrappers over the C ABI (``include/bma.h``).

torch supplies device memory and the current HIP stream; every computation below
happens inside ``libbma_hip.so``.  Each wrapper checks on the host that operand
shapes, dtypes, devices and strides are what the kernel and its grid assume before
anything is launched.
"""

from __future__ import annotations

import os as _os
from typing import List, Optional, Sequence, Tuple

import torch

from . import native
from .native import BMA_BF16, BMA_F16, BMA_F32, BMA_SEG_GATHER, BMA_SEG_PERCAND, BMA_SEG_SHARED, BmaSegment, check, lib

_DT = {torch.float32: BMA_F32, torch.bfloat16: BMA_BF16, torch.float16: BMA_F16}


def _dt(t: torch.Tensor) -> int:
    try:
        return _DT[t.dtype]
    except KeyError:
        raise TypeError(f"unsupported dtype {t.dtype}; the kernels take float32, bfloat16, float16") from None


_ARCH_CHECKED = set()          # device indices whose architecture has been looked at


def check_arch(arch_name: str) -> None:
    """The library is built for gfx950 and only for it: MFMA 16x16x32, ds_read_b64_tr_b16, LDS-DMA loads, 160 KB of LDS --
    and `bma_gemm_nt`'s split-K hand-off relies on gfx950's cache behaviour for write-through (sc1) stores and loads, which
    the HIP memory model does not promise elsewhere.  Anything else is refused, loudly, at the first launch."""
    if not str(arch_name).split(":")[0] == "gfx950":
        raise RuntimeError(f"bimodalattack_amd: the HIP kernels are built and validated for gfx950 (MI355X) only; this device "
                           f"reports '{arch_name}'.  There is no fallback path.")


def _need_gpu(*ts: torch.Tensor) -> torch.device:
    dev = ts[0].device
    for t in ts:
        if not t.is_cuda:
            raise RuntimeError(
                "bimodalattack_amd kernels run on an AMD GPU only (tensor on %s); there is no CPU path" % t.device)
        if t.device != dev:
            raise RuntimeError("tensors on different devices")
    if dev.index not in _ARCH_CHECKED:
        check_arch(getattr(torch.cuda.get_device_properties(dev), "gcnArchName", "unknown"))
        _ARCH_CHECKED.add(dev.index)
    return dev


def _stream(dev: torch.device) -> int:
    return torch.cuda.current_stream(dev).cuda_stream


# ---------------------------------------------------------------------------
def linf_step(x: torch.Tensor, g: torch.Tensor, x0: torch.Tensor, eps: float, alpha: float,
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """clamp(clamp(x - (alpha*eps)*sign(g), x0-eps, x0+eps), 0, 1)  -- reference :1030-1037."""
    dev = _need_gpu(x, g, x0)
    for t in (x, g, x0):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != x.shape:
            raise ValueError("linf_step wants three contiguous float32 tensors of one shape")
    if out is None:
        out = torch.empty_like(x)
    elif out.dtype != torch.float32 or not out.is_contiguous() or out.shape != x.shape or out.device != dev:
        raise ValueError("bad out tensor")
    step = float(alpha * eps)  # the reference multiplies the two Python floats first (:1033)
    check("bma_linf_step", lib.bma_linf_step(x.data_ptr(), g.data_ptr(), x0.data_ptr(), x.numel(), float(eps), step,
                                             out.data_ptr(), _stream(dev)))
    return out


# ---------------------------------------------------------------------------
def ce_target(logits: torch.Tensor, labels: torch.Tensor, want_match: bool = False, want_dlogits: bool = False,
              grad_scale: float = 1.0):
    """Mean target cross-entropy per candidate.

    logits (B,T,V) in the model dtype, last dim contiguous (a strided view of a
    bigger tensor is fine); labels (T,) int64.  Returns
    ``(loss fp32 (B,), match int32 (B,) | None, dlogits (B,T,V) | None, row_loss fp32 (B,T))``.
    """
    if logits.dim() != 3:
        raise ValueError("logits must be (B,T,V)")
    dev = _need_gpu(logits, labels)
    B, T, V = logits.shape
    if labels.dtype != torch.int64 or labels.numel() != T or not labels.is_contiguous():
        raise ValueError("labels must be T contiguous int64 values")
    if B == 0:
        z = torch.empty(0, dtype=torch.float32, device=dev)
        return z, (torch.empty(0, dtype=torch.int32, device=dev) if want_match else None), \
            (torch.empty_like(logits) if want_dlogits else None), torch.empty(0, T, dtype=torch.float32, device=dev)
    if logits.stride(2) != 1 or logits.stride(1) < V or (B > 1 and logits.stride(0) < 0):
        logits = logits.contiguous()
    ws = torch.empty(3 * B * T, dtype=torch.float32, device=dev)
    assert ws.numel() * 4 == lib.bma_ce_target_ws_bytes(B, T)
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    match = torch.empty(B, dtype=torch.int32, device=dev) if want_match else None
    dlog = torch.empty((B, T, V), dtype=logits.dtype, device=dev) if want_dlogits else None
    check("bma_ce_target", lib.bma_ce_target(
        logits.data_ptr(), logits.stride(0) if B > 1 else T * logits.stride(1), logits.stride(1), labels.data_ptr(),
        B, T, V, _dt(logits), ws.data_ptr(), loss.data_ptr(), match.data_ptr() if want_match else None,
        dlog.data_ptr() if want_dlogits else None, float(grad_scale), _stream(dev)))
    return loss, match, dlog, ws[: B * T].view(B, T)


class TargetCrossEntropy(torch.autograd.Function):
    """mean CE over the target rows with the backward produced by the same kernel
    launch sequence (the gradient pass, reference :1006-1028)."""

    @staticmethod
    def forward(ctx, logits: torch.Tensor, labels: torch.Tensor) -> torch.Tensor:
        x = logits.detach()
        if x.dim() == 2:
            x = x.unsqueeze(0)
        loss, _, dlog, _ = ce_target(x, labels, want_dlogits=True)
        ctx.save_for_backward(dlog)
        ctx.in_shape = logits.shape
        return loss[0]

    @staticmethod
    def backward(ctx, grad_out):
        (dlog,) = ctx.saved_tensors
        return (dlog * grad_out.to(dlog.dtype)).view(ctx.in_shape), None


# ---------------------------------------------------------------------------
def build_mask_bits(not_allowed_ids: Optional[torch.Tensor], V: int, device) -> Optional[torch.Tensor]:
    """ceil(V/32) words, bit set = token not allowed.  Built once per attack."""
    if not_allowed_ids is None:
        return None
    import numpy as np

    idn = not_allowed_ids.to("cpu", torch.int64).numpy()
    idn = idn[(idn >= 0) & (idn < V)]      # the list covers range(tokenizer.vocab_size) <= V
    w = np.zeros((V + 31) // 32, dtype=np.uint32)
    np.bitwise_or.at(w, idn >> 5, np.uint32(1) << (idn & 31).astype(np.uint32))   # several ids share a word
    return torch.from_numpy(w.view(np.int32)).to(device)


def mask_topk(grad: torch.Tensor, mask_bits: Optional[torch.Tensor], k: int) -> torch.Tensor:
    """topk(-grad with not-allowed = +inf, k).indices, ordered (grad asc, id asc) -- reference :144-147."""
    if grad.dim() != 2:
        raise ValueError("grad must be (rows, V)")
    dev = _need_gpu(grad)
    rows, V = grad.shape
    if grad.stride(1) != 1 or grad.stride(0) < V:
        grad = grad.contiguous()
    if not (1 <= k <= V):
        raise ValueError(f"topk={k} out of range for vocabulary {V}")
    if mask_bits is not None:
        if mask_bits.device != dev or mask_bits.dtype != torch.int32 or mask_bits.numel() != (V + 31) // 32 \
                or not mask_bits.is_contiguous():
            raise ValueError("mask_bits must be ceil(V/32) contiguous int32 words on the gradient's device")
    out = torch.empty((rows, k), dtype=torch.int64, device=dev)
    nws = lib.bma_mask_topk_ws_bytes(rows, V, k)      # long rows are cut across workgroups: slice winners pass through it
    ws = torch.empty(nws // 8, dtype=torch.int64, device=dev) if nws else None
    check("bma_mask_topk", lib.bma_mask_topk(grad.data_ptr(), grad.stride(0), rows, V, _dt(grad),
                                             mask_bits.data_ptr() if mask_bits is not None else None, k,
                                             out.data_ptr(), ws.data_ptr() if ws is not None else None, _stream(dev)))
    return out


def rand_positions(rnd: torch.Tensor, n_replace: int) -> torch.Tensor:
    """argsort(rnd)[..., :n_replace] -- reference :150-154."""
    dev = _need_gpu(rnd)
    if rnd.dim() != 2 or rnd.dtype != torch.float32 or not rnd.is_contiguous():
        raise ValueError("rnd must be a contiguous float32 (B, n_opt) tensor")
    B, n_opt = rnd.shape
    out = torch.empty((B, n_replace), dtype=torch.int64, device=dev)
    check("bma_rand_positions", lib.bma_rand_positions(rnd.data_ptr(), B, n_opt, n_replace, out.data_ptr(), _stream(dev)))
    return out


def sample_scatter(ids: torch.Tensor, topk_idx: torch.Tensor, pos: torch.Tensor, rank: torch.Tensor) -> torch.Tensor:
    """ids.repeat(B,1).scatter_(1, pos, topk_idx[pos, rank]) -- reference :142, :156-162."""
    dev = _need_gpu(ids, topk_idx, pos, rank)
    n_opt = ids.numel()
    if topk_idx.dim() != 2 or topk_idx.shape[0] != n_opt or pos.shape != rank.shape or pos.dim() != 2:
        raise ValueError("shape mismatch")
    for t in (ids, topk_idx, pos, rank):
        if t.dtype != torch.int64 or not t.is_contiguous():
            raise ValueError("index tensors must be contiguous int64")
    B, n_rep = pos.shape
    k = topk_idx.shape[1]
    out = torch.empty((B, n_opt), dtype=torch.int64, device=dev)
    check("bma_sample_scatter", lib.bma_sample_scatter(ids.data_ptr(), topk_idx.data_ptr(), pos.data_ptr(),
                                                       rank.data_ptr(), B, n_opt, n_rep, k, out.data_ptr(),
                                                       _stream(dev)))
    return out


# ---------------------------------------------------------------------------
Segment = Tuple[str, Optional[torch.Tensor]]   # ("shared"|"percand"|"gather", tensor or None)


def splice(segments: Sequence[Segment], B: int, emb_weight: Optional[torch.Tensor] = None,
           ids: Optional[torch.Tensor] = None, emb_scale: float = 1.0,
           out: Optional[torch.Tensor] = None, rows: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Build (B,S,D) candidate embeddings -- reference :1112-1225.

    ``("shared", t)``: t is (L,D) or (1,L,D), broadcast to every candidate;
    ``("percand", t)``: t is (B,L,D); ``("gather", None)``: rows emb_weight[ids[b]] * emb_scale.

    With ``rows`` (int32 (N,): slots b*S + s of that block) only those rows are built, as an (N,D) row list
    (``bma_splice_rows``: ragged scoring never materialises the block).
    """
    if not segments or len(segments) > native.BMA_MAX_SEGS:
        raise ValueError(f"1..{native.BMA_MAX_SEGS} segments")
    ref = next((t for _, t in segments if t is not None), emb_weight)
    if ref is None:
        raise ValueError("nothing to splice")
    dev, dtype, D = _need_gpu(ref), ref.dtype, ref.shape[-1]
    arr = (BmaSegment * len(segments))()
    keep: List[torch.Tensor] = []
    S, n_opt, V = 0, 0, 0
    for i, (kind, t) in enumerate(segments):
        if kind == "gather":
            if emb_weight is None or ids is None:
                raise ValueError("gather segment needs emb_weight and ids")
            if emb_weight.dim() != 2 or emb_weight.shape[1] != D or emb_weight.dtype != dtype or \
                    not emb_weight.is_contiguous() or emb_weight.device != dev:
                raise ValueError("emb_weight must be a contiguous (V,D) table of the segments' dtype")
            if ids.dim() != 2 or ids.shape[0] != B or ids.dtype != torch.int64 or not ids.is_contiguous() \
                    or ids.device != dev:
                raise ValueError("ids must be contiguous int64 (B, n_opt) on the table's device")
            n_opt, V = ids.shape[1], emb_weight.shape[0]
            arr[i] = BmaSegment(None, n_opt, BMA_SEG_GATHER)
            S += n_opt
            continue
        if t.device != dev or t.dtype != dtype or t.shape[-1] != D:
            raise ValueError("segments must share device, dtype and width")
        if kind == "shared":
            if t.dim() == 3:
                if t.shape[0] != 1:
                    raise ValueError("shared segment must have batch 1")
                t = t[0]
            code = BMA_SEG_SHARED
            L = t.shape[0]
        elif kind == "percand":
            if t.dim() != 3 or t.shape[0] != B:
                raise ValueError("percand segment must be (B,L,D)")
            code = BMA_SEG_PERCAND
            L = t.shape[1]
        else:
            raise ValueError(f"unknown segment kind {kind!r}")
        t = t.contiguous()
        keep.append(t)
        arr[i] = BmaSegment(t.data_ptr() if L else None, L, code)
        S += L
    if rows is not None:
        if rows.dtype != torch.int32 or rows.dim() != 1 or not rows.is_contiguous() or rows.device != dev:
            raise ValueError("rows must be a contiguous int32 vector on the segments' device")
        if B <= 0 or S <= 0:
            raise ValueError("row-list splice of an empty block")
        N = rows.shape[0]
        if out is None:
            out = torch.empty((N, D), dtype=dtype, device=dev)
        elif out.shape != (N, D) or out.dtype != dtype or out.device != dev or not out.is_contiguous():
            raise ValueError("bad out tensor")
        check("bma_splice_rows", lib.bma_splice_rows(
            arr, len(segments), emb_weight.data_ptr() if emb_weight is not None else None, V,
            ids.data_ptr() if ids is not None else None, B, n_opt, D, _DT[dtype], float(emb_scale), rows.data_ptr(), N,
            out.data_ptr(), _stream(dev)))
        return out
    if out is None:
        out = torch.empty((B, S, D), dtype=dtype, device=dev)
    elif out.shape != (B, S, D) or out.dtype != dtype or out.device != dev or not out.is_contiguous():
        raise ValueError("bad out tensor")
    check("bma_splice", lib.bma_splice(arr, len(segments), emb_weight.data_ptr() if emb_weight is not None else None,
                                       V, ids.data_ptr() if ids is not None else None, B, n_opt, D, _DT[dtype],
                                       float(emb_scale), out.data_ptr(), _stream(dev)))
    return out


# ---------------------------------------------------------------------------
# fused elementwise tail of a Llama-family layer (candidate scoring only, no autograd)
def rmsnorm(x: torch.Tensor, weight: torch.Tensor, eps: float, gemma_style: bool = False) -> torch.Tensor:
    dev = _need_gpu(x, weight)
    D = x.shape[-1]
    if weight.shape != (D,) or weight.dtype != x.dtype or not weight.is_contiguous():
        raise ValueError("weight must be a contiguous (D,) tensor of x's dtype")
    x = x.contiguous()
    out = torch.empty_like(x)
    check("bma_rmsnorm", lib.bma_rmsnorm(x.data_ptr(), weight.data_ptr(), float(eps), x.numel() // D, D, _dt(x),
                                         1 if gemma_style else 0, out.data_ptr(), _stream(dev)))
    return out


def add_rmsnorm(residual: torch.Tensor, h: torch.Tensor, weight: torch.Tensor, eps: float, gemma_style: bool = False,
                pre_weight: Optional[torch.Tensor] = None, pre_eps: float = 0.0):
    """(sum, y): sum = dt(residual + a), y = rmsnorm(sum; weight) with a = h or, given ``pre_weight``, the RMSNorm of h
    under it (Gemma-3's norm on the branch output) -- one pass (include/bma.h: bma_add_rmsnorm)."""
    dev = _need_gpu(residual, h, weight)
    D = h.shape[-1]
    if residual.shape != h.shape or residual.dtype != h.dtype or weight.shape != (D,) or weight.dtype != h.dtype \
            or not weight.is_contiguous():
        raise ValueError("residual and h must agree in shape and dtype; weight a contiguous (D,) tensor of that dtype")
    if pre_weight is not None and (pre_weight.shape != (D,) or pre_weight.dtype != h.dtype or not pre_weight.is_contiguous()):
        raise ValueError("pre_weight must be a contiguous (D,) tensor of h's dtype")
    residual, h = residual.contiguous(), h.contiguous()
    s_out, y = torch.empty_like(h), torch.empty_like(h)
    check("bma_add_rmsnorm", lib.bma_add_rmsnorm(
        residual.data_ptr(), h.data_ptr(), pre_weight.data_ptr() if pre_weight is not None else None, float(pre_eps),
        weight.data_ptr(), float(eps), h.numel() // D, D, _dt(h), 1 if gemma_style else 0, s_out.data_ptr(), y.data_ptr(),
        _stream(dev)))
    return s_out, y


def add_rmsnorm_ok(x: torch.Tensor, weight: torch.Tensor) -> bool:
    """Shapes bma_add_rmsnorm / bma_rmsnorm take: 16-byte rows of at most 16 KiB in one of the three dtypes."""
    rb = x.shape[-1] * x.element_size()
    return bool(x.is_cuda and x.dtype in _DT and weight.dtype == x.dtype and rb % 16 == 0 and rb <= 16384)


class AddRMSNormFn(torch.autograd.Function):
    """(residual, h) -> (residual + h, rmsnorm(residual + h)) under autograd (the gradient pass): the backward folds
    the gradient arriving at the sum through the residual stream into the norm's backward launch."""

    @staticmethod
    def forward(ctx, residual, h, weight, eps, gemma_style):
        s_out, y = add_rmsnorm(residual, h, weight, eps, gemma_style)
        ctx.save_for_backward(s_out, weight)
        ctx.eps, ctx.gemma = float(eps), bool(gemma_style)
        return s_out, y

    @staticmethod
    def backward(ctx, d_sum, d_y):
        x, w = ctx.saved_tensors
        if d_y is None:
            return d_sum, d_sum, None, None, None
        d_y = d_y.contiguous()
        d_sum = None if d_sum is None else d_sum.contiguous()
        dx = torch.empty_like(x)
        D = x.shape[-1]
        check("bma_add_rmsnorm_bwd", lib.bma_add_rmsnorm_bwd(
            x.data_ptr(), w.data_ptr(), d_y.data_ptr(), d_sum.data_ptr() if d_sum is not None else None, ctx.eps,
            x.numel() // D, D, _dt(x), 1 if ctx.gemma else 0, dx.data_ptr(), _stream(x.device)))
        return dx, dx, None, None, None


# ---------------------------------------------------------------------------
# LayerNorm (+ the residual add in front of it) of a pre-LN vision-tower block, one launch each way (csrc/fused_elementwise.hip)
def layernorm_ok(x: torch.Tensor, weight, bias) -> bool:
    """Rows bma_add_layernorm takes: a contiguous GPU tensor in one of the three dtypes whose rows are 16-byte multiples of at
    most 16 KiB, with contiguous (D,) weight and bias of its dtype."""
    if not (torch.is_tensor(weight) and torch.is_tensor(bias)):
        return False
    D = x.shape[-1]
    rb = D * x.element_size()
    return bool(x.is_cuda and x.dtype in _DT and x.is_contiguous() and rb % 16 == 0 and rb <= 16384
                and weight.shape == (D,) and bias.shape == (D,) and weight.dtype == x.dtype and bias.dtype == x.dtype
                and weight.is_contiguous() and bias.is_contiguous())


def add_layernorm(residual: Optional[torch.Tensor], h: torch.Tensor, weight: torch.Tensor, bias: torch.Tensor, eps: float,
                  want_stats: bool = False):
    """(sum, y, stats): sum = dt(residual + h) (None without a residual), y = LayerNorm(sum or h; weight, bias, eps), stats (rows, 2)
    fp32 = (mean, rstd) per row when asked for (the backward's input).  One launch (include/bma.h: bma_add_layernorm)."""
    dev = _need_gpu(h, weight, bias)
    if not layernorm_ok(h, weight, bias) or (residual is not None and (residual.shape != h.shape or residual.dtype != h.dtype
                                                                         or not residual.is_contiguous() or residual.device != dev)):
        raise ValueError("add_layernorm wants contiguous residual / h of one shape and dtype, rows of 16-byte multiples <= 16 KiB")
    D = h.shape[-1]
    rows = h.numel() // D
    s_out = torch.empty_like(h) if residual is not None else None
    y = torch.empty_like(h)
    stats = torch.empty((rows, 2), dtype=torch.float32, device=dev) if want_stats else None
    check("bma_add_layernorm", lib.bma_add_layernorm(
        residual.data_ptr() if residual is not None else None, h.data_ptr(), weight.data_ptr(), bias.data_ptr(), float(eps), rows, D,
        _dt(h), s_out.data_ptr() if s_out is not None else None, y.data_ptr(), stats.data_ptr() if stats is not None else None,
        _stream(dev)))
    return s_out, y, stats


def _layernorm_bwd(x, w, stats, d_y, d_sum):
    d_y = d_y.contiguous()
    d_sum = None if d_sum is None else d_sum.contiguous()
    dx = torch.empty_like(x)
    D = x.shape[-1]
    check("bma_add_layernorm_bwd", lib.bma_add_layernorm_bwd(
        x.data_ptr(), w.data_ptr(), d_y.data_ptr(), d_sum.data_ptr() if d_sum is not None else None, stats.data_ptr(),
        x.numel() // D, D, _dt(x), dx.data_ptr(), _stream(x.device)))
    return dx


class AddLayerNormFn(torch.autograd.Function):
    """(residual, h) -> (residual + h, LayerNorm(residual + h)) under autograd, with the gradient arriving at the sum through the
    residual stream folded into the norm's backward launch.  Weight and bias are constants of the attack (the engine only ever
    asks for gradients w.r.t. inputs)."""

    @staticmethod
    def forward(ctx, residual, h, weight, bias, eps):
        s_out, y, stats = add_layernorm(residual.contiguous(), h.contiguous(), weight, bias, eps, want_stats=True)
        ctx.save_for_backward(s_out, weight, stats)
        return s_out, y

    @staticmethod
    def backward(ctx, d_sum, d_y):
        x, w, stats = ctx.saved_tensors
        if d_y is None:
            return d_sum, d_sum, None, None, None
        dx = _layernorm_bwd(x, w, stats, d_y, d_sum)
        return dx, dx, None, None, None


class LayerNormFn(torch.autograd.Function):
    """LayerNorm(h) under autograd through the same kernels (no residual in front: the first block of the tower)."""

    @staticmethod
    def forward(ctx, h, weight, bias, eps):
        hc = h.contiguous()
        _, y, stats = add_layernorm(None, hc, weight, bias, eps, want_stats=True)
        ctx.save_for_backward(hc, weight, stats)
        return y

    @staticmethod
    def backward(ctx, d_y):
        x, w, stats = ctx.saved_tensors
        return _layernorm_bwd(x, w, stats, d_y, None), None, None, None


ACT_SILU, ACT_GELU_TANH = 0, 1


def swiglu(gate: torch.Tensor, up: torch.Tensor, act: int = ACT_SILU) -> torch.Tensor:
    """dt(dt(act(gate)) * up): the gate of a Llama (SiLU) or Gemma (GELU-tanh) MLP."""
    dev = _need_gpu(gate, up)
    if gate.shape != up.shape or gate.dtype != up.dtype:
        raise ValueError("gate and up must agree in shape and dtype")
    gate, up = gate.contiguous(), up.contiguous()
    out = torch.empty_like(gate)
    check("bma_gated_act", lib.bma_gated_act(gate.data_ptr(), up.data_ptr(), gate.numel(), _dt(gate), int(act),
                                             out.data_ptr(), _stream(dev)))
    return out


def interleave_gate_up(gate_w: torch.Tensor, up_w: torch.Tensor) -> torch.Tensor:
    """(2I, D) weight whose product with x gives gate and up as alternating 16-byte chunks: rows
    [16j, 16j+8) are gate rows [8j, 8j+8), rows [16j+8, 16j+16) the matching up rows (8 = 16 B of a 16-bit
    dtype, 4 for fp32)."""
    I, D = gate_w.shape
    ne = 16 // gate_w.element_size()
    if up_w.shape != (I, D) or I % ne:
        raise ValueError("gate/up weights must agree in shape, with a row count that is a multiple of one 16-byte chunk")
    return torch.stack([gate_w.reshape(I // ne, ne, D), up_w.reshape(I // ne, ne, D)], dim=1).reshape(2 * I, D).contiguous()


def swiglu_il(gate_up: torch.Tensor, act: int = ACT_SILU) -> torch.Tensor:
    """dt(dt(act(gate)) * up) for gate/up interleaved in 16-byte chunks along the last dimension (2I -> I)."""
    dev = _need_gpu(gate_up)
    gate_up = gate_up.contiguous()
    two_i = gate_up.shape[-1]
    if two_i % (2 * (16 // gate_up.element_size())):
        raise ValueError("last dimension must hold whole pairs of 16-byte chunks")
    out = torch.empty(gate_up.shape[:-1] + (two_i // 2,), dtype=gate_up.dtype, device=dev)
    check("bma_gated_act_il", lib.bma_gated_act_il(gate_up.data_ptr(), out.numel(), _dt(gate_up), int(act), out.data_ptr(),
                                                   _stream(dev)))
    return out


class SwiGLUInterleavedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate_up, act=ACT_SILU):
        gu = gate_up.contiguous()
        ctx.save_for_backward(gu)
        ctx.act = int(act)
        return swiglu_il(gu, act)

    @staticmethod
    def backward(ctx, dy):
        (gu,) = ctx.saved_tensors
        dy = dy.contiguous()
        dgu = torch.empty_like(gu)
        check("bma_gated_act_il_bwd", lib.bma_gated_act_il_bwd(gu.data_ptr(), dy.data_ptr(), dy.numel(), _dt(gu), ctx.act,
                                                               dgu.data_ptr(), _stream(gu.device)))
        return dgu, None


def rope_(q: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> torch.Tensor:
    """In-place rotary embedding of q (B,H,L,Dh; any strides with a contiguous last dim);
    cos/sin (1|B, L, Dh)."""
    dev = _need_gpu(q, cos, sin)
    if q.dim() != 4 or q.stride(3) != 1 or cos.shape != sin.shape or cos.dim() != 3:
        raise ValueError("q must be (B,H,L,Dh) with a contiguous last dim; cos/sin (1|B,L,Dh)")
    B, H, L, Dh = q.shape
    if cos.shape[1] != L or cos.shape[2] != Dh or cos.shape[0] not in (1, B) or cos.dtype != q.dtype:
        raise ValueError("cos/sin do not match q")
    cos, sin = cos.contiguous(), sin.contiguous()
    check("bma_rope_inplace", lib.bma_rope_inplace(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), B, H, L, Dh,
                                                   cos.data_ptr(), sin.data_ptr(), cos.shape[0], _dt(q), _stream(dev)))
    return q


def rope(q: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, inverse: bool = False) -> torch.Tensor:
    """Out-of-place rotary embedding (``inverse``: the rotation by -angle, i.e. the backward)."""
    dev = _need_gpu(q, cos, sin)
    if q.dim() != 4 or q.stride(3) != 1 or cos.shape != sin.shape or cos.dim() != 3:
        raise ValueError("q must be (B,H,L,Dh) with a contiguous last dim; cos/sin (1|B,L,Dh)")
    B, H, L, Dh = q.shape
    if cos.shape[1] != L or cos.shape[2] != Dh or cos.shape[0] not in (1, B) or cos.dtype != q.dtype:
        raise ValueError("cos/sin do not match q")
    cos, sin = cos.contiguous(), sin.contiguous()
    out = torch.empty((B, L, H, Dh), dtype=q.dtype, device=q.device).transpose(1, 2)     # the projection's memory order
    check("bma_rope", lib.bma_rope(q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), out.data_ptr(), out.stride(0),
                                   out.stride(1), out.stride(2), B, H, L, Dh, cos.data_ptr(), sin.data_ptr(), cos.shape[0],
                                   -1.0 if inverse else 1.0, _dt(q), _stream(dev)))
    return out


def rope2(q: torch.Tensor, k: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, inverse: bool = False,
          inplace: bool = False):
    """Rotary embedding of q (B,H,L,Dh) and k (B,Hk,L,Dh) in ONE launch (bma_rope2); out of place by default (outputs in
    the projections' (B,L,heads,Dh) memory order), ``inplace`` rotates the given tensors."""
    dev = _need_gpu(q, k, cos, sin)
    if q.dim() != 4 or k.dim() != 4 or q.stride(3) != 1 or k.stride(3) != 1 or cos.shape != sin.shape or cos.dim() != 3:
        raise ValueError("q/k must be (B,heads,L,Dh) with a contiguous last dim; cos/sin (1|B,L,Dh)")
    B, H, L, Dh = q.shape
    Hk = k.shape[1]
    if k.shape != (B, Hk, L, Dh) or k.dtype != q.dtype or cos.shape[1] != L or cos.shape[2] != Dh \
            or cos.shape[0] not in (1, B) or cos.dtype != q.dtype:
        raise ValueError("k / cos / sin do not match q")
    cos, sin = cos.contiguous(), sin.contiguous()
    if inplace:
        qo, ko = q, k
    else:
        qo = torch.empty((B, L, H, Dh), dtype=q.dtype, device=dev).transpose(1, 2)
        ko = torch.empty((B, L, Hk, Dh), dtype=q.dtype, device=dev).transpose(1, 2)
    check("bma_rope2", lib.bma_rope2(
        q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), qo.data_ptr(), qo.stride(0), qo.stride(1), qo.stride(2), H,
        k.data_ptr(), k.stride(0), k.stride(1), k.stride(2), ko.data_ptr(), ko.stride(0), ko.stride(1), ko.stride(2), Hk,
        B, L, Dh, cos.data_ptr(), sin.data_ptr(), cos.shape[0], -1.0 if inverse else 1.0, _dt(q), _stream(dev)))
    return qo, ko


def quick_gelu_ok(x: torch.Tensor) -> bool:
    # (not fp16: aten's half-precision sigmoid on this stack is up to 7 units in the last place off the correctly
    # rounded value -- x = -2.93 gives 0.0067968 for 0.0067713 -- and the kernel does not imitate that; such tensors keep
    # the eager chain)
    return (x.is_cuda and x.dtype in (torch.bfloat16, torch.float32) and x.is_contiguous()
            and (x.numel() * x.element_size()) % 16 == 0 and x.data_ptr() % 16 == 0)


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    """x * sigmoid(1.702 x) in one launch, bit-identical to HuggingFace's QuickGELUActivation on the same tensor."""
    dev = _need_gpu(x)
    if not x.is_contiguous():
        raise ValueError("quick_gelu needs a contiguous tensor")
    out = torch.empty_like(x)
    check("bma_quick_gelu", lib.bma_quick_gelu(x.data_ptr(), x.numel(), _dt(x), out.data_ptr(), _stream(dev)))
    return out


class QuickGELUFn(torch.autograd.Function):
    """QuickGELU with its whole autograd backward in one launch (the sigmoid is recomputed from the input)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return quick_gelu(x)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        check("bma_quick_gelu_bwd", lib.bma_quick_gelu_bwd(x.data_ptr(), dy.data_ptr(), x.numel(), _dt(x), dx.data_ptr(),
                                                             _stream(x.device)))
        return dx


def qknorm_rope_ok(x: torch.Tensor) -> bool:
    """Can bma_qknorm_rope2 take this (B,H,L,Dh) view?  A head's 16-byte chunks must be one aligned group of at most 64
    lanes (Dh * es / 16 a power of two), every stride a multiple of 16 bytes."""
    es = x.element_size()
    c = x.shape[-1] * es // 16
    return (x.is_cuda and x.dim() == 4 and x.stride(3) == 1 and (x.shape[-1] * es) % 32 == 0 and 0 < c <= 64 and (c & (c - 1)) == 0
            and all((st * es) % 16 == 0 for st in x.stride()[:3]) and x.data_ptr() % 16 == 0)


def qknorm_rope2(q: torch.Tensor, k: torch.Tensor, wq: torch.Tensor, wk: torch.Tensor, eps: float, gemma: bool,
                 cos: torch.Tensor, sin: torch.Tensor, inplace: bool = False):
    """Per-head RMSNorm of q (B,H,L,Dh) and k (B,Hk,L,Dh) -- weights wq / wk (Dh,) -- and their rotary embedding in ONE
    launch (bma_qknorm_rope2): bit for bit `rmsnorm` on the head rows followed by `rope2`.  No autograd form."""
    dev = _need_gpu(q, k, cos, sin, wq, wk)
    if q.dim() != 4 or k.dim() != 4 or q.stride(3) != 1 or k.stride(3) != 1 or cos.shape != sin.shape or cos.dim() != 3:
        raise ValueError("q/k must be (B,heads,L,Dh) with a contiguous last dim; cos/sin (1|B,L,Dh)")
    B, H, L, Dh = q.shape
    Hk = k.shape[1]
    if k.shape != (B, Hk, L, Dh) or k.dtype != q.dtype or cos.shape[1] != L or cos.shape[2] != Dh \
            or cos.shape[0] not in (1, B) or cos.dtype != q.dtype:
        raise ValueError("k / cos / sin do not match q")
    for w in (wq, wk):
        if w.shape != (Dh,) or w.dtype != q.dtype or not w.is_contiguous():
            raise ValueError("norm weights must be contiguous (Dh,) tensors of the q/k dtype")
    cos, sin = cos.contiguous(), sin.contiguous()
    if inplace:
        qo, ko = q, k
    else:
        qo = torch.empty((B, L, H, Dh), dtype=q.dtype, device=dev).transpose(1, 2)
        ko = torch.empty((B, L, Hk, Dh), dtype=q.dtype, device=dev).transpose(1, 2)
    check("bma_qknorm_rope2", lib.bma_qknorm_rope2(
        q.data_ptr(), q.stride(0), q.stride(1), q.stride(2), qo.data_ptr(), qo.stride(0), qo.stride(1), qo.stride(2), H,
        k.data_ptr(), k.stride(0), k.stride(1), k.stride(2), ko.data_ptr(), ko.stride(0), ko.stride(1), ko.stride(2), Hk,
        B, L, Dh, wq.data_ptr(), wk.data_ptr(), float(eps), 1 if gemma else 0, cos.data_ptr(), sin.data_ptr(), cos.shape[0],
        _dt(q), _stream(dev)))
    return qo, ko


class RoPE2Fn(torch.autograd.Function):
    """q and k rotated by one launch; the backward is the inverse rotation of (dq, dk), one launch again."""

    @staticmethod
    def forward(ctx, q, k, cos, sin):
        ctx.save_for_backward(cos, sin)
        return rope2(q, k, cos, sin)

    @staticmethod
    def backward(ctx, dq, dk):
        cos, sin = ctx.saved_tensors
        if dq is None or dk is None:            # one side unused: the one-tensor kernel
            return (None if dq is None else rope(dq, cos, sin, inverse=True),
                    None if dk is None else rope(dk, cos, sin, inverse=True), None, None)
        gq, gk = rope2(dq, dk, cos, sin, inverse=True)
        return gq, gk, None, None


def attn_merge(o1: torch.Tensor, o2: torch.Tensor, lse1: torch.Tensor, lse2: torch.Tensor) -> torch.Tensor:
    """Merge prefix-attention (o1, lse1) and self-attention (o2, lse2) partial results.
    o1, o2: (B,L,H,Dh) contiguous; lse1: (H, B*L) fp32; lse2: (B,H,L) fp32."""
    dev = _need_gpu(o1, o2, lse1, lse2)
    B, L, H, Dh = o2.shape
    if o1.shape != o2.shape or o1.dtype != o2.dtype or not o1.is_contiguous() or not o2.is_contiguous():
        raise ValueError("o1/o2 must be contiguous (B,L,H,Dh) tensors of one dtype")
    if lse1.dtype != torch.float32 or lse2.dtype != torch.float32 or lse1.numel() != B * L * H or \
            lse2.shape != (B, H, L) or not lse1.is_contiguous() or not lse2.is_contiguous():
        raise ValueError("lse1 must be (H, B*L) and lse2 (B,H,L), contiguous fp32")
    out = torch.empty_like(o2)
    check("bma_attn_merge", lib.bma_attn_merge(o1.data_ptr(), o2.data_ptr(), lse1.data_ptr(), lse2.data_ptr(), B, L, H, Dh,
                                               _dt(o2), out.data_ptr(), _stream(dev)))
    return out


def attn_merge_rows(o1: torch.Tensor, o2: torch.Tensor, lse1: torch.Tensor, lse2: torch.Tensor,
                    row_map: torch.Tensor) -> torch.Tensor:
    """Ragged form: o1 (N,H,Dh), lse1 (H,N); o2 (B2,L,H,Dh), lse2 (B2,H,L); row_map (N,) int32 with the
    padded row b*L+l each computed token pairs with.  Returns (N,H,Dh)."""
    dev = _need_gpu(o1, o2, lse1, lse2, row_map)
    N, H, Dh = o1.shape
    B2, L = o2.shape[0], o2.shape[1]
    if o2.shape != (B2, L, H, Dh) or o1.dtype != o2.dtype or not o1.is_contiguous() or not o2.is_contiguous():
        raise ValueError("o1 must be (N,H,Dh) and o2 (B2,L,H,Dh), contiguous, one dtype")
    if lse1.dtype != torch.float32 or lse2.dtype != torch.float32 or lse1.shape != (H, N) or \
            lse2.shape != (B2, H, L) or not lse1.is_contiguous() or not lse2.is_contiguous():
        raise ValueError("lse1 must be (H,N) and lse2 (B2,H,L), contiguous fp32")
    if row_map.dtype != torch.int32 or row_map.shape != (N,) or not row_map.is_contiguous():
        raise ValueError("row_map must be a contiguous int32 (N,) tensor")
    out = torch.empty_like(o1)
    check("bma_attn_merge_rows", lib.bma_attn_merge_rows(o1.data_ptr(), o2.data_ptr(), lse1.data_ptr(), lse2.data_ptr(),
                                                         row_map.data_ptr(), N, B2, L, H, Dh, _dt(o1), out.data_ptr(),
                                                         _stream(dev)))
    return out


def gather_rows(src: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[r] = src[idx[r]] over the first dimension of a contiguous src; idx int32 (R,)."""
    dev = _need_gpu(src, idx)
    if not src.is_contiguous() or src.dim() < 2:
        raise ValueError("src must be contiguous with at least two dimensions")
    if idx.dtype != torch.int32 or idx.dim() != 1 or not idx.is_contiguous():
        raise ValueError("idx must be a contiguous int32 vector")
    row_bytes = src[0].numel() * src.element_size()
    out = torch.empty((idx.shape[0],) + tuple(src.shape[1:]), dtype=src.dtype, device=src.device)
    check("bma_gather_rows", lib.bma_gather_rows(src.data_ptr(), idx.data_ptr(), idx.shape[0], src.shape[0], row_bytes,
                                                 out.data_ptr(), _stream(dev)))
    return out


RAGGED_ATTN_MAX_LEN = 4096


def ragged_attention_ok(query: torch.Tensor, key: torch.Tensor, max_len: int) -> bool:
    """Can bma_ragged_attention take these (.,H,.,Dh) / (.,Hk,.,Dh) tensors?"""
    return (query.is_cuda and query.dtype in (torch.bfloat16, torch.float16) and key.dtype == query.dtype
            and query.shape[-1] in (32, 64, 128, 256) and 0 < max_len <= RAGGED_ATTN_MAX_LEN
            and query.shape[1] % key.shape[1] == 0)


def ragged_attention(query: torch.Tensor, key: torch.Tensor, value: torch.Tensor, prefix_k: Optional[torch.Tensor],
                     prefix_v: Optional[torch.Tensor], start: torch.Tensor, first: torch.Tensor, length: torch.Tensor,
                     max_len: int, scale: float, o1: Optional[torch.Tensor] = None,
                     lse1: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Attention of a ragged scoring row list (include/bma.h: bma_ragged_attention).

    query (1,H,N,Dh), key/value (1,Hk,N,Dh): any row/head strides with a contiguous last dim;
    prefix_k/prefix_v (1,Hk,P,Dh) or None; start/first/length int32 (B2,); o1 (N,H,Dh) + lse1 (H,N)
    optionally carry a prefix partial to merge.  Returns (N,H,Dh)."""
    dev = _need_gpu(query, key, value, start, first, length)
    _, H, N, Dh = query.shape
    Hk = key.shape[1]
    for t in (query, key, value):
        if t.dim() != 4 or t.shape[0] != 1 or t.stride(3) != 1 or t.shape[2] != N or t.shape[3] != Dh:
            raise ValueError("query/key/value must be (1,heads,N,Dh) with a contiguous last dim")
    for t in (start, first, length):
        if t.dtype != torch.int32 or t.dim() != 1 or not t.is_contiguous() or t.shape != start.shape:
            raise ValueError("start/first/length must be contiguous int32 vectors of one size")
    P = 0
    pk_ptr = pv_ptr = 0
    pk_s = pv_s = (0, 0)
    if prefix_k is not None:
        if prefix_k.shape != prefix_v.shape or prefix_k.dim() != 4 or prefix_k.shape[0] != 1 or \
                prefix_k.shape[1] != Hk or prefix_k.shape[3] != Dh or prefix_k.stride(3) != 1 or prefix_v.stride(3) != 1 \
                or prefix_k.dtype != query.dtype:
            raise ValueError("prefix_k/prefix_v must be (1,Hk,P,Dh) of the query dtype with a contiguous last dim")
        P = prefix_k.shape[2]
        pk_ptr, pv_ptr = prefix_k.data_ptr(), prefix_v.data_ptr()
        pk_s, pv_s = (prefix_k.stride(2), prefix_k.stride(1)), (prefix_v.stride(2), prefix_v.stride(1))
    o1_ptr = l1_ptr = 0
    if o1 is not None:
        if o1.shape != (N, H, Dh) or not o1.is_contiguous() or o1.dtype != query.dtype or lse1 is None or \
                lse1.shape != (H, N) or lse1.dtype != torch.float32 or not lse1.is_contiguous():
            raise ValueError("o1 must be contiguous (N,H,Dh) and lse1 contiguous fp32 (H,N)")
        o1_ptr, l1_ptr = o1.data_ptr(), lse1.data_ptr()
    out = torch.empty((N, H, Dh), dtype=query.dtype, device=query.device)
    check("bma_ragged_attention", lib.bma_ragged_attention(
        query.data_ptr(), query.stride(2), query.stride(1), key.data_ptr(), key.stride(2), key.stride(1),
        value.data_ptr(), value.stride(2), value.stride(1), pk_ptr, pk_s[0], pk_s[1], pv_ptr, pv_s[0], pv_s[1], P,
        start.data_ptr(), first.data_ptr(), length.data_ptr(), start.shape[0], int(max_len), N, H, Hk, Dh, _dt(query),
        float(scale), o1_ptr, l1_ptr, out.data_ptr(), _stream(dev)))
    return out


# Measurement knobs of the two attention kernels behind bma_ragged_attention (tools/fuzz_long_attn.py, tools/pmc_kernel.py):
# the environment is read HERE, once, and handed to the library through its ABI -- the kernels read no environment.
if "BMA_RAGGED_LONG" in _os.environ or "BMA_RAGGED_LONG_MIN" in _os.environ:
    lib.bma_ragged_attention_set_long(int(_os.environ.get("BMA_RAGGED_LONG", "1") or 1),
                                      int(_os.environ.get("BMA_RAGGED_LONG_MIN", "0") or 0))


def prefix_attention_ok(query: torch.Tensor, key: torch.Tensor) -> bool:
    """Can bma_prefix_attention take these (.,H,N,Dh) queries / (1,Hk,P,Dh) prefix keys?"""
    return (query.is_cuda and query.dtype in (torch.bfloat16, torch.float16) and key.dtype == query.dtype
            and query.shape[-1] in (64, 128) and query.shape[1] % key.shape[1] == 0 and key.shape[2] >= 1)


def prefix_attention(query: torch.Tensor, prefix_k: torch.Tensor, prefix_v: torch.Tensor, scale: float):
    """All rows against the shared prefix, no mask (include/bma.h: bma_prefix_attention).  query (1,H,N,Dh),
    prefix_k/prefix_v (1,Hk,P,Dh): any row/head strides with a contiguous last dim.  Returns (o1 (N,H,Dh) in the
    query dtype, lse1 (H,N) fp32 natural log): the partial bma_ragged_attention merges."""
    dev = _need_gpu(query, prefix_k, prefix_v)
    if query.dim() != 4 or query.shape[0] != 1 or query.stride(3) != 1:
        raise ValueError("query must be (1,H,N,Dh) with a contiguous last dim")
    _, H, N, Dh = query.shape
    Hk, P = prefix_k.shape[1], prefix_k.shape[2]
    for t in (prefix_k, prefix_v):
        if t.shape != (1, Hk, P, Dh) or t.stride(3) != 1 or t.dtype != query.dtype:
            raise ValueError("prefix_k/prefix_v must be (1,Hk,P,Dh) of the query dtype with a contiguous last dim")
    out = torch.empty((N, H, Dh), dtype=query.dtype, device=dev)
    lse = torch.empty((H, N), dtype=torch.float32, device=dev)
    check("bma_prefix_attention", lib.bma_prefix_attention(
        query.data_ptr(), query.stride(2), query.stride(1), prefix_k.data_ptr(), prefix_k.stride(2), prefix_k.stride(1),
        prefix_v.data_ptr(), prefix_v.stride(2), prefix_v.stride(1), P, N, H, Hk, Dh, _dt(query), float(scale),
        out.data_ptr(), lse.data_ptr(), _stream(dev)))
    return out, lse


# ---------------------------------------------------------------------------
# The hand-written batch-1 kernels of the gradient pass, one by one (A/B measurements: BMA_SKINNY_GEMM=0 ...).  The engine
# has ONE option for the four (EngineOptions.own_b1_kernels); what it switches on is what this table allows.
_OFF = ("0", "false", "False")
OWN_KERNELS = {"skinny_gemm": _os.environ.get("BMA_SKINNY_GEMM", "1") not in _OFF,
               "mid_gemm": _os.environ.get("BMA_MID_GEMM", "1") not in _OFF,
               "causal_attention": _os.environ.get("BMA_CAUSAL_ATTENTION", "1") not in _OFF,
               "b1_attention": _os.environ.get("BMA_FUSE_B1_ATTENTION", "1") not in _OFF}

# rotary + causal attention of one short sequence, forward and backward (csrc/b1_attention.hip): the batch-1 gradient pass
B1_ATTENTION_MAX_TOKENS = 80


def b1_attention_ok(qkv: torch.Tensor, cos: torch.Tensor, heads: int, kv_heads: int, head_dim: int) -> bool:
    """Can bma_b1_attention take this fused q/k/v projection output (.., S, (H + 2 Hk) * Dh)?"""
    S = qkv.shape[-2] if qkv.dim() >= 2 else 0
    return bool(qkv.is_cuda and qkv.dtype in (torch.bfloat16, torch.float16) and head_dim == 128 and heads == kv_heads
                and qkv.numel() == S * 3 * heads * 128 and 1 <= S <= B1_ATTENTION_MAX_TOKENS and qkv.stride(-1) == 1
                and qkv.stride(-2) % 8 == 0 and qkv.data_ptr() % 16 == 0 and cos.dtype == qkv.dtype
                and cos.numel() == S * 128 and cos.shape[-1] == 128)


def b1_attention(qkv: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor, heads: int, scale: float):
    """(out (S, H*128), lse (H, S) fp32) of rotary + causal attention over qkv (S, 3*H*128) [q heads | k heads | v heads]
    (include/bma.h: bma_b1_attention).  cos / sin (S, 128)."""
    dev = _need_gpu(qkv, cos, sin)
    S = qkv.shape[0]
    cos, sin = cos.reshape(S, 128).contiguous(), sin.reshape(S, 128).contiguous()
    out = torch.empty((S, heads * 128), dtype=qkv.dtype, device=dev)
    lse = torch.empty((heads, S), dtype=torch.float32, device=dev)
    check("bma_b1_attention", lib.bma_b1_attention(qkv.data_ptr(), qkv.stride(0), cos.data_ptr(), sin.data_ptr(), S, heads,
                                                   _dt(qkv), float(scale), out.data_ptr(), out.stride(0), lse.data_ptr(),
                                                   _stream(dev)))
    return out, lse


def b1_attention_bwd(qkv, cos, sin, out, lse, dout, heads: int, scale: float) -> torch.Tensor:
    """d(qkv) (S, 3*H*128) from d(out) (S, H*128) (include/bma.h: bma_b1_attention_bwd)."""
    dev = _need_gpu(qkv, cos, sin, out, lse, dout)
    S = qkv.shape[0]
    cos, sin = cos.reshape(S, 128).contiguous(), sin.reshape(S, 128).contiguous()
    dout = dout.contiguous()
    dqkv = torch.empty((S, 3 * heads * 128), dtype=qkv.dtype, device=dev)
    check("bma_b1_attention_bwd", lib.bma_b1_attention_bwd(
        qkv.data_ptr(), qkv.stride(0), cos.data_ptr(), sin.data_ptr(), out.data_ptr(), out.stride(0), lse.data_ptr(),
        dout.data_ptr(), dout.stride(0), S, heads, _dt(qkv), float(scale), dqkv.data_ptr(), dqkv.stride(0), _stream(dev)))
    return dqkv


class B1AttentionFn(torch.autograd.Function):
    """out = attention(rope(q), rope(k), v) for ONE sequence, straight from the fused q/k/v projection output and into the
    layout o_proj reads; the backward hands d(qkv) back in the projection's layout -- two launches where the unfused
    route has ten (bma_b1_attention)."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, heads, scale):
        q2 = qkv.reshape(qkv.shape[-2], qkv.shape[-1])
        out, lse = b1_attention(q2, cos, sin, heads, scale)
        ctx.save_for_backward(q2, cos, sin, out, lse)
        ctx.heads, ctx.scale, ctx.shape = int(heads), float(scale), tuple(qkv.shape)
        return out.view(*qkv.shape[:-1], heads * 128)

    @staticmethod
    def backward(ctx, dout):
        q2, cos, sin, out, lse = ctx.saved_tensors
        d2 = dout.reshape(out.shape)
        return b1_attention_bwd(q2, cos, sin, out, lse, d2, ctx.heads, ctx.scale).view(ctx.shape), None, None, None, None


# ---------------------------------------------------------------------------
# causal attention of one long sequence at batch 1, forward and backward (csrc/causal_attention.hip)
CAUSAL_ATTENTION = True         # module switch (set by the engine: EngineOptions.own_b1_kernels and OWN_KERNELS["causal_attention"])
CAUSAL_ATTENTION_MAX_TOKENS = 4096


def _rows_heads(t: torch.Tensor):
    """(row stride, head stride) in elements of a (L, H, 128) view whose last dim is contiguous."""
    return t.stride(0), t.stride(1)


if _os.environ.get("BMA_CA_FWD_TQ2_MIN") is not None:        # experiment knob: the library itself reads no environment
    lib.bma_causal_attention_set_plan(int(_os.environ["BMA_CA_FWD_TQ2_MIN"]))
if _os.environ.get("BMA_PREFIX_ATTN_PLAN") is not None:      # 0 by shape / 1 the 16x16x32 kernel / 4, 8: the 32x32x16 kernel on that many waves
    lib.bma_prefix_attention_set_plan(int(_os.environ["BMA_PREFIX_ATTN_PLAN"]))


def causal_attention_ok(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor) -> bool:
    """q (Lq, H, Dh), k / v (Lk, Hkv, Dh) views with Dh 64, 72 (SigLIP; computed in 96-wide LDS images), 128 or 256 (Gemma-3's
    decoder), 16-bit, last dim contiguous, strides multiples of 8, H a multiple of Hkv (grouped queries: heads h*rep ..
    share key/value head h), Lq <= Lk (causal: the queries are the last Lq positions)."""
    if not (q.is_cuda and q.dtype in (torch.bfloat16, torch.float16) and k.dtype == q.dtype and v.dtype == q.dtype):
        return False
    if q.dim() != 3 or k.dim() != 3 or v.shape != k.shape or q.shape[2] != k.shape[2] or q.shape[2] not in (64, 72, 128, 256):
        return False
    if k.shape[1] < 1 or q.shape[1] % k.shape[1]:
        return False
    if not (0 < q.shape[0] <= k.shape[0] <= CAUSAL_ATTENTION_MAX_TOKENS):
        return False
    for t in (q, k, v):
        if t.stride(2) != 1 or t.stride(0) % 8 or t.stride(1) % 8 or t.data_ptr() % 16:
            return False
    return True


def causal_attention(q: torch.Tensor, k: torch.Tensor, v: torch.Tensor, scale: float, causal: bool = True):
    """(out (Lq, H, Dh) contiguous, lse2 (H, Lq) fp32) through bma_causal_attention_gqa (include/bma.h); ``causal=False``:
    every query sees every key (a vision tower)."""
    dev = _need_gpu(q, k, v)
    if not causal_attention_ok(q, k, v):
        raise ValueError("causal_attention wants 16-bit (L, H, 64 | 72 | 128 | 256) views with a contiguous last dim, Lq <= Lk, "
                         "H a multiple of the key/value heads")
    Lq, H, Dh = q.shape
    Lk, Hkv = k.shape[0], k.shape[1]
    out = torch.empty((Lq, H, Dh), dtype=q.dtype, device=dev)
    lse2 = torch.empty((H, Lq), dtype=torch.float32, device=dev)
    check("bma_causal_attention_gqa", lib.bma_causal_attention_gqa(
        q.data_ptr(), *_rows_heads(q), k.data_ptr(), *_rows_heads(k), v.data_ptr(), *_rows_heads(v), Lq, Lk, H, Hkv, Dh, _dt(q),
        1 if causal else 0, float(scale), out.data_ptr(), lse2.data_ptr(), _stream(dev)))
    return out, lse2


def causal_attention_bwd(q, k, v, out, lse2, d_out, scale: float, into=None, causal: bool = True):
    """(dq (Lq, H, Dh), dk, dv (Lk, Hkv, Dh)) through bma_causal_attention_bwd_gqa: fresh contiguous tensors, or -- `into` a
    (Lk, 3, H, Dh) buffer with Lq == Lk and Hkv == H -- views of it, the three written straight into the gradient of a fused
    q/k/v projection."""
    dev = _need_gpu(q, k, v)
    Lq, H, Dh = q.shape
    Lk, Hkv = k.shape[0], k.shape[1]
    d_out = d_out.contiguous()
    if into is None:
        dq = torch.empty((Lq, H, Dh), dtype=q.dtype, device=dev)
        dk = torch.empty((Lk, Hkv, Dh), dtype=q.dtype, device=dev)
        dv = torch.empty((Lk, Hkv, Dh), dtype=q.dtype, device=dev)
        dq_rs, dkv_rs = H * Dh, Hkv * Dh
    else:
        if Lq != Lk or Hkv != H or into.shape != (Lk, 3, H, Dh) or not into.is_contiguous() or into.dtype != q.dtype:
            raise ValueError("`into` must be a contiguous (L, 3, H, Dh) buffer of the operands' type with Lq == Lk and Hkv == H")
        dq, dk, dv = into[:, 0], into[:, 1], into[:, 2]
        dq_rs = dkv_rs = 3 * H * Dh
    delta = torch.empty((H, Lq), dtype=torch.float32, device=dev)
    check("bma_causal_attention_bwd_gqa", lib.bma_causal_attention_bwd_gqa(
        q.data_ptr(), *_rows_heads(q), k.data_ptr(), *_rows_heads(k), v.data_ptr(), *_rows_heads(v), out.data_ptr(), lse2.data_ptr(),
        d_out.data_ptr(), Lq, Lk, H, Hkv, Dh, _dt(q), 1 if causal else 0, float(scale), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
        dq_rs, dkv_rs, delta.data_ptr(), _stream(dev)))
    return dq, dk, dv


class CausalAttentionFn(torch.autograd.Function):
    """out (Lq, H, Dh) = attention of q (Lq, H, Dh) against k / v (Lk, H, Dh) -- causal with the queries as the last Lq
    positions, or (``causal=False``) every key visible; the backward is two launches of the same library (no atomics)."""

    @staticmethod
    def forward(ctx, q, k, v, scale, causal=True):
        out, lse2 = causal_attention(q, k, v, scale, causal)
        ctx.save_for_backward(q, k, v, out, lse2)
        ctx.scale, ctx.causal = float(scale), bool(causal)
        return out

    @staticmethod
    def backward(ctx, d_out):
        q, k, v, out, lse2 = ctx.saved_tensors
        dq, dk, dv = causal_attention_bwd(q, k, v, out, lse2, d_out, ctx.scale, causal=ctx.causal)
        return dq, dk, dv, None, None


class RotaryCausalAttentionFn(torch.autograd.Function):
    """What HuggingFace's attention block does between its (fused) q/k/v projection and o_proj, for ONE long sequence at
    batch 1: rotary embedding of q and k (one launch), causal attention (one launch), and backward: the attention's two
    launches writing dq / dk / dv straight into the projection's gradient, the inverse rotation in place (one launch) --
    no split, no concatenation, no transposed copies.  qkv (1, S, 3*H*128) as the fused projection leaves it; cos / sin
    (S, 128).  Returns (1, S, H*128)."""

    @staticmethod
    def forward(ctx, qkv, cos, sin, heads, scale):
        S = qkv.shape[1]
        x = qkv.view(1, S, 3, heads, 128)
        q4, k4 = x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2)            # (1, H, S, 128) views
        qo, ko = rope2(q4, k4, cos.unsqueeze(0), sin.unsqueeze(0))
        q3, k3, v3 = qo[0].transpose(0, 1), ko[0].transpose(0, 1), x[0, :, 2]       # (S, H, 128)
        out, lse2 = causal_attention(q3, k3, v3, scale)
        ctx.save_for_backward(q3, k3, qkv, out, lse2, cos, sin)
        ctx.heads, ctx.scale = int(heads), float(scale)
        return out.view(1, S, heads * 128)

    @staticmethod
    def backward(ctx, d_out):
        q3, k3, qkv, out, lse2, cos, sin = ctx.saved_tensors
        S, H = qkv.shape[1], ctx.heads
        v3 = qkv.view(S, 3, H, 128)[:, 2]
        g = torch.empty((S, 3, H, 128), dtype=qkv.dtype, device=qkv.device)
        dq, dk, _ = causal_attention_bwd(q3, k3, v3, out, lse2, d_out.reshape(S, H, 128), ctx.scale, into=g)
        rope2(dq.transpose(0, 1).unsqueeze(0), dk.transpose(0, 1).unsqueeze(0), cos.unsqueeze(0), sin.unsqueeze(0), inverse=True,
              inplace=True)
        return g.view(1, S, 3 * H * 128), None, None, None, None


def rotary_causal_attention_ok(qkv: torch.Tensor, cos: torch.Tensor, heads: int) -> bool:
    """A contiguous 16-bit (1, S, 3*heads*128) projection with cos / sin (1, S, 128) of its type, S within the kernels'
    range and beyond the one-launch kernel's."""
    return bool(CAUSAL_ATTENTION and qkv.is_cuda and qkv.dim() == 3 and qkv.shape[0] == 1 and qkv.is_contiguous()
                and qkv.dtype in (torch.bfloat16, torch.float16) and qkv.shape[2] == 3 * heads * 128
                and B1_ATTENTION_MAX_TOKENS < qkv.shape[1] <= CAUSAL_ATTENTION_MAX_TOKENS
                and cos.dtype == qkv.dtype and cos.shape == (1, qkv.shape[1], 128))


# ---------------------------------------------------------------------------
# skinny products of the batch-1 gradient pass on the hand-written kernel (csrc/gemm_nt.hip)
SKINNY_GEMM = True              # module switch (set by the engine: EngineOptions.own_b1_kernels and OWN_KERNELS["skinny_gemm"])
GEMM_NT_MAX_ROWS = int(_os.environ.get("BMA_GEMM_NT_MAX_ROWS", "96"))   # one 64- or 96-row tile: the shapes the kernel is built and measured for
# Routed where the kernel measures faster than the tuned library (tools/gemm_bench.py, profiles/r4_gemm_bench.txt): every
# product of the pass whose long side is at least 2.5x its short one -- long reductions (the library has to split K
# itself: the input gradients through the transposed copies, down_proj: 1.3-1.5x) and wide outputs (gate/up, q/k/v, the
# input gradient of down_proj: 1.0-1.13x); the square o_proj ties (0.98x) and stays with the library.
# GEMM_NT_MIN_K_OVER_N = 0 routes EVERY shape (tools, tests); GEMM_NT_MIN_N_OVER_K = 0 switches the wide-output rule OFF
# (only K >= MIN_K_OVER_N * N is routed then).
GEMM_NT_MIN_K_OVER_N = float(_os.environ.get("BMA_GEMM_NT_MIN_K_OVER_N", "2.5"))
GEMM_NT_MIN_N_OVER_K = float(_os.environ.get("BMA_GEMM_NT_MIN_N_OVER_K", "2.5"))
_GEMM_WS_BYTES = 64 << 20
_GEMM_COUNTERS = 4096
_GEMM_WS = {}
_GEMM_WS_MAX_EAGER = 4
GEMM_NT_HOOK = None             # measurement: called as hook(x, w) for every product routed to the kernel (bench.py)


def gemm_workspace(dev: torch.device):
    """(partial-sum workspace, zeroed tile tickets) of the split-K reduction.  Launches that share a pair must be ordered
    on one stream (the last arriver of a tile re-zeroes its ticket for the NEXT launch), so there is one pair per
    (device, stream) for eager work -- two attack objects or a side stream on one device get their own -- and one per
    device for everything captured into hipGraphs, whose replays the engine issues from one stream; captured graphs hold
    the addresses, so pairs are never freed or reallocated.  None while a capture is running and its pair does not exist
    yet (allocate it beforehand: ``FusedInference`` does at construction)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    capturing = torch.cuda.is_current_stream_capturing()
    key = (dev.type, idx, "graphs" if capturing else torch.cuda.current_stream(dev).cuda_stream)
    ws = _GEMM_WS.get(key)
    if ws is None:
        if capturing:
            return None
        # eager pairs are bounded (every graph warm-up on a fresh side stream left 64 MB behind, 2 GB over the
        # stream pool): the LEAST RECENTLY USED eager pair goes when a fifth stream asks (by insertion order the
        # main stream's pair -- the one every eager call uses -- went first and was reallocated again and again) -- its
        # memory returns to the caching allocator under the stream it was allocated on, i.e. behind that stream's own last
        # launch; the "graphs" pair (captured launches hold its address) is never dropped
        eager = [k for k in _GEMM_WS if k[:2] == key[:2] and k[2] != "graphs"]
        if len(eager) >= _GEMM_WS_MAX_EAGER:
            del _GEMM_WS[eager[0]]
        ws = (torch.empty(_GEMM_WS_BYTES, dtype=torch.uint8, device=dev), torch.zeros(_GEMM_COUNTERS, dtype=torch.int32, device=dev))
        _GEMM_WS[key] = ws
    elif not capturing:
        _GEMM_WS[key] = _GEMM_WS.pop(key)        # a hit moves the pair to the young end (dicts keep insertion order)
    return ws


_GRAPH_REPLAY_STREAM = {}      # device index -> (stream handle captured graphs are being replayed from, owner token)
_GRAPH_OWNER = [None]          # the attack whose run() started last (BimodalAttack.run sets it): default owner of a replay


def set_graph_owner(token) -> None:
    """Called by the engine when an attack's run starts: replays from here on belong to it (see note_graph_replay)."""
    _GRAPH_OWNER[0] = token


def note_graph_replay(dev: torch.device, owner=None) -> None:
    """Every captured graph of a device shares ONE split-K workspace and ticket array (``gemm_workspace``): their replays
    must be ordered on one stream, or two graphs would race on partial sums and tickets.  The engine replays from the
    stream current at the call; this makes a second stream an error instead of a silent race.  The pin belongs
    to an `owner` (the attack object whose graphs are being replayed): a LATER owner -- the next attack of the process, a
    test -- may legitimately replay its own graphs under another stream once the previous owner is done, and takes the
    pin over (the first replay stream used to be pinned for the whole process); two streams within one owner
    are the race."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    handle = torch.cuda.current_stream(dev).cuda_stream
    owner = _GRAPH_OWNER[0] if owner is None else owner
    first = _GRAPH_REPLAY_STREAM.get(idx)
    if first is None or (owner is not None and first[1] != owner):
        _GRAPH_REPLAY_STREAM[idx] = (handle, owner)
        return
    if first[0] != handle:
        raise RuntimeError("bimodalattack_amd: captured graphs of one device must be replayed from one stream (they share the "
                           f"split-K workspace); first replay came from stream {first[0]:#x}, this one from {handle:#x}")


def release_graph_replay(dev: torch.device, owner) -> None:
    """The owner's graphs are gone (its attack finished): the next replay on this device may come from any stream."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if _GRAPH_REPLAY_STREAM.get(idx, (None, None))[1] == owner:
        _GRAPH_REPLAY_STREAM.pop(idx, None)


def gemm_workspace_for_graphs(dev: torch.device):
    """Allocate the pair captured launches use (call outside any capture)."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    key = (dev.type, idx, "graphs")
    if key not in _GEMM_WS and not torch.cuda.is_current_stream_capturing():
        _GEMM_WS[key] = (torch.empty(_GEMM_WS_BYTES, dtype=torch.uint8, device=dev),
                         torch.zeros(_GEMM_COUNTERS, dtype=torch.int32, device=dev))
    return _GEMM_WS.get(key)


def gemm_nt_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    """Can bma_gemm_nt compute linear(x, w)?  16-bit, K a multiple of 64, at most GEMM_NT_MAX_ROWS rows, rows 16-byte
    aligned, workspace within the fixed budget."""
    if not (SKINNY_GEMM and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and w.dim() == 2
            and x.dim() >= 2 and x.shape[-1] == w.shape[1] and w.is_contiguous() and x.stride(-1) == 1):
        return False
    K, N = w.shape[1], w.shape[0]
    M = x.numel() // K if K else 0
    if K % 64 or N % 4 or not (0 < M <= GEMM_NT_MAX_ROWS) or not x.is_contiguous():
        return False
    routed = GEMM_NT_MIN_K_OVER_N <= 0.0 or K >= GEMM_NT_MIN_K_OVER_N * N or \
        (GEMM_NT_MIN_N_OVER_K > 0.0 and N >= GEMM_NT_MIN_N_OVER_K * K)
    if not routed:
        return False                 # the library is as fast or faster there (csrc/gemm_nt.hip, "Measured")
    return lib.bma_gemm_nt_ws_bytes(M, N, K) <= _GEMM_WS_BYTES and lib.bma_gemm_nt_tiles(M, N, K) <= _GEMM_COUNTERS


# Cross-product weight prefetch (bma_gemm_nt_next): weight (by data pointer) -> the weight of the NEXT product of the
# gradient pass's chain, registered by fused.FusedInference (forward: qkv -> gate/up -> down -> the next layer's qkv; backward:
# the transposed copies in reverse).  Weights do not depend on activations, so the workgroups a split launch lets go early
# load the first stages of the next launch's weight rows.  A hint only; empty = plain bma_gemm_nt.
# MEASURED AND SWITCHED OFF (round 5, VERDICT r4 item 3's own rule: keep only above 3 %): the pass's chain of 192 launches
# 6.435 -> 6.347 ms at 65 rows (x1.014), 5.692 -> 5.642 at 44 (x1.009); per product x0.97-1.02 (profiles/r5_gemm_chain.txt).
# A launch's fixed cost is not the first stages' HBM latency -- it is the split-K tail (the last arriver reads S x 48 KB
# at one CU's L2 rate) and, for the unsplit gate/up product, 172 workgroups x ~25 GB/s of LDS-DMA per CU.
GEMM_NT_PREFETCH = _os.environ.get("BMA_GEMM_NT_PREFETCH", "0") not in ("0", "false", "False")
_GEMM_NT_NEXT = {}


def gemm_nt_chain(weights) -> int:
    """Register `weights` (contiguous 16-bit (N, K) tensors, in launch order; None entries are skipped) as one chain: each
    one's successor is prefetched by its launch.  Successors are held weakly (a derived weight copy that is rebuilt drops
    out by itself); returns the number of links made."""
    import weakref
    ws = [w for w in weights if w is not None and w.dim() == 2 and w.is_contiguous() and w.shape[1] % 64 == 0
          and w.dtype in (torch.bfloat16, torch.float16)]
    for a, b in zip(ws[:-1], ws[1:]):
        _GEMM_NT_NEXT[a.data_ptr()] = weakref.ref(b)
    return max(0, len(ws) - 1)


def gemm_nt_chain_clear() -> None:
    _GEMM_NT_NEXT.clear()


def _next_weight(w: torch.Tensor):
    ref = _GEMM_NT_NEXT.get(w.data_ptr())
    nxt = None if ref is None else ref()
    if ref is not None and nxt is None:
        del _GEMM_NT_NEXT[w.data_ptr()]
    return nxt


def gemm_nt(x: torch.Tensor, w: torch.Tensor, next_w: Optional[torch.Tensor] = None) -> torch.Tensor:
    """linear(x, w) = x @ w^T through bma_gemm_nt (include/bma.h); x (..., K) contiguous, w (N, K) contiguous.  `next_w`
    (or the registered successor of `w`, ``gemm_nt_chain``): the weight of the next product, prefetched (bma_gemm_nt_next)."""
    dev = _need_gpu(x, w)
    if w.dim() != 2 or x.dim() < 1 or x.shape[-1] != w.shape[1] or x.dtype != w.dtype or x.dtype not in (torch.bfloat16, torch.float16) \
            or not x.is_contiguous() or not w.is_contiguous() or w.shape[1] % 64:
        raise ValueError("gemm_nt wants contiguous 16-bit x (..., K) and w (N, K) of one dtype with K a multiple of 64")
    K, N = w.shape[1], w.shape[0]
    M = x.numel() // K
    need = lib.bma_gemm_nt_ws_bytes(M, N, K)
    if need:                                     # (an unsplit product takes no workspace: none is allocated for it)
        pair = gemm_workspace(dev)
        if pair is None:
            raise RuntimeError("bma_gemm_nt workspace requested inside a graph capture before it was allocated")
        ws, cnt = pair
        if need > ws.numel() or lib.bma_gemm_nt_tiles(M, N, K) > cnt.numel():
            raise ValueError("product beyond the fixed split-K workspace")
        ws_ptr, ws_len, cnt_ptr, cnt_len = ws.data_ptr(), ws.numel(), cnt.data_ptr(), cnt.numel()
    else:
        ws_ptr = ws_len = cnt_ptr = cnt_len = 0
    if GEMM_NT_HOOK is not None:
        GEMM_NT_HOOK(x, w)
    y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=dev)
    if next_w is None and GEMM_NT_PREFETCH:
        next_w = _next_weight(w)
    if next_w is not None and next_w.dtype == w.dtype and next_w.device == w.device and next_w.dim() == 2 and next_w.is_contiguous() \
            and next_w.shape[1] % 64 == 0:
        check("bma_gemm_nt_next", lib.bma_gemm_nt_next(x.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), N, M, N, K, _dt(x), ws_ptr,
                                                       ws_len, cnt_ptr, cnt_len, next_w.data_ptr(), next_w.shape[1],
                                                       next_w.shape[0], next_w.shape[1], _stream(dev)))
        return y
    check("bma_gemm_nt", lib.bma_gemm_nt(x.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), N, M, N, K, _dt(x), ws_ptr,
                                         ws_len, cnt_ptr, cnt_len, _stream(dev)))
    return y


# the same products at 599-644 rows (the pass with the image in the prompt) on csrc/gemm_mid.hip
MID_GEMM = True                 # module switch (set by the engine: EngineOptions.own_b1_kernels and OWN_KERNELS["mid_gemm"])
GEMM_MID_MIN_ROWS = int(_os.environ.get("BMA_GEMM_MID_MIN_ROWS", "560"))    # three 224-row tiles, the third at least half full
GEMM_MID_MAX_ROWS = int(_os.environ.get("BMA_GEMM_MID_MAX_ROWS", "672"))
# Routed where the kernel measures faster than the tuned library at 599-644 rows (tools/gemm_bench.py --mid,
# profiles/r4_gemm_mid_bench.txt): long reductions (N = 4096 with K = 11008 / 12288 / 22016, where the library has to split
# K itself: 1.25-1.65x) and the widest output (gate/up, N = 22016: 1.04-1.2x); q/k/v and the input gradient of down_proj
# tie and the square o_proj loses (0.8x): those stay with the library.  MIN_K_OVER_N = 0 routes every shape; MIN_N_OVER_K = 0
# switches the wide-output rule off.
GEMM_MID_MIN_K_OVER_N = float(_os.environ.get("BMA_GEMM_MID_MIN_K_OVER_N", "2.5"))
GEMM_MID_MIN_N_OVER_K = float(_os.environ.get("BMA_GEMM_MID_MIN_N_OVER_K", "4.0"))
GEMM_MID_HOOK = None            # measurement: called as hook(x, w) for every product routed to the kernel (bench.py)


def gemm_mid_ok(x: torch.Tensor, w: torch.Tensor) -> bool:
    """Can (and should) bma_gemm_mid compute linear(x, w)?  16-bit, K a multiple of 64, GEMM_MID_MIN_ROWS..MAX_ROWS
    rows, 16-byte aligned rows, partial sums within the fixed workspace."""
    if not (MID_GEMM and x.is_cuda and x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and w.dim() == 2
            and x.dim() >= 2 and x.shape[-1] == w.shape[1] and w.is_contiguous() and x.is_contiguous()):
        return False
    K, N = w.shape[1], w.shape[0]
    M = x.numel() // K if K else 0
    if K % 64 or N % 4 or not (GEMM_MID_MIN_ROWS <= M <= GEMM_MID_MAX_ROWS):
        return False
    routed = GEMM_MID_MIN_K_OVER_N <= 0.0 or K >= GEMM_MID_MIN_K_OVER_N * N or \
        (GEMM_MID_MIN_N_OVER_K > 0.0 and N >= GEMM_MID_MIN_N_OVER_K * K)
    if not routed:
        return False
    return lib.bma_gemm_mid_ws_bytes(M, N, K) <= _GEMM_WS_BYTES


def gemm_mid(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """linear(x, w) = x @ w^T through bma_gemm_mid (include/bma.h); x (..., K) contiguous, w (N, K) contiguous."""
    dev = _need_gpu(x, w)
    if w.dim() != 2 or x.dim() < 1 or x.shape[-1] != w.shape[1] or x.dtype != w.dtype or x.dtype not in (torch.bfloat16, torch.float16) \
            or not x.is_contiguous() or not w.is_contiguous() or w.shape[1] % 64:
        raise ValueError("gemm_mid wants contiguous 16-bit x (..., K) and w (N, K) of one dtype with K a multiple of 64")
    K, N = w.shape[1], w.shape[0]
    M = x.numel() // K
    need = lib.bma_gemm_mid_ws_bytes(M, N, K)
    ws_ptr, ws_len = 0, 0
    if need:
        pair = gemm_workspace(dev)
        if pair is None:
            raise RuntimeError("bma_gemm_mid workspace requested inside a graph capture before it was allocated")
        if need > pair[0].numel():
            raise ValueError("product beyond the fixed split-K workspace")
        ws_ptr, ws_len = pair[0].data_ptr(), pair[0].numel()
    if GEMM_MID_HOOK is not None:
        GEMM_MID_HOOK(x, w)
    y = torch.empty(x.shape[:-1] + (N,), dtype=x.dtype, device=dev)
    check("bma_gemm_mid", lib.bma_gemm_mid(x.data_ptr(), K, w.data_ptr(), K, y.data_ptr(), N, M, N, K, _dt(x), ws_ptr, ws_len,
                                           _stream(dev)))
    return y


def linear_b1(x: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """x @ w^T for a bias-free weight: the hand-written kernels where they apply, the library otherwise."""
    if gemm_nt_ok(x, w):
        K, N = w.shape[1], w.shape[0]
        if lib.bma_gemm_nt_ws_bytes(x.numel() // K, N, K) == 0 or gemm_workspace(x.device) is not None:
            return gemm_nt(x, w)
    if gemm_mid_ok(x, w):
        K, N = w.shape[1], w.shape[0]
        if lib.bma_gemm_mid_ws_bytes(x.numel() // K, N, K) == 0 or gemm_workspace(x.device) is not None:
            return gemm_mid(x, w)
    return torch.nn.functional.linear(x, w)


class FrozenLinearFn(torch.autograd.Function):
    """y = x W^T for a weight that is a constant of the attack; the input gradient dX = dY W is
    computed as ``linear(dY, W^T-copy)``.  Both products then run in the library's "weight rows along
    the reduction" form, which for the ~70-row gradient pass is the faster one on MI355X (tuned
    hipBLASLt, M=66: 25/40/37 us against 40/54/59 us for the q-k-v-o / gate-up / down shapes)."""

    @staticmethod
    def forward(ctx, x, weight, weight_t):
        ctx.weight_t = weight_t
        return linear_b1(x, weight)

    @staticmethod
    def backward(ctx, dy):
        return linear_b1(dy.contiguous(), ctx.weight_t), None, None


# ---------------------------------------------------------------------------
# the same ops under autograd (the gradient pass): fused forward + fused backward
class RMSNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, eps, gemma_style):
        xc = x.contiguous()
        ctx.save_for_backward(xc, weight)
        ctx.eps, ctx.gemma = float(eps), bool(gemma_style)
        return rmsnorm(xc, weight, eps, gemma_style)

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        # the engine borrows the model's weights read-only and only ever asks autograd for
        # gradients w.r.t. inputs: no weight gradient is produced, whatever requires_grad says
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        D = x.shape[-1]
        check("bma_rmsnorm_bwd", lib.bma_rmsnorm_bwd(x.data_ptr(), w.data_ptr(), dy.data_ptr(), ctx.eps, x.numel() // D, D,
                                                     _dt(x), 1 if ctx.gemma else 0, dx.data_ptr(), _stream(x.device)))
        return dx, None, None, None


class SwiGLUFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gate, up, act=ACT_SILU):
        g, u = gate.contiguous(), up.contiguous()
        ctx.save_for_backward(g, u)
        ctx.act = int(act)
        return swiglu(g, u, act)

    @staticmethod
    def backward(ctx, dy):
        g, u = ctx.saved_tensors
        dy = dy.contiguous()
        dg, du = torch.empty_like(g), torch.empty_like(u)
        check("bma_gated_act_bwd", lib.bma_gated_act_bwd(g.data_ptr(), u.data_ptr(), dy.data_ptr(), g.numel(), _dt(g),
                                                         ctx.act, dg.data_ptr(), du.data_ptr(), _stream(g.device)))
        return dg, du, None


class RoPEFn(torch.autograd.Function):
    """Out of place under autograd (one launch, no clone); the rotation is orthogonal, so the
    backward is the same kernel with the sign of sin flipped (no negated copy of sin either)."""

    @staticmethod
    def forward(ctx, q, cos, sin):
        ctx.save_for_backward(cos, sin)
        return rope(q, cos, sin)

    @staticmethod
    def backward(ctx, dq):
        cos, sin = ctx.saved_tensors
        return rope(dq, cos, sin, inverse=True), None, None
