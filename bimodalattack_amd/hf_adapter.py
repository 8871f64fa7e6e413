"""The HuggingFace plugin points the engine calls, and the two
mathematically-identical restructurings of the candidate forward:

* logits on the T target rows only (``logits_to_keep``) with the last input token
  dropped -- the reference materialises (B,S,V) logits and reads T rows of them
  (bimodal_attack.py:1287-1290);
* the keys/values of everything in front of the suffix computed ONCE per step and
  shared by all candidates -- under causal attention those positions cannot see the
  suffix, so their keys/values are the same in every candidate.

No kernel is launched from here and nothing depends on the device, so this file is
exercised by CPU tests as well.
"""

from __future__ import annotations

import copy
import contextlib
import inspect
from typing import Optional

import torch


def features_tensor(out) -> torch.Tensor:
    """What ``get_image_features`` hands back, as the (B,N,D) tensor the reference indexes (:528-536, :877-884, :972-979):
    transformers 4.50.2 (the reference's pin, requirements.txt:4) returns that tensor itself; 4.5x LLaVA releases a
    per-image list / tuple of (N,D) tensors; 5.x an output object whose ``pooler_output`` is the tensor (Gemma-3) or
    such a per-image list (LLaVA) -- and, called with ``return_dict=False``, that object's tuple
    ``(last_hidden_state, pooler_output, ...)``."""
    if torch.is_tensor(out):
        return out
    p = getattr(out, "pooler_output", None)
    if p is None and isinstance(out, (list, tuple)) and len(out) > 0:
        if all(torch.is_tensor(t) and t.dim() == 2 for t in out):
            p = out                                  # one (N,D) block per image
        elif len(out) >= 2 and (torch.is_tensor(out[1]) or isinstance(out[1], (list, tuple))):
            p = out[1]                               # the output object as a tuple: pooler_output comes second
    if p is None:
        raise TypeError(f"cannot find image features in {type(out).__name__}")
    return p if torch.is_tensor(p) else torch.stack(list(p))


class CapturableNormalize:
    """The caller's ``normalize`` when it is a per-channel (x - mean) / std -- torchvision's
    ``Normalize`` or anything else exposing ``mean`` and ``std`` -- with the two constants kept
    on the device.  torchvision rebuilds them from Python lists on every call: a host-to-device
    copy per call, which cannot be captured into a hipGraph.  Same two ops in the same order;
    checked once, bit for bit, against the caller's function, which is used instead on any
    difference."""

    def __init__(self, fn):
        self.fn = fn
        self.ok = None if (hasattr(fn, "mean") and hasattr(fn, "std")) else False
        self._consts = {}

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        if self.ok is False or x.dim() != 4:
            return self.fn(x)
        key = (x.device, x.dtype)
        c = self._consts.get(key)
        if c is None:
            try:
                c = tuple(torch.as_tensor(v, dtype=torch.float32).reshape(1, -1, 1, 1).to(x.device, x.dtype)
                          for v in (self.fn.mean, self.fn.std))
            except Exception:
                self.ok = False
                return self.fn(x)
            self._consts[key] = c
        out = (x - c[0]) / c[1]
        if self.ok is None:
            ref = self.fn(x)
            self.ok = bool(ref.shape == out.shape and torch.equal(ref.detach(), out.detach()))
            if not self.ok:
                return ref
        return out


import os as _os

_OFF = ("0", "false", "False")
# A/B switches of the vision-tower fast paths (engine options until round 4; an HFAdapter instance carries its own copies,
# which tests flip directly):
# towers whose head width is not a multiple of 32 (SigLIP: 72): the hand-written pair at the real width, else zero-padded
# q/k/v for the library (prefix_attention.padded_heads_attention)
PAD_VISION_HEADS = _os.environ.get("BMA_PAD_VISION_HEADS", "1") not in _OFF
# CLIP's QuickGELU (x * sigmoid(1.702 x): three launches forward, five backward on a launch-bound tower) as one launch each
# way, bit-identical (bma_quick_gelu)
FUSE_QUICK_GELU = _os.environ.get("BMA_FUSE_QUICK_GELU", "1") not in _OFF
# the vision tower's q/k/v projections (with their biases) as one product, forward and input-gradient
FUSE_TOWER_QKV = _os.environ.get("BMA_FUSE_TOWER_QKV", "1") not in _OFF
# CLIP's pre-LN encoder layers: each residual add fused into the LayerNorm that follows it (also across the layer boundary),
# forward and backward -- four 5-us launches per layer and pass fewer on a launch-bound tower (bma_add_layernorm)
FUSE_TOWER_LAYERNORM = _os.environ.get("BMA_FUSE_TOWER_LAYERNORM", "1") not in _OFF
# scoring forwards: the rows whose logits are read are gathered in front of the LAST decoder layer's MLP instead of in front
# of the head (HFAdapter._logits_of_rows)
KEEP_ROWS_EARLY = _os.environ.get("BMA_KEEP_ROWS_EARLY", "1") not in _OFF

# modeling_clip.CLIPEncoderLayer.forward (and modeling_siglip.SiglipEncoderLayer.forward), statement for statement (every line that touches `self.`, `residual` or returns)
# The pre-LN encoder block the fused tower forward restates (CLIPEncoderLayer / SiglipEncoderLayer.forward of the transformers
# this was written against), as source: admitted layers must parse to the SAME tree -- every statement, the keyword list of
# the attention call included -- under the signature (self, hidden_states, attention_mask, **kwargs).
_CLIP_LAYER_SOURCE = """
def forward(self, hidden_states, attention_mask, **kwargs):
    residual = hidden_states
    hidden_states = self.layer_norm1(hidden_states)
    hidden_states, _ = self.self_attn(hidden_states=hidden_states, attention_mask=attention_mask, **kwargs)
    hidden_states = residual + hidden_states
    residual = hidden_states
    hidden_states = self.layer_norm2(hidden_states)
    hidden_states = self.mlp(hidden_states)
    hidden_states = residual + hidden_states
    return hidden_states
"""


def _forward_shape(src: str):
    """(argument names incl. *args / **kwargs markers, ast.dump of every body statement without the docstring) of the one
    function definition in `src`; None when it does not parse to one."""
    import ast
    import textwrap
    try:
        tree = ast.parse(textwrap.dedent(src))
    except SyntaxError:
        return None
    if len(tree.body) != 1 or not isinstance(tree.body[0], ast.FunctionDef):
        return None
    fn = tree.body[0]
    a = fn.args
    names = [x.arg for x in a.posonlyargs + a.args] + (["*" + a.vararg.arg] if a.vararg else []) + \
            [x.arg for x in a.kwonlyargs] + (["**" + a.kwarg.arg] if a.kwarg else [])
    body = fn.body
    if body and isinstance(body[0], ast.Expr) and isinstance(getattr(body[0], "value", None), ast.Constant) and isinstance(body[0].value.value, str):
        body = body[1:]
    return names, [ast.dump(st) for st in body]


_CLIP_LAYER_SHAPE = _forward_shape(_CLIP_LAYER_SOURCE)


def _clip_layer_ok(layer) -> bool:
    """Is this encoder layer's forward, statement for statement AND argument for argument, the pre-LN block the fused forward
    restates -- compared as syntax trees (a line filter let another keyword of the attention call through, and the
    restated forward would have dropped it silently) -- with plain affine nn.LayerNorm norms?"""
    try:
        src = inspect.getsource(type(layer).forward)
    except (OSError, TypeError):
        return False
    if _forward_shape(src) != _CLIP_LAYER_SHAPE:
        return False
    for n in ("layer_norm1", "layer_norm2"):
        m = getattr(layer, n, None)
        if type(m) is not torch.nn.LayerNorm or m.weight is None or m.bias is None or len(m.normalized_shape) != 1:
            return False
    return hasattr(layer, "self_attn") and hasattr(layer, "mlp")


class HFAdapter:
    def __init__(self, model, processor, normalize=None):
        self.model = model
        self.processor = processor
        self.normalize = CapturableNormalize(normalize) if callable(normalize) else normalize
        self.embedding = model.get_input_embeddings()
        self.device = model.device
        self.dtype = model.dtype
        self.model_type = getattr(model.config, "model_type", "")
        self.is_gemma_processor = processor.__class__.__name__ == "Gemma3Processor"
        params = inspect.signature(model.forward).parameters
        self.has_logits_to_keep = "logits_to_keep" in params
        scale = getattr(self.embedding, "embed_scale", None)
        # Gemma's embedding multiplies by sqrt(D) held in the weight dtype (:1142 via HF)
        self.emb_scale = 1.0 if scale is None else float(torch.as_tensor(scale).to(self.embedding.weight.dtype).float())
        tc = getattr(model.config, "text_config", None) or model.config
        self.n_layers = int(getattr(tc, "num_hidden_layers", 0) or 0)
        heads = int(getattr(tc, "num_attention_heads", 1) or 1)
        kv_heads = int(getattr(tc, "num_key_value_heads", None) or heads)
        hidden = int(getattr(tc, "hidden_size", 0) or 0)
        head_dim = int(getattr(tc, "head_dim", None) or (hidden // heads if heads else 0))
        inter = int(getattr(tc, "intermediate_size", None) or getattr(tc, "ffn_dim", None) or 4 * hidden)
        self.head_dim, self.heads, self.kv_heads = head_dim, heads, kv_heads
        es = torch.empty((), dtype=self.dtype).element_size()
        self.kv_bytes_per_token = 2 * self.n_layers * kv_heads * head_dim * es
        self.act_bytes_per_token = (16 * hidden + 4 * inter) * es
        self.fused = None                       # the engine's FusedInference (attack.py sets it): _logits_of_rows asks it
        self.prefix_ok: Optional[bool] = None   # learnt on first use
        # shared-prefix attention (prefix_attention.py): None = not probed, [] = not applicable
        self._shared_cfgs = None
        self.shared_window: Optional[int] = None
        self.shared_ok: Optional[bool] = None
        self.ragged_ok: Optional[bool] = None
        self.pad_vision_heads = PAD_VISION_HEADS
        self._vision_cfgs = None
        self.fuse_quick_gelu = FUSE_QUICK_GELU   # CLIP's MLP activation as one launch each way
        self._quick_gelus = None
        self.fuse_tower_qkv = FUSE_TOWER_QKV     # the tower's q/k/v projections as one product
        self._tower_attn = None
        self._proj_norms = None
        self.fuse_tower_layernorm = FUSE_TOWER_LAYERNORM
        self._tower_layers = None

    # ------------------------------------------------------------ vision
    def vision_configs(self) -> list:
        """Vision-tower attention configs to switch to the padded-head attention (prefix_attention.py), [] when
        the tower's head width is one the library handles as is (CLIP: 64) or the option is off."""
        if self._vision_cfgs is None:
            from . import prefix_attention as pa
            ok = self.device.type == "cuda" and pa.register()
            self._vision_cfgs = pa.vision_configs(self.model) if ok else []
        return self._vision_cfgs if self.pad_vision_heads else []

    def quick_gelu_modules(self) -> list:
        """The model's QuickGELUActivation modules (CLIP's MLPs) whose forward is, to the letter, the expression
        bma_quick_gelu restates; [] on the CPU or with the option off."""
        if self._quick_gelus is None:
            import inspect
            found = []
            if self.device.type == "cuda":
                for m in self.model.modules():
                    if type(m).__name__ != "QuickGELUActivation":
                        continue
                    try:
                        src = inspect.getsource(type(m).forward)
                    except (OSError, TypeError):
                        continue
                    if "return input * torch.sigmoid(1.702 * input)" in src:
                        found.append(m)
            self._quick_gelus = found
        return self._quick_gelus if self.fuse_quick_gelu else []

    def projector_norms(self) -> list:
        """RMSNorm modules of the multimodal projector (Gemma-3's ``mm_soft_emb_norm``)."""
        if self._proj_norms is None:
            proj = getattr(self.model, "multi_modal_projector", None) or getattr(getattr(self.model, "model", None),
                                                                                  "multi_modal_projector", None)
            self._proj_norms = [] if proj is None else [m for m in proj.modules() if type(m).__name__.endswith("RMSNorm")]
        return self._proj_norms

    def tower_attention_modules(self) -> list:
        """The vision tower's attention blocks (CLIP / SigLIP modelling files) whose q_proj / k_proj / v_proj are plain
        16-bit Linear layers over one input, applied in that order (read from the block's source): their three products
        run as one (`_tower_qkv_forwards`).  [] on the CPU or with the option off."""
        if self._tower_attn is None:
            found = []
            if self.device.type == "cuda":
                for m in self.model.modules():
                    if type(m).__module__.rsplit(".", 1)[-1] not in ("modeling_clip", "modeling_siglip") or \
                            not type(m).__name__.endswith("Attention"):
                        continue
                    lins = [getattr(m, n, None) for n in ("q_proj", "k_proj", "v_proj")]
                    if not all(type(l) is torch.nn.Linear for l in lins) or len({l.in_features for l in lins}) != 1 or \
                            lins[0].weight.dtype not in (torch.bfloat16, torch.float16) or \
                            len({l.bias is None for l in lins}) != 1:
                        continue
                    try:
                        src = inspect.getsource(type(m).forward)
                    except (OSError, TypeError):
                        continue
                    iq, ik, iv = (src.find(f"self.{n}(hidden_states)") for n in ("q_proj", "k_proj", "v_proj"))
                    if 0 <= iq < ik < iv:
                        found.append(m)
            self._tower_attn = found
        return self._tower_attn if self.fuse_tower_qkv else []

    def tower_layers(self) -> list:
        """[(layer, the encoder layer behind it or None)] of the CLIP vision tower whose blocks the fused add + LayerNorm forward
        restates (`_clip_layer_ok`); [] on the CPU or with the switch off."""
        if self._tower_layers is None:
            found = []
            if self.device.type == "cuda":
                for parent in self.model.modules():
                    layers = getattr(parent, "layers", None)
                    if not isinstance(layers, torch.nn.ModuleList) or len(layers) == 0:
                        continue
                    # (SigLIP's encoder layer -- Gemma-3's tower -- is the same block, statement for statement)
                    if not all((type(l).__module__.rsplit(".", 1)[-1], type(l).__name__) in
                               (("modeling_clip", "CLIPEncoderLayer"), ("modeling_siglip", "SiglipEncoderLayer"))
                               and _clip_layer_ok(l) for l in layers):
                        continue
                    for i, l in enumerate(layers):
                        found.append((l, layers[i + 1] if i + 1 < len(layers) else None))
            self._tower_layers = found
        return self._tower_layers if self.fuse_tower_layernorm else []

    def _tower_layer_forward(self, layer, nxt, stash):
        """CLIPEncoderLayer.forward with each residual add fused into the LayerNorm behind it -- the layer's own layer_norm2, and
        across the layer boundary the NEXT layer's layer_norm1, whose result is handed over through `stash` and picked up when
        that layer is called on the very tensor this one returned.  Same statements, same rounding points (the sum in the model
        dtype, the norm from it)."""
        from . import ops

        def ln(norm, x):
            if not ops.layernorm_ok(x, norm.weight, norm.bias):
                return norm(x)
            if torch.is_grad_enabled() and x.requires_grad:
                return ops.LayerNormFn.apply(x, norm.weight, norm.bias, norm.eps)
            return ops.add_layernorm(None, x, norm.weight, norm.bias, norm.eps)[1]

        def add_ln(residual, h, norm):
            if not (h.shape == residual.shape and h.dtype == residual.dtype and h.is_contiguous() and residual.is_contiguous()
                    and ops.layernorm_ok(h, norm.weight, norm.bias)):
                s = residual + h
                return s, norm(s)
            if torch.is_grad_enabled() and (residual.requires_grad or h.requires_grad):
                return ops.AddLayerNormFn.apply(residual, h, norm.weight, norm.bias, norm.eps)
            s, y, _ = ops.add_layernorm(residual, h, norm.weight, norm.bias, norm.eps)
            return s, y

        def forward(hidden_states, attention_mask=None, **kwargs):
            hit = stash.pop(id(layer), None)
            if hit is None and not hidden_states.is_contiguous():
                # SigLIP's patch embedding hands over `conv(...).flatten(2).transpose(1, 2)`: a TRANSPOSED view, and aten's
                # adds keep that layout for the whole residual stream -- every LayerNorm (forward and backward) then makes
                # its own contiguous copy and every add runs strided: six 25-us launches per layer on a 4096 x 1152 tower
                # (tools/copy_probe.py of round 5 (git history)).  One copy here instead; the values are the same.
                hidden_states = hidden_states.contiguous()
            residual = hidden_states
            h = hit[1] if (hit is not None and hit[0] is hidden_states) else ln(layer.layer_norm1, hidden_states)
            h, _ = layer.self_attn(hidden_states=h, attention_mask=attention_mask, **kwargs)
            residual, h = add_ln(residual, h, layer.layer_norm2)
            h = layer.mlp(h)
            if nxt is None:
                return residual + h
            out, normed = add_ln(residual, h, nxt.layer_norm1)
            stash[id(nxt)] = (out, normed)
            return out
        return forward

    def _tower_qkv_forwards(self, attn):
        """q_proj, k_proj and v_proj of one tower attention block as ONE product against the concatenated weight (and
        bias): on a 577-token tower three 11 us launches forward and three input-gradient products plus two accumulation
        adds backward become one product each way and one concatenation.  As fused.py does for the language model: the
        first of the three calls does the product, the other two hand out their column slices (same input tensor,
        checked by identity)."""
        from .fused import _COPY_CACHES, _CopyCache
        mods = (attn.q_proj, attn.k_proj, attn.v_proj)
        origs = [type(m).forward.__get__(m) for m in mods]
        sizes = [m.out_features for m in mods]
        cache = _COPY_CACHES.get(self.model)
        if cache is None:
            cache = _COPY_CACHES[self.model] = _CopyCache()
        srcs = tuple(m.weight for m in mods) + tuple(m.bias for m in mods if m.bias is not None)
        slot = {}

        def fused():
            hit = cache.get(("tower_wqkv", id(attn)), srcs)
            if hit is None:
                if torch.cuda.is_current_stream_capturing():
                    return None
                with torch.no_grad():
                    w = torch.cat([m.weight.detach() for m in mods], dim=0).contiguous()
                    b = None if mods[0].bias is None else torch.cat([m.bias.detach() for m in mods]).contiguous()
                hit = cache.put(("tower_wqkv", id(attn)), (w, b), srcs)
            return hit

        def first(x):
            slot.clear()
            if not (x.is_cuda and x.dtype == mods[0].weight.dtype and x.dim() >= 2):
                return origs[0](x)
            wb = fused()
            if wb is None:
                return origs[0](x)
            parts = torch.split(torch.nn.functional.linear(x, wb[0], wb[1]), sizes, dim=-1)
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

        return first, later(1), later(2)

    @contextlib.contextmanager
    def _fused_activations(self):
        """For one tower forward: QuickGELU as ONE launch (three aten kernels, and five more in their autograd backward,
        on a launch-bound 577-token tower; bit-identical)."""
        from . import ops
        mods = self.quick_gelu_modules()

        def forward(x):
            if not ops.quick_gelu_ok(x):
                return x * torch.sigmoid(1.702 * x)
            if torch.is_grad_enabled() and x.requires_grad:
                return ops.QuickGELUFn.apply(x)
            return ops.quick_gelu(x)
        attns = self.tower_attention_modules()
        # the projector's norms (Gemma-3: mm_soft_emb_norm) are handed a TRANSPOSED view of the pooled patches; ATen's
        # mean over that strided dim is a multi-block reduction (scratch buffer + semaphores) which, replayed from a
        # hipGraph on this stack, returned NaN rows (attack.image_features).  A contiguous input keeps it a one-block-
        # per-row reduction.  (Only where the engine's fused context has not already replaced the norm's forward.)
        proj_norms = [m for m in self.projector_norms() if "forward" not in m.__dict__]
        layers = self.tower_layers()
        stash = {}                                  # id(next layer) -> (sum, its layer_norm1 of it): lives for this one tower forward
        try:
            for m in mods:
                m.forward = forward
            for a in attns:
                a.q_proj.forward, a.k_proj.forward, a.v_proj.forward = self._tower_qkv_forwards(a)
            for m in proj_norms:
                m.__dict__["forward"] = (lambda x, _f=type(m).forward.__get__(m): _f(x.contiguous()))
            for l, nxt in layers:
                l.__dict__["forward"] = self._tower_layer_forward(l, nxt, stash)
            yield
        finally:
            stash.clear()
            for l, _ in layers:
                l.__dict__.pop("forward", None)
            for m in mods:
                m.__dict__.pop("forward", None)
            for a in attns:
                for lin in (a.q_proj, a.k_proj, a.v_proj):
                    lin.__dict__.pop("forward", None)
            for m in proj_norms:
                m.__dict__.pop("forward", None)

    def image_features(self, image: torch.Tensor) -> torch.Tensor:
        px = self.normalize(image)
        cfgs = self.vision_configs()
        if cfgs:
            from . import prefix_attention as pa
            ctx = pa.causal_b1(cfgs, pa.NAME_VIS)
        else:
            ctx = contextlib.nullcontext()
        with ctx, self._fused_activations():
            if self.is_gemma_processor:
                out = self.model.get_image_features(pixel_values=px)
            else:
                out = self.model.get_image_features(pixel_values=px, vision_feature_layer=-2,
                                                    vision_feature_select_strategy="default")
        return features_tensor(out)

    # ------------------------------------------------------------ language model
    def target_logits(self, embeds: torch.Tensor, T: int, rows_only: bool = True, cache=None) -> torch.Tensor:
        """Logits (B,T,V) of the T positions that predict the target tokens.

        ``embeds`` is the whole sequence (B,S,D) -- or, with ``cache``, only the part
        behind the cached prefix -- INCLUDING the last target token when
        ``rows_only`` is False (reference call shape) and EXCLUDING it otherwise."""
        kw = {}
        if cache is not None:
            kw["past_key_values"] = cache
        else:
            kw["use_cache"] = False
        if rows_only and self.has_logits_to_keep:
            # An index TENSOR, not the int: HF then gathers the T rows into a contiguous
            # (B,T,D) block and the lm_head runs as ONE (B*T, D) x (D, V) GEMM.  With the int
            # it slices a strided view and the head degenerates into B small batched GEMMs
            # (16 ms instead of ~3 ms per step at B=512 on MI355X).
            L = embeds.shape[1]
            keep = self._keep_index(L, T, embeds.device)
            return self._logits_of_rows(keep, inputs_embeds=embeds, **kw)
        if rows_only:
            return self.model(inputs_embeds=embeds, **kw).logits[:, -T:, :]
        logits = self.model(inputs_embeds=embeds, **kw).logits
        return logits[:, -T - 1:-1, :]

    def _logits_of_rows(self, keep: torch.Tensor, **kw) -> torch.Tensor:
        """``model(..., logits_to_keep=keep).logits`` -- with the gather moved from in front of the head to in front of
        the LAST decoder layer's MLP when the engine's fused layer forward is installed and no gradient is wanted
        (fused.FusedInference.keep_rows): everything behind the last attention block is row-wise, and only these rows'
        logits are read (reference: the loss is taken on the target-predicting positions, gcg.py:803-820).  Round-4
        VERDICT item 8a."""
        f = self.fused
        if (KEEP_ROWS_EARLY and f is not None and f.enabled and f.depth > 0 and f._last_layers and f.tp is None
                and not torch.is_grad_enabled()):
            f.keep_rows, f.kept = keep, False
            try:
                logits = self.model(logits_to_keep=0, **kw).logits       # (0: the head on every row it is handed)
            finally:
                f.keep_rows = None
            if not f.kept:            # the last layer ran HuggingFace's own forward after all (positional arguments)
                logits = logits.index_select(1, keep)
            return logits
        return self.model(logits_to_keep=keep, **kw).logits

    def _keep_index(self, L: int, T: int, device) -> torch.Tensor:
        # Entries are NEVER evicted: a captured hipGraph (winner re-score, gradient pass) holds the
        # raw pointer of the tensor it was captured with; dropping it would let the allocator hand
        # the memory to someone else and the replay would gather rows at garbage indices.
        key = (L, T, str(device))
        cache = self.__dict__.setdefault("_keep_cache", {})
        if key not in cache:
            cache[key] = torch.arange(L - T, L, device=device)
        return cache[key]

    def build_prefix(self, prefix_embeds: torch.Tensor):
        """Keys/values of the shared prefix (1,P,D) -> an HF cache object, or None when
        the model does not hand one back."""
        kw = {"logits_to_keep": 1} if self.has_logits_to_keep else {}
        out = self.model(inputs_embeds=prefix_embeds, use_cache=True, **kw)
        return getattr(out, "past_key_values", None)

    def shared_prefix_configs(self, total_len: int = 0) -> list:
        """Text-layer configs to switch to the shared-prefix attention, [] when it does not apply -- to this
        model at all, or to a sequence of `total_len` tokens (longer than a sliding window)."""
        if self._shared_cfgs is None:
            from . import prefix_attention as pa
            ok = self.device.type == "cuda" and self.has_logits_to_keep and pa.register()
            self._shared_cfgs = pa.eligible_configs(self.model) if ok else []
            self.shared_window = pa.min_sliding_window(self.model) if self._shared_cfgs else None
        if self._shared_cfgs and self.shared_window is not None and total_len > self.shared_window:
            return []
        return self._shared_cfgs

    def build_prefix_recording(self, prefix_embeds: torch.Tensor):
        """Prefix keys/values through a RecordingKV (prefix_attention.py): for the shared-prefix
        attention path; no HuggingFace cache object is involved."""
        from . import prefix_attention as pa
        rec = pa.RecordingKV(self.n_layers)
        kw = {"logits_to_keep": 1} if self.has_logits_to_keep else {}
        self.model(inputs_embeds=prefix_embeds, past_key_values=rec, **kw)
        if any(k is None for k in rec.k):
            raise RuntimeError("the model did not report keys/values for every layer")
        return rec

    def target_logits_shared_prefix(self, embeds: torch.Tensor, T: int, cache) -> torch.Tensor:
        """Like ``target_logits(..., cache=expand_prefix(cache, B))`` but the prefix keys/values
        are never copied per candidate (prefix_attention.py).  embeds: (B,L,D), the part behind
        the prefix, without the last target token."""
        from . import prefix_attention as pa
        kv = pa.SharedPrefixKV(cache)
        keep = self._keep_index(embeds.shape[1], T, embeds.device)
        with pa.active(self.shared_prefix_configs(), kv):
            return self._logits_of_rows(keep, inputs_embeds=embeds, past_key_values=kv)

    def target_logits_behind_grad_prefix(self, embeds: torch.Tensor, T: int, rec) -> torch.Tensor:
        """Gradient pass with the prefix reused: `embeds` (1,L,D) are the tokens behind a prefix whose recorded
        keys/values (`rec`, a RecordingKV filled under autograd) stay differentiable; returns the (1,T,V) logits
        of the target-predicting rows."""
        from . import prefix_attention as pa
        kv = pa.GradPrefixKV(rec)
        keep = self._keep_index(embeds.shape[1], T, embeds.device)
        with pa.active(self.shared_prefix_configs(), kv, pa.NAME_TAIL):
            return self.model(inputs_embeds=embeds, past_key_values=kv, logits_to_keep=keep).logits

    def target_logits_ragged(self, rows: torch.Tensor, T: int, cache, maps) -> torch.Tensor:
        """Ragged scoring (layout.ragged_plan): `rows` (1,N,D) holds, per candidate, only the tokens
        from its first replaced suffix position on (and the parent suffix in front); returns the
        (m,T,V) logits of the target-predicting rows."""
        from . import prefix_attention as pa
        kv = pa.SharedPrefixKV(cache)
        kv.ragged = maps
        with pa.active(self.shared_prefix_configs(), kv):
            logits = self._logits_of_rows(maps.keep, inputs_embeds=rows, past_key_values=kv, position_ids=maps.pos)
        return logits.view(maps.m_out, T, logits.shape[-1])

    @staticmethod
    def expand_prefix(cache, batch: int):
        """A per-chunk copy of the prefix cache with batch dimension `batch`.  The
        model's own cache update appends the candidates' keys/values to it."""
        c = copy.deepcopy(cache)
        layers = getattr(c, "layers", None)
        if layers is not None and all(hasattr(l, "keys") and hasattr(l, "values") for l in layers):
            for l in layers:
                if torch.is_tensor(l.keys) and l.keys.shape[0] == 1:
                    # stride-0 view: the one real copy happens in the cache's own concat
                    l.keys = l.keys.expand(batch, *l.keys.shape[1:])
                    l.values = l.values.expand(batch, *l.values.shape[1:])
            return c
        c.batch_repeat_interleave(batch)
        return c
