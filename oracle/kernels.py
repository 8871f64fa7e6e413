"""ORACLE -- test infrastructure, not product code.

CPU restatement (numpy) of the small per-step pieces of the reference's hot
path.  Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg may import this package; the product
(``bimodalattack_amd``) never does and fails loudly without its HIP library.

Parity status: PINNED.  Every function here is checked against golden vectors
captured from the real reference (``tests/golden/make_golden.py`` ->
``tests/golden/g1..g6``) by ``tests/test_oracle_golden.py``.

Citations are into /root/reference/bimodalattack/bimodal_attack.py unless a
file is named.
"""

from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np


# --------------------------------------------------------------------------
# a5 -- PGD L-inf step (reference :1030-1037)
# --------------------------------------------------------------------------
def linf_step(x: np.ndarray, g: np.ndarray, x0: np.ndarray, eps: float, alpha: float) -> np.ndarray:
    """x' = clamp(clamp(x - (alpha*eps)*sign(g), x0-eps, x0+eps), 0, 1) in fp32.

    The reference's step is alpha*eps (a Python double product, :1033) applied
    as an fp32 scalar; the ball radius is eps as an fp32 scalar (:1034)."""
    x = np.asarray(x, np.float32)
    g = np.asarray(g, np.float32)
    x0 = np.asarray(x0, np.float32)
    step = np.float32(alpha * eps)
    e = np.float32(eps)
    sgn = (g > 0).astype(np.float32) - (g < 0).astype(np.float32)   # torch.sign: 0 for NaN and for -0.0
    y = (x - step * sgn).astype(np.float32)
    lo = (x0 - e).astype(np.float32)
    hi = (x0 + e).astype(np.float32)
    y = np.where(y < lo, lo, y)
    y = np.where(y > hi, hi, y)
    y = np.where(y < np.float32(0), np.float32(0), y)
    y = np.where(y > np.float32(1), np.float32(1), y)
    return y.astype(np.float32)


# --------------------------------------------------------------------------
# a2 -- cross-entropy over the target slice (reference :1006-1012, :1289-1298)
# --------------------------------------------------------------------------
def ce_rows(logits: np.ndarray, labels: np.ndarray) -> np.ndarray:
    """Per-row CE of logits (..., T, V) against labels (T,): logsumexp - x[label].
    Accumulates in float64: this is the 'truth' both the reference's fp32 result
    and the kernel's fp32 result are compared to."""
    x = np.asarray(logits, np.float64)
    m = x.max(axis=-1, keepdims=True)
    lse = np.log(np.exp(x - m).sum(axis=-1)) + m[..., 0]
    lab = np.asarray(labels, np.int64).reshape(-1)
    picked = np.take_along_axis(x, np.broadcast_to(lab.reshape((1,) * (x.ndim - 2) + (-1, 1)), x.shape[:-1] + (1,)), axis=-1)[..., 0]
    return lse - picked


def ce_target(logits: np.ndarray, labels: np.ndarray):
    """Candidate scoring (:1289-1306): logits (B,T,V), labels (T,) ->
    (loss (B,) = mean over T, match (B,) = every row's argmax equals its label)."""
    rows = ce_rows(logits, labels)
    lab = np.asarray(labels, np.int64).reshape(-1)
    match = (np.argmax(np.asarray(logits), axis=-1) == lab[None, :]).all(axis=-1)
    return rows.mean(axis=-1), match


def ce_target_grad(logits: np.ndarray, labels: np.ndarray, scale: float = 1.0) -> np.ndarray:
    """d(mean CE)/d(logits) for one sequence (T,V): (softmax - onehot) / T * scale
    (what autograd hands back through :1010-1012)."""
    x = np.asarray(logits, np.float64)
    T = x.shape[-2]
    m = x.max(axis=-1, keepdims=True)
    p = np.exp(x - m)
    p /= p.sum(axis=-1, keepdims=True)
    lab = np.asarray(labels, np.int64).reshape(-1)
    p[np.arange(T), lab] -= 1.0
    return p * (scale / T)


# --------------------------------------------------------------------------
# a3 -- candidate sampling (reference :130-163)
# --------------------------------------------------------------------------
def mask_topk(grad: np.ndarray, not_allowed: Optional[np.ndarray], k: int) -> np.ndarray:
    """Per position, the k token ids with the most negative gradient among the
    allowed ones (:144-147): not-allowed columns count as +inf.  Order: gradient
    ascending, ties by lower token id (the build's tie policy; the reference's
    own goldens are tie-free).  NaN gradients rank first, as torch.topk(-grad)
    ranks NaN above everything."""
    g = np.array(grad, dtype=np.float32, copy=True)
    if not_allowed is not None and len(not_allowed):
        g[:, np.asarray(not_allowed, np.int64)] = np.inf
    g = np.where(np.isnan(g), -np.inf, g)
    order = np.argsort(g, axis=1, kind="stable")
    return order[:, :k].astype(np.int64)


def rand_positions(rnd: np.ndarray, n_replace: int) -> np.ndarray:
    """argsort(rand)[..., :n_replace] (:150-154): the n_replace positions with the
    smallest random keys, in that order."""
    return np.argsort(np.asarray(rnd), axis=1, kind="stable")[:, :n_replace].astype(np.int64)


def sample_scatter(ids: np.ndarray, topk_idx: np.ndarray, pos: np.ndarray, rank: np.ndarray) -> np.ndarray:
    """new_ids[b] = ids with position pos[b,j] replaced by topk_idx[pos[b,j], rank[b,j]]
    (:142, :156-162)."""
    ids = np.asarray(ids, np.int64).reshape(-1)
    B, n_rep = pos.shape
    out = np.tile(ids, (B, 1))
    for j in range(n_rep):
        out[np.arange(B), pos[:, j]] = topk_idx[pos[:, j], rank[:, j]]
    return out


def sample_ids_from_grad(ids, grad, topk: int, n_replace: int, not_allowed, rnd, rank) -> np.ndarray:
    """The whole of :130-163 with the two random draws passed in as inputs
    (rnd: (sw, n_opt) uniform floats; rank: (sw, n_replace) ints in [0, topk))."""
    tk = mask_topk(grad, not_allowed, topk)
    pos = rand_positions(rnd, n_replace)
    return sample_scatter(ids, tk, pos, np.asarray(rank, np.int64))


def dynamic_width(step: int, search_width: int, num_steps: int, min_search_width: int, dynamic: bool) -> int:
    """:919-928."""
    if not dynamic:
        return search_width
    return max(min_search_width, int(search_width * (1 - step / num_steps)))


# --------------------------------------------------------------------------
# a7 -- embedding splice (reference :1112-1225)
# --------------------------------------------------------------------------
def segment_order(mode: str, model_type: str, single: bool = False, no_joint_eval: bool = False,
                  no_target: bool = False) -> List[str]:
    """Which segments, in which order (:1150-1215).  ``gemma3`` puts the suffix in
    front of the image; everything else puts it behind."""
    g = model_type == "gemma3"
    with_img = (["before_img", "optim", "before_suffix", "image", "after"] if g
                else ["before_img", "image", "before_suffix", "optim", "after"])
    if mode == "pgd":
        if not single:
            raise AssertionError("PGD mode only supports single=True")
        return with_img + ["target"]
    if mode == "gcg":
        if single:
            return (["before_img", "optim", "before_suffix", "after", "target"] if g
                    else ["before_img", "before_suffix", "optim", "after", "target"])
        if no_joint_eval:
            return ["before", "optim", "after", "target"]
        if no_target:
            return ["before", "optim", "after"]
        raise ValueError("Invalid flags for BimodalAttack mode")
    if mode == "gcg_pgd":
        if not single and no_target:
            return with_img
        return with_img + ["target"]
    raise ValueError(f"Unknown mode '{mode}'")


def splice(order: Sequence[str], segments: Dict[str, np.ndarray], table: np.ndarray, ids: np.ndarray,
           search_width: Optional[int], emb_scale: Optional[float] = None) -> np.ndarray:
    """Concatenate segments along the sequence axis (:1217-1225).  'optim' is the
    embedding gather table[ids] (times the Gemma embedding scale when the model's
    embedding layer has one, :1142); every other segment is (1,L,D) and is
    repeated to ``search_width`` rows when search_width is given."""
    ids = np.asarray(ids, np.int64)
    parts = []
    for name in order:
        if name == "optim":
            t = table[ids]
            if emb_scale is not None:
                t = (t * table.dtype.type(emb_scale)).astype(table.dtype)
        else:
            t = segments[name]
        if search_width is not None and t.shape[0] == 1:
            t = np.repeat(t, search_width, axis=0)
        parts.append(t)
    return np.concatenate(parts, axis=1)


# --------------------------------------------------------------------------
# a4 / a9 -- tokenizer-side helpers
# --------------------------------------------------------------------------
def nonascii_tokens(tokenizer) -> np.ndarray:
    """reference utils.py:14-33: ids in range(vocab_size) whose decoded text is not
    printable ASCII, then bos/eos/pad/unk appended (duplicates kept)."""
    bad = [i for i in range(tokenizer.vocab_size)
           if not (lambda s: s.isascii() and s.isprintable())(tokenizer.decode([i]))]
    for attr in ("bos_token_id", "eos_token_id", "pad_token_id", "unk_token_id"):
        v = getattr(tokenizer, attr)
        if v is not None:
            bad.append(v)
    return np.asarray(bad, np.int64)


def filter_ids(ids: np.ndarray, tokenizer) -> np.ndarray:
    """:166-186, one string at a time: keep rows that decode -> encode back to
    themselves.  Raises RuntimeError when nothing survives."""
    ids = np.asarray(ids, np.int64)
    texts = tokenizer.batch_decode(ids.tolist())
    keep = []
    for row, text in zip(ids, texts):
        again = tokenizer(text, add_special_tokens=False)["input_ids"]
        if len(again) == len(row) and all(int(a) == int(b) for a, b in zip(again, row)):
            keep.append(row)
    if not keep:
        raise RuntimeError(
            "No token sequences are the same after decoding and re-encoding. "
            "Consider setting filter_ids=False or trying a different optim_str_init"
        )
    return np.stack(keep)
