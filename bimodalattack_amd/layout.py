"""Host logic of the attack loop that needs no device: which segments make up a
candidate sequence in which mode (reference bimodal_attack.py:1150-1215), the
dynamic search-width schedule (:919-928) and where the shared prefix ends.

Table-driven: a layout is looked up, not branched to.
"""

from __future__ import annotations

from typing import List, Optional, Sequence, Tuple

# Sequence layouts WITH the image, keyed by "is the model gemma3".  Gemma-3 puts
# the suffix in front of the image, everything else behind it (:1153-1162).
_WITH_IMAGE = {
    False: ("before_img", "image", "before_suffix", "optim", "after"),
    True: ("before_img", "optim", "before_suffix", "image", "after"),
}
# Layouts of the PGD prompt scored WITHOUT the image (joint_eval=False, :1166-1171).
_NO_IMAGE = {
    False: ("before_img", "before_suffix", "optim", "after"),
    True: ("before_img", "optim", "before_suffix", "after"),
}
_TEXT_ONLY = ("before", "optim", "after")


def segment_order(mode: str, model_type: str, single: bool = False, no_joint_eval: bool = False,
                  no_target: bool = False) -> List[str]:
    """Names of the segments of one candidate sequence, in order.

    mode "pgd" / "gcg_pgd": prompt with image features; "gcg": text only, either the
    PGD prompt minus the image (``single``) or the GCG-only prompt (``no_joint_eval``
    with the target, ``no_target`` without).  Flag precedence follows the reference:
    ``single`` beats ``no_joint_eval`` beats ``no_target``.
    """
    gemma = model_type == "gemma3"
    if mode == "pgd":
        assert single, "PGD mode only supports single=True"
        return [*_WITH_IMAGE[gemma], "target"]
    if mode == "gcg_pgd":
        drop_target = no_target and not single
        return list(_WITH_IMAGE[gemma]) + ([] if drop_target else ["target"])
    if mode == "gcg":
        if single:
            return [*_NO_IMAGE[gemma], "target"]
        if no_joint_eval:
            return [*_TEXT_ONLY, "target"]
        if no_target:
            return list(_TEXT_ONLY)
        raise ValueError("Invalid flags for BimodalAttack mode")
    raise ValueError(f"Unknown mode '{mode}'")


def dynamic_width(step: int, search_width: int, num_steps: int, min_search_width: int, dynamic: bool) -> int:
    """Candidates to sample at `step`: linear decay to a floor when dynamic_search (:919-923)."""
    if not dynamic:
        return search_width
    return max(min_search_width, int(search_width * (1 - step / num_steps)))


def split_at_suffix(order: Sequence[str]) -> Tuple[List[str], List[str]]:
    """(segments in front of the suffix, suffix and everything behind it).  The first
    part is identical for every candidate of a step, so under causal attention its
    keys/values can be computed once and shared."""
    i = list(order).index("optim")
    return list(order[:i]), list(order[i:])


# ---------------------------------------------------------------------------------------------
# Ragged scoring.  A candidate equals its parent suffix up to the first replaced position p, so
# under causal attention its hidden states at suffix positions < p equal the parent's: only
# tokens j >= p of the L tokens behind the shared prefix are computed (GEMMs, norms and MLP
# gates see 1 - (n_opt-1)/(2L) of the rows), and attention reads the parent's keys/values for
# the rest.  The parent's n_opt suffix tokens ride along as the first rows of the same forward.
def first_diff_stats(n_opt: int, n_replace: int) -> Tuple[float, float]:
    """Mean and variance of the first replaced position when `n_replace` distinct positions are
    drawn uniformly from n_opt (the reference's argsort-of-uniforms draw, :150-152)."""
    from math import comb
    r = max(1, min(int(n_replace), n_opt))
    tot = comb(n_opt, r)
    # P(min >= k) = C(n_opt - k, r) / C(n_opt, r)
    surv = [comb(n_opt - k, r) / tot for k in range(n_opt + 1)]
    pk = [surv[k] - surv[k + 1] for k in range(n_opt)]
    mean = sum(k * q for k, q in enumerate(pk))
    var = sum((k - mean) ** 2 * q for k, q in enumerate(pk))
    return mean, var


def expected_unique(m: int, n_opt: int, n_replace: int, topk: int) -> Tuple[float, float]:
    """Mean and variance of the number of DISTINCT candidates among m draws of (positions, top-k
    ranks) -- the occupancy problem over K = C(n_opt, r) * topk^r equally likely outcomes (:150-160).
    Exact duplicates score identically, so the ragged forward computes each distinct candidate once."""
    from math import comb
    r = max(1, min(int(n_replace), n_opt))
    K = float(comb(n_opt, r)) * float(max(1, topk)) ** r
    if K > 1e12 or m <= 1:
        return float(m), 0.0
    q1, q2 = (1.0 - 1.0 / K) ** m, (1.0 - 2.0 / K) ** m
    mean = K * (1.0 - q1)
    var = max(0.0, K * (K - 1.0) * q2 + K * q1 - K * K * q1 * q1)
    return mean, var


def ragged_rows(needed: int, cap: int) -> int:
    """Rows a ragged forward computes for `needed` useful ones: the next point of a coarse grid (GEMM
    shapes then come from a small set that the selection table covers: tools/tune_rows.py), never more
    than `cap` (every token of every candidate)."""
    gran = 256 if needed >= 8192 else 128 if needed >= 4096 else 64 if needed >= 1024 else 8
    return min(-(-needed // gran) * gran, cap)


def expected_row_counts(m: int, n_opt: int, L: int, n_replace: int, topk: int, worlds=(1,), sigmas: float = 4.5) -> list:
    """The row counts (grid points of ``ragged_rows``) the ragged forwards of a run can meet: every grid point within
    `sigmas` standard deviations of the expected count for `m` sampled candidates of `L` tokens behind the prefix --
    first replaced positions and duplicates are random --, for one rank of each world size in `worlds`.  The GEMM
    selection table is tuned over this set (tools/tune_rows.py) and the engine touches it once before the first step
    (``BimodalAttack._warm_gemms``)."""
    mean_p, var_p = first_diff_stats(n_opt, n_replace)
    u, var_u = expected_unique(m, n_opt, n_replace, topk)
    rows = L - mean_p
    out = set()
    for w in worlds:
        mean = n_opt + u / w * rows
        sd = (u / w * var_p + var_u / (w * w) * rows * rows) ** 0.5
        lo, hi = int(mean - sigmas * sd), int(mean + sigmas * sd) + L
        cap = n_opt + (-(-m // w)) * L
        v = max(n_opt + 1, lo)
        while v <= hi:
            r = ragged_rows(v, cap)
            out.add(r)
            v = r + 1
    return sorted(out)


_HASH_MUL = None


def unique_rows(rows, return_first: bool = False):
    """(distinct rows in order of first appearance, inverse map) of an integer matrix -- what
    ``np.unique(axis=0)`` yields up to the order, ~10x faster at 512 x 19: rows are grouped by a 64-bit
    multiplicative hash and the grouping is then VERIFIED element by element (an unequal pair in one group
    sends the call to np.unique), so the result is exact whatever the hash does.  `return_first`: also the index
    of each distinct row's first appearance in `rows` (uniq == rows[first])."""
    import numpy as np
    global _HASH_MUL
    rows = np.ascontiguousarray(rows)
    n, w = rows.shape
    if n <= 1:
        out = rows.copy(), np.zeros(n, dtype=np.int64)
        return (*out, np.arange(n, dtype=np.int64)) if return_first else out
    if _HASH_MUL is None or _HASH_MUL.shape[0] < w:
        _HASH_MUL = (np.random.RandomState(0x5eed).randint(1, 2 ** 62, size=max(w, 64), dtype=np.int64).astype(np.uint64) << np.uint64(1)) | np.uint64(1)
    with np.errstate(over="ignore"):
        h = (rows.astype(np.uint64) * _HASH_MUL[None, :w]).sum(axis=1, dtype=np.uint64)
        h ^= h >> np.uint64(29)
    _, first_idx, inv = np.unique(h, return_index=True, return_inverse=True)
    order = np.argsort(first_idx, kind="stable")            # groups by first appearance
    rank = np.empty_like(order)
    rank[order] = np.arange(order.shape[0])
    inv = rank[np.asarray(inv).reshape(-1)]
    first = first_idx[order]
    uniq = rows[first]
    if not np.array_equal(uniq[inv], rows):                  # a hash collision: the exact, slower way
        uniq, first, inv = np.unique(rows, axis=0, return_index=True, return_inverse=True)
        inv = np.asarray(inv).reshape(-1)
    if return_first:
        return uniq, inv.astype(np.int64), np.asarray(first, dtype=np.int64)
    return uniq, inv.astype(np.int64)


def ragged_plan(cand, parent, L: int, T: int, P: int, n_rows: Optional[int] = None, dedup: bool = True,
                padded_maps: bool = True, inverse=None):
    """Index maps of one ragged scoring forward, numpy in / numpy out.

    cand (m,n_opt) candidate suffix ids, parent (n_opt,) the ids they were sampled from, L tokens
    per candidate behind the shared prefix (suffix first), T target rows, P prefix length, n_rows
    the row count to build (None: ``ragged_rows`` of what this draw needs); `inverse`: cand holds distinct
    rows already and inverse[i] is the row of original candidate i (``unique_rows``).  Returns None when the draw
    does not fit n_rows, else
      flat  (N,)        row n -> padded slot b*L+j  (parent = block m; also the embedding gather)
      q_src (B2*L,)     padded slot -> row holding its query (any own row where none is computed)
      kv_src(B2*L,)     padded slot -> row holding its key/value (parent rows in front of p)
                        (both None with padded_maps=False: only the library-attention route reads them)
      pos   (N,)        rotary position of row n
      keep  (m_out*T,)  rows that predict the target tokens, candidate-major, for EVERY input
                        candidate (duplicates point at the rows of their one computed copy)
      p     (m,)        first computed position per distinct candidate
      cand  (m,n_opt)   the distinct candidates, in the order the maps number them (first appearance)
      cstart/cfirst/clen (B2,)  per block: first row, first position, row count (bma_ragged_attention)
    with m the number of distinct candidates (all of them with dedup=False), B2 = m + 1, N = n_rows."""
    import numpy as np
    cand = np.asarray(cand)
    parent = np.asarray(parent).reshape(-1)
    m_out = cand.shape[0]
    inv = None
    if inverse is not None:                          # the caller removed the duplicates already
        inv = np.asarray(inverse).reshape(-1)
        m_out = inv.shape[0]
    elif dedup and m_out > 1:
        cand, inv = unique_rows(cand)
    m, n_opt = cand.shape
    if parent.shape[0] != n_opt or L - T < n_opt - 1 or n_opt < 1:
        raise ValueError("ragged_plan: inconsistent shapes")
    diff = cand != parent[None, :]
    p = np.where(diff.any(1), diff.argmax(1), n_opt - 1).astype(np.int64)
    needed = n_opt + int((L - p).sum())
    if n_rows is None:
        n_rows = ragged_rows(needed, n_opt + m * L)
    deficit = n_rows - needed
    if deficit < 0:
        return None
    if deficit > 0:                                  # lower p (recompute a few parent rows): exact fit
        before = np.cumsum(p) - p
        p = p - np.minimum(p, np.maximum(0, deficit - before))
        if int((L - p).sum()) != n_rows - n_opt:
            return None                              # n_rows > n_opt + m*L: more rows than tokens
    lens = L - p
    starts = n_opt + np.cumsum(lens) - lens
    n_c = int(lens.sum())
    tok_i = np.repeat(np.arange(m), lens)
    tok_j = np.arange(n_c) - np.repeat(starts - n_opt, lens) + np.repeat(p, lens)
    flat = np.concatenate([m * L + np.arange(n_opt), tok_i * L + tok_j]).astype(np.int32)
    pos = np.concatenate([np.arange(n_opt), tok_j]).astype(np.int64) + P
    q_src = kv_src = None
    if padded_maps:                 # only the library-attention route reads these (fp32 models, odd head sizes)
        J = np.arange(L)[None, :]
        own = starts[:, None] + (J - p[:, None])
        valid = J >= p[:, None]
        par = np.minimum(np.arange(L), n_opt - 1)
        q_src = np.concatenate([np.where(valid, own, starts[:, None]).reshape(-1), par]).astype(np.int32)
        kv_src = np.concatenate([np.where(valid, own, J).reshape(-1), par]).astype(np.int32)
    # per padded block (the parent is block m): rows it owns, its first position, how many
    cstart = np.concatenate([starts, [0]]).astype(np.int32)
    cfirst = np.concatenate([p, [0]]).astype(np.int32)
    clen = np.concatenate([lens, [n_opt]]).astype(np.int32)
    keep = starts[:, None] + (L - T - p[:, None]) + np.arange(T)[None, :]
    if inv is not None:
        keep = keep[inv]
    keep = keep.reshape(-1).astype(np.int64)
    return dict(m_out=m_out, cand=cand, cstart=cstart, cfirst=cfirst, clen=clen, flat=flat, q_src=q_src, kv_src=kv_src, pos=pos, keep=keep, p=p, m=m, L=L, n_opt=n_opt, N=int(n_rows),
                needed=needed)
