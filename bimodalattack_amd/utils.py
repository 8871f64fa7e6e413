"""Tokenizer-side helpers of the attack loop (reference bimodalattack/utils.py).

* ``get_nonascii_toks``  <- utils.py:14-33   forbidden-token list, built once
* ``filter_ids``         <- bimodal_attack.py:166-186   retokenisation filter
* ``INIT_CHARS``         <- utils.py:8-12
* ``plan_chunk``         replaces utils.py:57-115 (OOM-halving retry, restarted from the
                         full batch on every step) with analytic sizing; the halving
                         survives only as a safety net that REMEMBERS what it learnt.
"""

from __future__ import annotations

import logging
from typing import List, Optional

import torch

logger = logging.getLogger("gcg")

INIT_CHARS = [
    ".", ",", "!", "?", ";", ":", "(", ")", "[", "]", "{", "}",
    "@", "#", "$", "%", "&", "*",
    "w", "x", "y", "z",
]


def get_nonascii_toks(tokenizer, device="cpu") -> torch.Tensor:
    """Token ids that must never be sampled: every id in range(vocab_size) whose decoded
    text is not printable ASCII, then bos/eos/pad/unk (appended even if already listed).
    One batched decode instead of vocab_size single calls; same strings, same result."""
    n = tokenizer.vocab_size
    texts = tokenizer.batch_decode([[i] for i in range(n)])
    bad: List[int] = [i for i, s in enumerate(texts) if not (s.isascii() and s.isprintable())]
    for name in ("bos_token_id", "eos_token_id", "pad_token_id", "unk_token_id"):
        tid = getattr(tokenizer, name, None)
        if tid is not None:
            bad.append(tid)
    return torch.tensor(bad, device=device)


def roundtrip_keep(rows: List[List[int]], tokenizer) -> List[int]:
    """Indices of the candidates whose decode -> encode round trip reproduces them exactly
    (reference :166-186), from ids already on the host.  One batched decode, one batched
    encode, a list comparison; raises like the reference when nothing survives."""
    texts = tokenizer.batch_decode(rows)
    again = tokenizer(texts, add_special_tokens=False, padding=False)["input_ids"]
    keep = [i for i, (a, b) in enumerate(zip(rows, again)) if a == list(b)]
    if not keep:
        raise RuntimeError(
            "No token sequences are the same after decoding and re-encoding. "
            "Consider setting filter_ids=False or trying a different optim_str_init"
        )
    return keep


def filter_ids(ids: torch.Tensor, tokenizer) -> torch.Tensor:
    """Keep the candidates whose decode -> encode round trip reproduces them exactly.

    The reference tokenises one string per call and compares on the device (512 syncs
    per step, 0.18 s at sw=512: SURVEY.md 8 f1).  Here: one device->host copy, one
    batched decode, one batched encode, a host-side list comparison, one gather."""
    rows = ids.tolist()
    keep = roundtrip_keep(rows, tokenizer)
    if len(keep) == len(rows):
        return ids
    return ids[torch.tensor(keep, device=ids.device)]


class FilterJob:
    """The retokenisation filter taken off the critical path: the candidate ids start
    their way to the host (pinned buffer, non-blocking copy) right after sampling, the
    GPU goes on to score EVERY sampled candidate, and the host runs the tokenizer round
    trip meanwhile: in ``result()``, called once the forward is enqueued (nothing on the
    way there may block the host behind the stream -- see dist.dealt_index).  (A worker
    thread for it was measured no faster -- both threads want the interpreter lock while
    the main one enqueues -- and removed in round 4.)
    ``result()`` returns the surviving indices; the caller masks the losses with them --
    the same candidates win as if they had been filtered first.  ``seconds`` is the round
    trip's own duration, ``waited`` what the caller actually spent blocked in ``result()``."""

    def __init__(self, ids: torch.Tensor, tokenizer, enabled: bool, pinned: Optional[torch.Tensor] = None):
        self.n = ids.shape[0]
        self.tokenizer = tokenizer
        self.enabled = enabled
        self.seconds = 0.0
        self.waited = 0.0
        self._keep: Optional[List[int]] = None
        self._error: Optional[BaseException] = None
        if enabled:
            # `pinned`: a block the caller keeps between steps.  A fresh pinned allocation per step is recycled by
            # PyTorch's host allocator only once the stream has passed the point where the previous one was dropped;
            # with work always queued ahead that is late, and a NEW pinned allocation stops the host behind the stream
            if pinned is not None and pinned.dtype == ids.dtype and pinned.numel() >= ids.numel():
                self.host = pinned[:ids.numel()].view(ids.shape)
            else:
                self.host = torch.empty(ids.shape, dtype=ids.dtype, pin_memory=True)
            self.host.copy_(ids, non_blocking=True)
            self.event = torch.cuda.Event()
            self.event.record(torch.cuda.current_stream(ids.device))

    def _run(self) -> None:
        import time
        try:
            self.event.synchronize()
            t0 = time.perf_counter()
            self._keep = roundtrip_keep(self.host.tolist(), self.tokenizer)
            self.seconds = time.perf_counter() - t0
        except BaseException as e:          # handed to the caller of result(), reference behaviour (:183-186)
            self._error = e

    def result(self) -> List[int]:
        if self._keep is None and self._error is None:
            if not self.enabled:
                self._keep = list(range(self.n))
            else:
                import time
                t0 = time.perf_counter()
                self._run()
                self.waited = time.perf_counter() - t0
        if self._error is not None:
            raise self._error
        return self._keep


def is_oom(exc: BaseException) -> bool:
    """The messages the reference matches (utils.py:39-54) plus HIP's spelling."""
    if not (isinstance(exc, RuntimeError) and len(exc.args) == 1 and isinstance(exc.args[0], str)):
        return isinstance(exc, torch.OutOfMemoryError) if hasattr(torch, "OutOfMemoryError") else False
    msg = exc.args[0]
    return any(s in msg for s in ("CUDA out of memory.", "HIP out of memory.", "out of memory",
                                  "DefaultCPUAllocator: can't allocate memory"))


def plan_chunk(n_candidates: int, new_tokens: int, prefix_tokens: int, kv_bytes_per_token: int,
               act_bytes_per_token: int, free_bytes: int, user_batch: Optional[int] = None,
               token_budget: int = 49152, quantum: int = 1) -> int:
    """Candidates per forward.  ``user_batch`` (config.batch_size) wins when given, as in
    the reference (:521-523).  Otherwise bound (a) the new tokens in flight and (b) the
    memory of activations plus the per-candidate copy of the shared-prefix keys/values,
    using at most half of the free HBM; a bound of `quantum` candidates or more is rounded down
    to a multiple of it (the engine pads a short last chunk up to one: GEMM shapes then come
    from a small set whatever the search width, which a lookup-only GEMM selection can cover)."""
    if user_batch is not None:
        return max(1, min(n_candidates, int(user_batch)))
    per_cand = new_tokens * act_bytes_per_token + (prefix_tokens + new_tokens) * kv_bytes_per_token
    by_mem = max(1, int(free_bytes * 0.5) // max(per_cand, 1))
    by_tok = max(1, token_budget // max(new_tokens, 1))
    cap = min(by_mem, by_tok)
    if quantum > 1 and cap >= quantum:
        cap -= cap % quantum
    return max(1, min(n_candidates, cap))
