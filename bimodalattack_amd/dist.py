"""Candidate sharding across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  Weights
are replicated; the candidates of a step are independent given (sampled_ids,
image features), so rank r scores the contiguous slice [r*per, (r+1)*per) and the
only data-path exchange is one all-gather of `per` fp32 losses per rank (<= 2 KiB
in total at search_width 512) -- latency-bound, so a single one-shot collective,
never a ring of point-to-point hops.  N varies per step (filter, dynamic width):
slices are padded to `per` slots with +inf, which can never win the argmin.

Rank 0's sampled ids, PGD image and winner loss are broadcast so that ranks cannot
drift apart through last-bit differences in redundantly computed gradients.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class CandidateSharder:
    def __init__(self, group=None):
        self.enabled = dist.is_available() and dist.is_initialized()
        self.group = group
        if self.enabled:
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.enabled = self.enabled and self.world > 1

    # -- partition ---------------------------------------------------------
    def per_rank(self, n: int) -> int:
        return -(-n // self.world)

    def bounds(self, n: int, rank: Optional[int] = None) -> Tuple[int, int]:
        r = self.rank if rank is None else rank
        per = self.per_rank(n)
        lo = min(n, r * per)
        return lo, min(n, lo + per)

    # -- exchange ----------------------------------------------------------
    def gather(self, local: torch.Tensor, n: int, pad: float = float("inf")) -> torch.Tensor:
        """local: this rank's slice (length hi-lo) of a per-candidate fp32 vector.  Returns the
        n values in candidate order on every rank.  One collective of per_rank(n) floats per
        rank; short slices are padded with `pad` (+inf can never win an argmin)."""
        if not self.enabled:
            return local
        per = self.per_rank(n)
        send = torch.full((per,), pad, dtype=torch.float32, device=local.device)
        send[: local.numel()] = local.to(torch.float32)
        if dist.get_backend(self.group) == "nccl":
            recv = torch.empty((self.world * per,), dtype=torch.float32, device=local.device)
            dist.all_gather_into_tensor(recv, send, group=self.group)
        else:
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(parts, send, group=self.group)
            recv = torch.cat(parts)
        return recv[:n].contiguous()

    # -- dealt partition (ragged scoring) -----------------------------------------------
    def deal(self, order, rank: Optional[int] = None):
        """This rank's share of `order` (any sequence of candidate indices): every world-th entry.
        With `order` sorted by a per-candidate cost, every rank gets the same mix of costs."""
        r = self.rank if rank is None else rank
        return order[r::self.world]

    def gather_dealt(self, local: torch.Tensor, order, pad: float = float("inf")) -> torch.Tensor:
        """Inverse of ``deal``: `local` holds this rank's values for ``deal(order)``, in that order;
        returns the len(order) values indexed by candidate (order's entries) on every rank."""
        import numpy as np
        order = np.asarray(order)
        n = int(order.shape[0])
        if not self.enabled:
            out = torch.empty((n,), dtype=torch.float32, device=local.device)
            out[torch.from_numpy(order).to(local.device)] = local.to(torch.float32)
            return out
        per = self.per_rank(n)
        recv = self.gather(local, self.world * per, pad=pad) if per else local.new_empty((0,), dtype=torch.float32)
        src = np.empty((n,), dtype=np.int64)
        for r in range(self.world):
            idx = order[r::self.world]
            src[idx] = r * per + np.arange(idx.shape[0])
        return recv[torch.from_numpy(src).to(recv.device)]

    def gather_losses(self, local: torch.Tensor, n: int, flag: bool = False,
                      want_flag: bool = False) -> Tuple[torch.Tensor, bool]:
        """Losses plus the OR of a per-rank flag (kept for callers that stop per rank)."""
        full = self.gather(local, n)
        if not self.enabled:
            return full, bool(flag)
        any_flag = False
        if want_flag:
            f = self.gather(torch.tensor([1.0 if flag else 0.0], device=local.device), self.world, pad=0.0)
            any_flag = bool((f > 0).any().item())
        return full, any_flag

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.enabled:
            dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                           group=self.group)
        return t

    def broadcast_ids(self, ids: torch.Tensor) -> torch.Tensor:
        """Rank 0's (N, n_opt) candidate ids to everyone; N itself may differ per rank
        only if ranks have drifted, so it is sent first."""
        if not self.enabled:
            return ids
        shape = torch.tensor(list(ids.shape), dtype=torch.int64, device=ids.device)
        self.broadcast_(shape)
        n, w = (int(v) for v in shape.tolist())
        if tuple(ids.shape) != (n, w):
            ids = torch.empty((n, w), dtype=torch.int64, device=ids.device)
        return self.broadcast_(ids.contiguous())
