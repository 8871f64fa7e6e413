"""Candidate sharding across the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" is RCCL over xGMI).  Weights
are replicated; the candidates of a step are independent given (sampled_ids,
image features), so rank r scores the contiguous slice [r*per, (r+1)*per) and the
only data-path exchange is one all-gather of `per` fp32 losses per rank (<= 2 KiB
in total at search_width 512) -- latency-bound, so a single one-shot collective,
never a ring of point-to-point hops.  N varies per step (filter, dynamic width):
slices are padded to `per` slots with +inf, which can never win the argmin.

Every rank runs gradient -> PGD -> sampling redundantly; rank 0's sampled ids and PGD
image then overwrite everybody's in ONE packed broadcast per step (``sync_state``), so ranks
cannot drift apart through last-bit differences in redundantly computed gradients.  With the
loss all-gather that makes two collectives per step (early_stop adds a two-float one; GCG-only steps with
early_plan one more, rank 0's random draws -- 8 KB, queued in front of the gradient pass).

The C ABI's collective, ``bma_allgather_f32(local, n_local, out, rank, world, comm, stream)``, takes
the HOST's ``ncclComm_t`` and calls ``ncclAllGather`` of the RCCL instance already in the process: it is for hosts that
created their communicator themselves (C, C++, Go).  This module does not go through it: torch.distributed owns the
RCCL communicator of a PyTorch job, its stream ordering and its error handling, and does not hand the handle out; a
second communicator inside libbma_hip.so would need its own bootstrap (unique-id exchange) for a <= 2 KiB payload.
The boundary for the exchange under PyTorch is therefore this module.
"""

from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.distributed as dist


class CandidateSharder:
    def __init__(self, group=None, enabled: Optional[bool] = None):
        """`enabled=False`: a sharder of one (world 1, rank 0, no collectives) whatever torch.distributed says."""
        self.enabled = dist.is_available() and dist.is_initialized() and enabled is not False
        self.group = group
        if self.enabled:
            self.world = dist.get_world_size(group)
            self.rank = dist.get_rank(group)
        else:
            self.world, self.rank = 1, 0
        self.enabled = self.enabled and self.world > 1
        self.n_collectives = 0          # data-path collectives issued (tests and bench.py read it)

    def backend(self) -> Optional[str]:
        """"nccl" (= RCCL), "gloo", ... of the group the candidates are sharded over; None without a process group."""
        if not (dist.is_available() and dist.is_initialized()):
            return None
        return dist.get_backend(self.group)

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
        return self.gather2(local, None, n, pad)[0]

    def gather2(self, local: torch.Tensor, extra: Optional[torch.Tensor], n: int, pad: float = float("inf"),
                pad_extra: float = 0.0) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        """``gather`` of two per-candidate vectors (losses and early-stop hits) in ONE collective:
        each rank sends [per losses | per hits]."""
        if not self.enabled:
            return local, extra
        per = self.per_rank(n)
        k = 1 if extra is None else 2
        send = torch.empty((k, per), dtype=torch.float32, device=local.device)
        send[0].fill_(pad)
        send[0, : local.numel()] = local.to(torch.float32)
        if extra is not None:
            send[1].fill_(pad_extra)
            send[1, : extra.numel()] = extra.to(torch.float32)
        self.n_collectives += 1
        if dist.get_backend(self.group) == "nccl":
            recv = torch.empty((self.world, k, per), dtype=torch.float32, device=local.device)
            dist.all_gather_into_tensor(recv.view(-1), send.view(-1), group=self.group)
        else:
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(parts, send, group=self.group)
            recv = torch.stack(parts)
        a = recv[:, 0, :].reshape(-1)[:n].contiguous()
        b = recv[:, 1, :].reshape(-1)[:n].contiguous() if extra is not None else None
        return a, b

    # -- dealt partition (ragged scoring) -----------------------------------------------
    def deal(self, order, rank: Optional[int] = None):
        """This rank's share of `order` (any sequence of candidate indices): every world-th entry.
        With `order` sorted by a per-candidate cost, every rank gets the same mix of costs."""
        r = self.rank if rank is None else rank
        return order[r::self.world]

    def dealt_index(self, order, inverse=None, world: Optional[int] = None):
        """Host index vector for ``gather_dealt(..., at=)``: position of every candidate's value in the gathered
        buffer (rank r's k-th dealt entry sits at r*per + k), composed with `inverse` (candidate -> distinct
        candidate) when given.  Built -- and uploaded by the caller -- BEFORE the forward is enqueued: a host-to-
        device copy from pageable memory issued behind it blocks the host until the stream has drained (ROCm),
        which put the whole retokenisation filter behind the forward instead of beside it."""
        import numpy as np
        order = np.asarray(order)
        n = int(order.shape[0])
        w = self.world if world is None else world
        per = -(-n // w)
        src = np.empty((n,), dtype=np.int64)
        for r in range(w):
            idx = order[r::w]
            src[idx] = r * per + np.arange(idx.shape[0])
        return src if inverse is None else src[np.asarray(inverse)]

    def gather_dealt(self, local: torch.Tensor, order, pad: float = float("inf"),
                     extra: Optional[torch.Tensor] = None, at: Optional[torch.Tensor] = None):
        """Inverse of ``deal``: `local` holds this rank's values for ``deal(order)``, in that order;
        returns the len(order) values indexed by candidate (order's entries) on every rank (and, with
        `extra`, a second vector gathered in the same collective: a pair is returned).  `at`: the device copy
        of ``dealt_index(order[, inverse])`` uploaded earlier (then len(at) values come back)."""
        import numpy as np
        order = np.asarray(order)
        n = int(order.shape[0])
        if not self.enabled:
            if at is None:
                at = torch.from_numpy(order).to(local.device)
                outs = []
                for v in (local, extra):
                    if v is None:
                        outs.append(None)
                        continue
                    o = torch.empty((n,), dtype=torch.float32, device=local.device)
                    o[at] = v.to(torch.float32)
                    outs.append(o)
                return outs[0] if extra is None else tuple(outs)
            outs = [None if v is None else v.to(torch.float32)[at] for v in (local, extra)]   # world 1: per = n
            return outs[0] if extra is None else tuple(outs)
        per = self.per_rank(n)
        if per:
            recv, recv2 = self.gather2(local, extra, self.world * per, pad=pad)
        else:
            recv = local.new_empty((0,), dtype=torch.float32)
            recv2 = None if extra is None else recv
        if at is None:
            at = torch.from_numpy(self.dealt_index(order)).to(recv.device)
        return recv[at] if extra is None else (recv[at], recv2[at])

    def broadcast_(self, t: torch.Tensor, src: int = 0) -> torch.Tensor:
        if self.enabled:
            self.n_collectives += 1
            dist.broadcast(t, src=dist.get_global_rank(self.group, src) if self.group is not None else src,
                           group=self.group)
        return t

    def all_ok(self, ok: bool, device) -> bool:
        """True when `ok` holds on EVERY rank (one all-reduce(MIN) of a flag; every rank must call it).  Decisions that
        change which collectives a rank issues later -- replaying a captured graph with its all-reduces inside, or running
        them eagerly -- are taken with it, never on one rank alone."""
        if not self.enabled:
            return bool(ok)
        flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float32, device=device)
        self.n_collectives += 1
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() > 0.5)

    def sync_state(self, *tensors: torch.Tensor) -> None:
        """Rank 0's values of `tensors` (any dtypes, fixed shapes known to every rank: the sampled
        ids, the PGD image) overwrite everybody's, in ONE broadcast of their packed bytes."""
        if not self.enabled:
            return
        flat = [t.detach().contiguous().view(-1).view(torch.uint8) for t in tensors]
        buf = flat[0] if len(flat) == 1 else torch.cat(flat)
        self.broadcast_(buf)
        if self.rank != 0:
            at = 0
            with torch.no_grad():
                for t, f in zip(tensors, flat):
                    t.detach().copy_(buf[at:at + f.numel()].view(t.dtype).view(t.shape))
                    at += f.numel()
