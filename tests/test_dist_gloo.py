"""world_size-2 (3 and 8) sharding of candidate scoring over gloo on CPU: slices are
contiguous and cover every candidate, ragged N pads with +inf, the gathered vector and
its argmin are identical on every rank and equal to the unsharded result."""

import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from bimodalattack_amd.dist import CandidateSharder
        sh = CandidateSharder()
        assert sh.enabled and sh.world == world and sh.rank == rank
        res = {}
        for n in (1, 2, 5, 64, 511, 512):
            truth = torch.arange(n, dtype=torch.float32).mul(0.37).sin() + 3.0     # every rank knows all
            lo, hi = sh.bounds(n)
            hits = (torch.arange(n) == n - 1).float()                              # one hit, on the last rank's slice
            before = sh.n_collectives
            full, got = sh.gather2(truth[lo:hi].clone(), hits[lo:hi].clone(), n)
            assert sh.n_collectives == before + 1                                   # losses and hits share ONE collective
            assert full.shape == (n,) and torch.equal(full, truth), (n, rank)
            assert torch.equal(got, hits)
            assert torch.equal(sh.gather(truth[lo:hi].clone(), n), truth)
            res[n] = int(full.argmin())
        # dealt partition (ragged scoring): every world-th entry of a cost-sorted order; the gather
        # puts each rank's values back under their candidate index
        for n in (world, 5, 64, 487):
            g = torch.Generator().manual_seed(n)
            truth = torch.rand(n, generator=g) + 1.0
            order = np.argsort(torch.rand(n, generator=g).numpy(), kind="stable")
            take = sh.deal(order)
            assert len(take) in (n // world, n // world + 1)
            mine = torch.from_numpy(np.ascontiguousarray(take))
            full = sh.gather_dealt(truth[mine].clone(), order)
            assert torch.equal(full, truth), (n, rank)
            full, neg = sh.gather_dealt(truth[mine].clone(), order, extra=-truth[mine])
            assert torch.equal(full, truth) and torch.equal(neg, -truth)
            # every candidate is dealt exactly once
            assert sorted(np.concatenate([sh.deal(order, r) for r in range(world)]).tolist()) == list(range(n))
            # the index uploaded before the forward (attack._score_candidates): composed with a duplicate map,
            # one gather hands back one value per ORIGINAL candidate
            if n:
                inverse = torch.randint(0, n, (n + 3,), generator=g).numpy()
                at = torch.from_numpy(sh.dealt_index(order, inverse))
                assert torch.equal(sh.gather_dealt(truth[mine].clone(), order, at=at), truth[torch.from_numpy(inverse)])
                a, b = sh.gather_dealt(truth[mine].clone(), order, extra=-truth[mine], at=at)
                assert torch.equal(a, truth[torch.from_numpy(inverse)]) and torch.equal(b, -truth[torch.from_numpy(inverse)])
        # a decision every rank must take alike (replay a captured graph with collectives inside, or run them eagerly):
        # true only when it is true on EVERY rank
        assert sh.all_ok(True, "cpu") is True
        assert sh.all_ok(rank != world - 1, "cpu") is False
        assert sh.all_ok(rank == 0, "cpu") is (world == 1)
        # rank 0's ids and image overwrite a drifted rank's, in one packed broadcast
        ids = torch.arange(12, dtype=torch.int64).view(4, 3) + (0 if rank == 0 else 100 * rank)
        img = torch.full((1, 3, 4, 4), float(rank) + 0.25).requires_grad_()
        before = sh.n_collectives
        sh.sync_state(ids, img)
        assert sh.n_collectives == before + 1
        assert torch.equal(ids, torch.arange(12).view(4, 3)) and img.requires_grad
        assert float(img.detach().sum()) == 0.25 * 48
        out.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_scoring_gloo(world):
    """World 8 is the driver's node: with n = 1, 2, 5 candidates (a decayed width the filter has thinned) every rank's
    slice is one candidate or EMPTY (`per` = 1, +inf padding only), with n = 8 dealt one each."""
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    results = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    first = results[0][1]
    assert all(r[1] == first for r in results)                                     # same argmin everywhere
    for n, idx in first.items():
        assert idx == int((torch.arange(n, dtype=torch.float32).mul(0.37).sin() + 3.0).argmin())


def test_bounds_cover_everything():
    from bimodalattack_amd.dist import CandidateSharder
    sh = CandidateSharder()
    assert (sh.world, sh.rank, sh.enabled) == (1, 0, False) and sh.bounds(7) == (0, 7)
    sh.world = 8
    for n in (0, 1, 7, 8, 9, 64, 500, 512):
        spans = [sh.bounds(n, r) for r in range(8)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(hi - lo for lo, hi in spans) <= sh.per_rank(n)


# ------------------------------------------------------------------ tensor-parallel gradient pass (round 3)
def _tp_worker(rank, world, port, kind, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.manual_seed(0)
        from bimodalattack_amd import synthetic as S
        from bimodalattack_amd.fused import FusedInference
        model, tok, proc, image = S.tiny_case(kind)
        lm = model.language_model if hasattr(model, "language_model") else model.model.language_model
        D = model.get_input_embeddings().weight.shape[1]
        g = torch.Generator().manual_seed(5)
        x0 = torch.randn((1, 23, D), generator=g) * 0.5
        fused = FusedInference(model)
        assert fused.tp_ok(world) and fused.layers

        def run(tp):
            x = x0.clone().requires_grad_()
            fused.tp = (rank, world, None) if tp else None
            try:
                with fused:
                    h = lm(inputs_embeds=x, use_cache=False).last_hidden_state
                loss = (h.float() * torch.linspace(-1, 1, h.numel()).view_as(h)).sum()
                (gx,) = torch.autograd.grad(loss, x)
            finally:
                fused.tp = None
            return h.detach(), gx

        h_ref, g_ref = run(False)
        h_tp, g_tp = run(True)
        assert not any("forward" in m.__dict__ for m in model.modules())          # every patch removed again
        out.put((rank, float((h_tp - h_ref).abs().max() / h_ref.abs().max()), float((g_tp - g_ref).abs().max() / g_ref.abs().max()),
                 h_tp.double().sum().item(), g_tp.double().sum().item()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("kind", ["llava", "gemma3"])
def test_tensor_parallel_gradient_pass_equals_replicated(kind):
    """The batch-1 pass cut over 2 ranks (EngineOptions.tp_gradient: q/k/v/gate/up by output rows = whole heads, o/down
    by input columns, f/g operators = two all-reduces per layer and direction) gives the hidden states and the input
    gradient of the replicated pass to fp32 summation-order noise, identically on both ranks.  gloo on CPU: the pass's
    plumbing (HuggingFace attention on the local heads, the fused layer forward carrying the f operator, patches
    removed afterwards) is device-independent."""
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, world, port, kind, out)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(out.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, eh, eg, sh_, sg in got:
        assert eh < 1e-5 and eg < 1e-4, (rank, eh, eg)
    assert got[0][3:] == got[1][3:]                                               # both ranks hold the same result
