"""Parity of the HIP kernels (called through the C ABI) with the oracle and with the
golden vectors captured from the reference.  Run on the MI355X box: pytest -m gpu.

Bars: indices / ids / copied bytes / PGD pixels bit-exact; fp32 losses <= 1e-4 rel
(BASELINE.json north_star), tolerances written at each assert.
"""

import os

import numpy as np
import pytest
import torch

from oracle import kernels as K

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    from bimodalattack_amd import native, ops as _ops
    native.check_single_hip_runtime()
    return _ops


def load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def dev(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    return t if dtype is None else t.to(dtype)


def bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


# ------------------------------------------------------------------ a5 L-inf step
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_linf_golden(ops, golden_dir, case):
    z = load(golden_dir, "g3_pgd.npz")
    eps, alpha = (float(v) for v in z[f"{case}_eps_alpha"])
    y = ops.linf_step(dev(z[f"{case}_x"]), dev(z[f"{case}_g"]), dev(z[f"{case}_x0"]), eps, alpha)
    assert np.array_equal(bits(y.cpu().numpy()), bits(z[f"{case}_y"]))


def test_linf_chain_golden_inplace(ops, golden_dir):
    z = load(golden_dir, "g3_pgd.npz")
    eps, alpha = (float(v) for v in z["chain_eps_alpha"])
    x0 = dev(z["chain_x0"])
    x = x0.clone()
    for g, want in zip(z["chain_g"], z["chain_x"]):
        ops.linf_step(x, dev(g), x0, eps, alpha, out=x)          # out aliases x
        assert np.array_equal(bits(x.cpu().numpy()), bits(want))


@pytest.mark.parametrize("n", [1, 3, 5, 1023, 4099, 3 * 336 * 336, 3 * 896 * 896])
def test_linf_sizes_vs_oracle(ops, n):
    rs = np.random.RandomState(n % 9973)
    x0 = rs.uniform(0, 1, n).astype(np.float32)
    x = np.clip(x0 + rs.uniform(-0.3, 0.3, n), 0, 1).astype(np.float32)
    g = rs.standard_normal(n).astype(np.float32)
    g[::5] = 0
    g[1::17] = np.nan
    g[2::19] = -np.inf
    eps, alpha = 64 / 255, 4 / 255
    want = K.linf_step(x, g, x0, eps, alpha)
    got = ops.linf_step(dev(x), dev(g), dev(x0), eps, alpha).cpu().numpy()
    assert np.array_equal(bits(got), bits(want))
    assert np.abs(got - x0).max() <= np.float32(eps) + 1e-7 and got.min() >= 0 and got.max() <= 1
    # idempotent projection: a zero gradient leaves a feasible point untouched
    again = ops.linf_step(dev(got), torch.zeros(n, device=DEV), dev(x0), eps, alpha).cpu().numpy()
    assert np.array_equal(bits(again), bits(got))


def test_linf_unaligned_and_empty(ops):
    rs = np.random.RandomState(0)
    n = 1001
    buf = [dev(rs.uniform(0, 1, n + 1).astype(np.float32)) for _ in range(3)]
    x, g, x0 = (b[1:] for b in buf)                     # 4-byte aligned only -> scalar path
    g = g - 0.5
    want = K.linf_step(x.cpu().numpy(), g.cpu().numpy(), x0.cpu().numpy(), 0.1, 0.5)
    out = torch.empty(n + 1, device=DEV)[1:]
    from bimodalattack_amd.native import lib, check
    check("bma_linf_step", lib.bma_linf_step(x.data_ptr(), g.contiguous().data_ptr() if False else g.data_ptr(),
                                             x0.data_ptr(), n, 0.1, float(0.5 * 0.1), out.data_ptr(),
                                             torch.cuda.current_stream().cuda_stream))
    assert np.array_equal(bits(out.cpu().numpy()), bits(want))
    e = torch.empty(0, device=DEV)
    assert ops.linf_step(e, e, e, 0.1, 0.1).numel() == 0


# ------------------------------------------------------------------ a2 CE target
def test_ce_golden(ops, golden_dir):
    z = load(golden_dir, "g2_ce.npz")
    logits, T = z["logits"], z["target"].shape[1]
    full = dev(logits)
    sl = full[:, logits.shape[1] - T - 1:-1]                    # strided view, as the engine passes it
    loss, match, dlog, rows = ops.ce_target(sl, dev(z["target"][0]), want_match=True, want_dlogits=True)
    np.testing.assert_allclose(loss.cpu().numpy(), z["loss"], rtol=1e-4)
    np.testing.assert_allclose(rows.cpu().numpy(), z["row_loss"], rtol=1e-4)
    assert not match.any()
    np.testing.assert_allclose(dlog[0].cpu().numpy(), z["dlogits0"], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(loss[0].item(), float(z["mean0"]), rtol=1e-4)
    loss_hit, match_hit, _, _ = ops.ce_target(sl, dev(z["target_hit"][0]), want_match=True)
    np.testing.assert_allclose(loss_hit.cpu().numpy(), z["loss_hit"], rtol=1e-4)
    assert match_hit.cpu().tolist() == [int(i == 5) for i in range(8)]
    # the oracle itself (float64) is a tighter reference than the fp32 golden
    want, _ = K.ce_target(logits[:, logits.shape[1] - T - 1:-1], z["target"][0])
    np.testing.assert_allclose(loss.cpu().numpy(), want, rtol=2e-6)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V", [1000, 1003, 32064, 7])
def test_ce_dtypes_vs_oracle(ops, dtype, V):
    rs = np.random.RandomState(V)
    B, T = 5, 4
    x = (rs.standard_normal((B, T + 2, V)) * 3).astype(np.float32)
    xt = dev(x, dtype)
    lab = rs.randint(0, V, size=T).astype(np.int64)
    sl = xt[:, 1:1 + T]
    xr = sl.float().cpu().numpy()                                # the rounded inputs the kernel saw
    want, wmatch = K.ce_target(xr, lab)
    loss, match, dlog, rows = ops.ce_target(sl, dev(lab), want_match=True, want_dlogits=True, grad_scale=0.5)
    np.testing.assert_allclose(loss.cpu().numpy(), want, rtol=1e-5)
    np.testing.assert_allclose(rows.cpu().numpy(), K.ce_rows(xr, lab), rtol=1e-5, atol=1e-6)
    assert match.cpu().numpy().astype(bool).tolist() == wmatch.tolist()
    tol = dict(rtol=1e-5, atol=1e-8) if dtype == torch.float32 else dict(rtol=1e-2, atol=1e-6)
    for b in range(B):
        np.testing.assert_allclose(dlog[b].float().cpu().numpy(), K.ce_target_grad(xr[b], lab, 0.5), **tol)


def test_ce_argmax_first_maximum_and_extremes(ops):
    V, T = 520, 3
    x = np.full((2, T, V), -5.0, np.float32)
    x[0, :, 100] = 7.0
    x[0, :, 300] = 7.0            # tie: argmax is the FIRST maximum (index 100)
    x[1, :, 300] = 7.0
    x[1, 2, 10] = 80.0            # large logit: no overflow in the online sum
    loss, match, _, _ = ops.ce_target(dev(x), dev(np.array([300, 300, 300])), want_match=True)
    assert match.cpu().tolist() == [0, 0]
    loss2, match2, _, _ = ops.ce_target(dev(x), dev(np.array([100, 100, 100])), want_match=True)
    assert match2.cpu().tolist() == [1, 0]
    want, _ = K.ce_target(x, np.array([300, 300, 300]))
    np.testing.assert_allclose(loss.cpu().numpy(), want, rtol=1e-5)
    assert torch.isfinite(loss).all()


def test_ce_autograd_function(ops):
    rs = np.random.RandomState(1)
    T, V = 6, 264
    x = dev(rs.standard_normal((T, V)).astype(np.float32)).requires_grad_()
    lab = dev(rs.randint(0, V, T))
    loss = ops.TargetCrossEntropy.apply(x, lab)
    (g,) = torch.autograd.grad(loss * 2.0, x)
    xr = x.detach().cpu().numpy()
    np.testing.assert_allclose(loss.item(), K.ce_rows(xr, lab.cpu().numpy()).mean(), rtol=1e-5)
    np.testing.assert_allclose(g.cpu().numpy(), K.ce_target_grad(xr, lab.cpu().numpy(), 2.0), rtol=1e-4, atol=1e-8)


def test_ce_baseline_size(ops):
    """B=512, T=20, V=32064, bf16 (657 MB): a sample of candidates against the
    oracle; size-independent properties for the rest."""
    B, T, V = 512, 20, 32064
    g = torch.Generator(device=DEV).manual_seed(0)
    x = (torch.randn((B, T + 1, V), generator=g, device=DEV, dtype=torch.float32) * 2).to(torch.bfloat16)
    lab = torch.randint(0, V, (T,), generator=g, device=DEV)
    sl = x[:, :T]
    loss, match, _, rows = ops.ce_target(sl, lab, want_match=True)
    pick = [0, 1, 255, 256, 511]
    want, _ = K.ce_target(sl[pick].float().cpu().numpy(), lab.cpu().numpy())
    np.testing.assert_allclose(loss[pick].cpu().numpy(), want, rtol=1e-5)
    assert torch.isfinite(loss).all() and not match.any()
    # permutation of candidates permutes the losses, bit for bit
    perm = torch.randperm(B, generator=g, device=DEV)
    loss_p, _, _, _ = ops.ce_target(sl[perm].contiguous(), lab)
    assert torch.equal(loss_p, loss[perm])
    # mean of rows in fixed order
    np.testing.assert_allclose(loss.cpu().numpy(), rows.cpu().numpy().astype(np.float64).mean(1), rtol=1e-6)
    # against torch's own fp32 log-softmax on the device (a second opinion, not the oracle)
    ref = torch.nn.functional.cross_entropy(sl.float().reshape(-1, V), lab.repeat(B), reduction="none").view(B, T).mean(1)
    np.testing.assert_allclose(loss.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5)


def test_ce_gemma_size(ops):
    """Gemma-3-4b vocabulary (BASELINE configs[4]): B=64, T=20, V=262208, bf16 (671 MB) -- 4 KiB-unaligned row
    length, 8x the LLaVA row; a sample of candidates against the oracle, the early-stop argmax, properties."""
    B, T, V = 64, 20, 262208
    g = torch.Generator(device=DEV).manual_seed(1)
    x = (torch.randn((B, T, V), generator=g, device=DEV, dtype=torch.float32) * 2).to(torch.bfloat16)
    lab = torch.randint(0, V, (T,), generator=g, device=DEV)
    x[5, torch.arange(T, device=DEV), lab] = 40.0                  # candidate 5 predicts every target token
    x[6, torch.arange(T - 1, device=DEV), lab[:-1]] = 40.0         # candidate 6 misses the last one
    loss, match, _, rows = ops.ce_target(x, lab, want_match=True)
    pick = [0, 5, 6, 63]
    want, wmatch = K.ce_target(x[pick].float().cpu().numpy(), lab.cpu().numpy())
    # candidate 5's loss is 8e-12 in float64: below fp32 resolution next to a logit of 40, hence the atol
    np.testing.assert_allclose(loss[pick].cpu().numpy(), want, rtol=1e-5, atol=1e-6)
    assert match.cpu().tolist() == [int(i == 5) for i in range(B)] and wmatch.tolist() == [False, True, False, False]
    perm = torch.randperm(B, generator=g, device=DEV)
    assert torch.equal(ops.ce_target(x[perm].contiguous(), lab)[0], loss[perm])
    ref = torch.nn.functional.cross_entropy(x.float().reshape(-1, V), lab.repeat(B), reduction="none").view(B, T).mean(1)
    np.testing.assert_allclose(loss.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-6)
    # the gradient-pass form at this width: dlogits of one candidate against the oracle
    _, _, dlog, _ = ops.ce_target(x[:1], lab, want_dlogits=True)
    np.testing.assert_allclose(dlog[0].float().cpu().numpy(), K.ce_target_grad(x[0].float().cpu().numpy(), lab.cpu().numpy()),
                               rtol=1e-2, atol=1e-6)


# ------------------------------------------------------------------ a3 sampling
@pytest.mark.parametrize("case", ["a", "b", "c", "d"])
def test_sampling_golden(ops, golden_dir, case):
    z = load(golden_dir, "g1_sampling.npz")
    n_opt, V, sw, topk, n_rep = (int(v) for v in z[f"{case}_meta"])
    na = z[f"{case}_not_allowed"]
    mask = ops.build_mask_bits(torch.from_numpy(na), V, DEV) if na.size else None
    tk = ops.mask_topk(dev(z[f"{case}_grad"]), mask, topk)
    assert np.array_equal(tk.cpu().numpy(), z[f"{case}_topk_ids"])
    pos = ops.rand_positions(dev(z[f"{case}_rand"]), n_rep)
    assert np.array_equal(pos.cpu().numpy(), z[f"{case}_pos"])
    new = ops.sample_scatter(dev(z[f"{case}_ids"]), tk, pos, dev(z[f"{case}_rank"]))
    assert np.array_equal(new.cpu().numpy(), z[f"{case}_new_ids"])


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16, torch.float16])
@pytest.mark.parametrize("V,k,rows", [(32064, 256, 19), (32003, 256, 3), (262208, 256, 19), (300, 300, 2),
                                      (5000, 2048, 2), (64, 1, 5)])
def test_topk_vs_oracle_with_ties(ops, dtype, V, k, rows):
    """bf16 gradients over a 32k vocabulary have ~2k distinct values: heavy ties at
    the k-th boundary.  Policy (value asc, id asc) must hold exactly."""
    rs = np.random.RandomState(V + k)
    g = (rs.standard_normal((rows, V)) * 0.01).astype(np.float32)
    g[:, ::97] = 0.0
    g[:, 5::101] = -0.0
    if V > 1000:
        g[0, 11] = np.nan
        g[1, 13] = -np.inf
        g[0, 17] = np.inf
    gt = dev(g, dtype)
    gr = gt.float().cpu().numpy()
    na = np.unique(rs.randint(0, V, size=max(1, V // 50)))
    if k >= V:
        na = np.zeros(0, np.int64)
    mask = ops.build_mask_bits(torch.from_numpy(na), V, DEV)
    got = ops.mask_topk(gt, mask, k).cpu().numpy()
    want = K.mask_topk(gr, na, k)
    assert np.array_equal(got, want)
    # no forbidden token is ever selected; rows hold k distinct ids
    assert not np.isin(got, na).any()
    assert all(len(set(r.tolist())) == k for r in got)
    # unmasked variant
    assert np.array_equal(ops.mask_topk(gt, None, k).cpu().numpy(), K.mask_topk(gr, None, k))


def test_topk_strided_rows_and_grad_untouched(ops):
    rs = np.random.RandomState(5)
    big = dev(rs.standard_normal((4, 2, 1024)).astype(np.float32))
    view = big[:, 1]                                            # row stride 2048 elements
    before = view.clone()
    na = np.arange(0, 1024, 3)
    got = ops.mask_topk(view, ops.build_mask_bits(torch.from_numpy(na), 1024, DEV), 32).cpu().numpy()
    assert np.array_equal(got, K.mask_topk(before.cpu().numpy(), na, 32))
    assert torch.equal(view, before)


def test_rand_positions_and_scatter_vs_oracle(ops):
    rs = np.random.RandomState(9)
    for B, n_opt, n_rep, k in [(512, 19, 1, 256), (128, 19, 3, 64), (1, 1, 1, 4), (77, 64, 64, 8),
                            (33, 100, 2, 16), (9, 257, 257, 4)]:
        rnd = rs.uniform(size=(B, n_opt)).astype(np.float32)
        rnd[0, :] = 0.5 if n_opt > 1 else rnd[0, :]              # ties: lowest position first
        pos = ops.rand_positions(dev(rnd), n_rep)
        assert np.array_equal(pos.cpu().numpy(), K.rand_positions(rnd, n_rep))
        ids = rs.randint(0, 32000, n_opt).astype(np.int64)
        tk = rs.randint(0, 32000, (n_opt, k)).astype(np.int64)
        rank = rs.randint(0, k, (B, n_rep)).astype(np.int64)
        new = ops.sample_scatter(dev(ids), dev(tk), pos, dev(rank))
        assert np.array_equal(new.cpu().numpy(), K.sample_scatter(ids, tk, pos.cpu().numpy(), rank))
        # every candidate differs from the parent in at most n_rep positions
        assert ((new.cpu().numpy() != ids[None]).sum(1) <= n_rep).all()


# ------------------------------------------------------------------ a7 splice
COMBOS = {
    "pgd_single": dict(mode="pgd", single=True), "gcg_single": dict(mode="gcg", single=True),
    "gcg_nojoint": dict(mode="gcg", no_joint_eval=True), "gcg_notarget": dict(mode="gcg", no_target=True),
    "gcgpgd_single": dict(mode="gcg_pgd", single=True), "gcgpgd_notarget": dict(mode="gcg_pgd", no_target=True),
    "gcgpgd_full": dict(mode="gcg_pgd"),
}


@pytest.mark.parametrize("mt", ["llava", "gemma3"])
def test_splice_golden(ops, golden_dir, mt):
    from bimodalattack_amd.layout import segment_order
    z = load(golden_dir, "g4_splice.npz")
    seg = {k[4:]: dev(z[k]) for k in z.files if k.startswith("seg_")}
    seg["image"] = dev(z["image"])
    table, ids = dev(z["table"]), dev(z["ids"])
    scale = float(np.float32(8 ** 0.5)) if mt == "gemma3" else 1.0
    for name, kw in COMBOS.items():
        order = segment_order(kw["mode"], mt, **{k: v for k, v in kw.items() if k != "mode"})
        segs = [("gather", None) if n == "optim" else ("shared", seg[n]) for n in order]
        y = ops.splice(segs, ids.shape[0], table, ids, scale)
        assert np.array_equal(bits(y.cpu().numpy()), bits(z[f"{mt}_{name}"])), name
    order = segment_order("gcg_pgd", mt)
    segs = [("gather", None) if n == "optim" else ("shared", seg[n]) for n in order]
    y1 = ops.splice(segs, 1, table, ids[2:3].contiguous(), scale)
    assert np.array_equal(bits(y1.cpu().numpy()), bits(z[f"{mt}_gcgpgd_one"]))


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16, torch.float32])
@pytest.mark.parametrize("B,lens,D", [(512, (21, 19, 6, 19), 4096), (64, (5, 576, 18, 19, 6, 20), 4096),
                                      (3, (2, 19, 1), 8), (33, (0, 19, 0, 7), 2560)])
def test_splice_sizes_bitwise(ops, dtype, B, lens, D):
    """Copy semantics: every output byte equals its source byte (checked with torch
    indexing on the device as the independent reference)."""
    g = torch.Generator(device=DEV).manual_seed(B)
    V, n_opt = 1000, 19
    table = torch.randn((V, D), generator=g, device=DEV).to(dtype)
    ids = torch.randint(0, V, (B, n_opt), generator=g, device=DEV)
    segs, parts = [], []
    for i, L in enumerate(lens):
        if L == n_opt and not any(k == "gather" for k, _ in segs):
            segs.append(("gather", None))
            parts.append(table[ids])
        elif i % 2 == 0:
            t = torch.randn((1, L, D), generator=g, device=DEV).to(dtype)
            segs.append(("shared", t))
            parts.append(t.expand(B, L, D))
        else:
            t = torch.randn((B, L, D), generator=g, device=DEV).to(dtype)
            segs.append(("percand", t))
            parts.append(t)
    y = ops.splice(segs, B, table, ids)
    want = torch.cat(parts, dim=1)
    assert y.shape == want.shape and torch.equal(y.view(torch.uint8), want.contiguous().view(torch.uint8))


def test_splice_gemma_scale_matches_hf_embedding(ops):
    from transformers.models.gemma3.modeling_gemma3 import Gemma3TextScaledWordEmbedding
    V, D, B, n = 512, 2560, 16, 19
    for dtype in (torch.bfloat16, torch.float32):
        emb = Gemma3TextScaledWordEmbedding(V, D, padding_idx=0, embed_scale=D ** 0.5).to(DEV, dtype)
        with torch.no_grad():
            emb.weight.copy_(torch.randn(V, D, device=DEV) * 0.02)
        ids = torch.randint(0, V, (B, n), device=DEV)
        with torch.no_grad():
            want = emb(ids)
        scale = float(emb.embed_scale.to(dtype).float())
        y = ops.splice([("gather", None)], B, emb.weight.detach(), ids, scale)
        assert torch.equal(y.view(torch.uint8), want.contiguous().view(torch.uint8))


def test_splice_clamps_bad_ids(ops):
    table = torch.randn(10, 8, device=DEV)
    ids = torch.tensor([[-3, 99]], device=DEV)
    y = ops.splice([("gather", None)], 1, table, ids)
    assert torch.equal(y[0, 0], table[0]) and torch.equal(y[0, 1], table[9])


# ------------------------------------------------------------------ bma_gemm_nt (round 3)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_nt_matches_fp32_reference(dtype, monkeypatch):
    """y = x W^T on the hand-written skinny kernel against the product in fp32 (one rounding of the fp32 sum to the
    16-bit type: at most 1 ulp from the rounded fp32 reference): the gradient pass's shapes (65 and 44 rows; forward
    and transposed-copy input-gradient shapes, 3- to 16-way split-K), one row, N off the 128-row slab and off 16,
    row counts on and off the 64- and 96-row tiles and several row tiles per slab; bitwise equal over repeated launches
    (the split-K reducer adds in split order whichever workgroup arrives last) and the tile tickets are zero again
    after every launch.  ops.gemm_nt_ok routes only K >= 3N (where the kernel beats the library): lifted here."""
    from bimodalattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(123)
    monkeypatch.setattr(ops, "GEMM_NT_MIN_K_OVER_N", 0.0)
    shapes = [(65, 4096, 4096), (65, 22016, 4096), (44, 4096, 22016), (65, 12288, 4096), (45, 4096, 11008), (1, 128, 64),
              (96, 132, 128), (17, 32064, 4096), (64, 4096, 11008), (33, 260, 192), (96, 1024, 4096), (7, 8, 64), (65, 512, 64), (80, 22016, 128), (300, 4096, 4096)]
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for M, N, K in shapes:
        x = (torch.randn((M, K), generator=g, device=DEV)).to(dtype)
        w = (torch.randn((N, K), generator=g, device=DEV) * 0.05).to(dtype)
        assert ops.gemm_nt_ok(x, w) or M > ops.GEMM_NT_MAX_ROWS       # (more rows: several row tiles per slab; not routed by default)
        y = ops.gemm_nt(x, w)
        ref = x.float() @ w.float().t()
        err = (y.float() - ref).abs()
        tol = 2.02 * eps * ref.abs() + 1e-30 + (2.0 ** -24 if dtype == torch.float16 else 0.0)   # (1 ulp: a last-bit difference of the two fp32 sums can flip the rounding)
        assert y.shape == (M, N) and bool((err <= tol + 3e-3 * ref.abs().max() * eps).all()), (M, N, K, float(err.max()))
        for _ in range(4):
            assert torch.equal(ops.gemm_nt(x, w), y)
        ws, cnt = ops.gemm_workspace(torch.device(DEV))
        assert int(cnt.sum()) == 0
        # against the library on the same operands: both are one rounding of (almost) the same fp32 sum
        lib_y = torch.nn.functional.linear(x, w)
        assert float((y.float() - lib_y.float()).abs().max()) <= 2.5 * eps * float(ref.abs().max())
    # a 3-D activation, as the decoder hands it over
    x = torch.randn((1, 65, 4096), generator=g, device=DEV).to(dtype)
    w = (torch.randn((4096, 4096), generator=g, device=DEV) * 0.05).to(dtype)
    assert ops.linear_b1(x, w).shape == (1, 65, 4096)
    assert torch.equal(ops.linear_b1(x, w)[0], ops.gemm_nt(x[0], w))
    # shapes it does not take go to the library
    assert not ops.gemm_nt_ok(torch.zeros((97, 4096), device=DEV, dtype=dtype), w)
    assert not ops.gemm_nt_ok(torch.zeros((8, 100), device=DEV, dtype=dtype), torch.zeros((16, 100), device=DEV, dtype=dtype))
    assert not ops.gemm_nt_ok(torch.zeros((8, 4096), device=DEV), w.float())
    monkeypatch.undo()
    assert not ops.gemm_nt_ok(x, w) and ops.gemm_nt_ok(torch.zeros((65, 22016), device=DEV, dtype=dtype),
                                                     torch.zeros((4096, 22016), device=DEV, dtype=dtype))


def test_gemm_nt_split_k_handoff_under_alternating_operands():
    """the split-K hand-off has no release / acquire fence -- partials leave as write-through (sc1)
    stores, the reducer reads them with sc1 loads -- which is sound under gfx950's cache behaviour, not under the HIP
    memory model; and a test that re-runs IDENTICAL operands would pass on a stale partial of the previous launch.  So:
    the same shape and plan with DIFFERENT operands on consecutive launches (two activations x two weights, alternating,
    the workspace reused each time), splits 2..16, with the XCD grouping OFF (producer and reducer on different XCD L2s)
    and on, thousands of launches; every result must equal, bit for bit, what the FENCED variant of the kernel (flags
    bit 3: agent-scope release behind the stores, acquire in front of the reducer's loads) gives for those operands --
    and a stale partial of the other operand pair could not."""
    import ctypes
    from bimodalattack_amd import ops
    from bimodalattack_amd.native import lib
    dev = torch.device(DEV)
    g = torch.Generator(device=DEV).manual_seed(11)
    M, N, K = 65, 4096, 8192
    xs = [torch.randn((M, K), generator=g, device=DEV).to(torch.bfloat16) for _ in range(2)]
    ws = [(torch.randn((N, K), generator=g, device=DEV) * 0.05).to(torch.bfloat16) for _ in range(2)]
    plan = (ctypes.c_int * 8)()
    launches = 0
    try:
        for S in (2, 3, 4, 7, 8, 16):
            want = {}
            lib.bma_gemm_nt_set_plan(2, 128, S, 2 | 8)                     # fenced, no XCD grouping: the reference bits
            assert lib.bma_gemm_nt_plan(M, N, K, plan) == 0 and plan[5] == S and plan[6] == 0
            for i in range(2):
                for j in range(2):
                    want[i, j] = ops.gemm_nt(xs[i], ws[j]).clone()
                    ref = xs[i].float() @ ws[j].float().t()
                    assert float((want[i, j].float() - ref).abs().max()) <= 2.0 ** -7 * float(ref.abs().max())
            assert not torch.equal(want[0, 0], want[1, 1]) and not torch.equal(want[0, 0], want[0, 1])
            for flags in (2, 3):                                            # unfenced: XCD grouping off, then on
                lib.bma_gemm_nt_set_plan(2, 128, S, flags)
                lib.bma_gemm_nt_plan(M, N, K, plan)
                assert plan[5] == S and plan[6] == (1 if (flags & 1) and (32 * S) % 8 == 0 else 0)
                outs = []
                n_iter = 400
                for it in range(n_iter):
                    i, j = it & 1, (it >> 1) & 1
                    outs.append(((i, j), ops.gemm_nt(xs[i], ws[j])))
                    if len(outs) == 50:                                     # compare in batches: the launches stay back to back
                        for key, y in outs:
                            assert torch.equal(y, want[key]), (S, flags, key)
                        outs = []
                launches += n_iter
                _, cnt = ops.gemm_workspace(dev)
                assert int(cnt.sum()) == 0
    finally:
        lib.bma_gemm_nt_set_plan(0, 0, 0, -1)
    assert launches >= 4000


def test_gemm_nt_next_weight_prefetch_changes_nothing():
    """bma_gemm_nt_next: a launch told about the next product of its chain -- the workgroups a split
    launch lets go early, or prefetch-only workgroups on the CUs an unsplit one leaves idle, load the first stages of the
    next weight -- returns exactly what bma_gemm_nt returns, for every (this, next) pair of the pass's shapes (and odd ones:
    a next weight smaller than one slab, fewer stages than the prefetch depth).  The hint reads rows [0, next_N) x columns
    [0, next_K) of the next weight only, by construction (csrc/gemm_nt.hip prefetch_next: rows clipped to the slab's real
    rows, stages to the split's own range); the weights here are views that start inside a larger block, so that a wrong
    row pitch or base would at least read foreign data -- which cannot change a result either: the loads are discarded."""
    from bimodalattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(5)
    shapes = [(12288, 4096), (22016, 4096), (4096, 11008), (11008, 4096), (4096, 22016), (4096, 12288), (132, 128), (8, 64)]
    for M in (65, 44, 1):
        mats = {}
        for N, K in shapes:
            # carve the weight out of the END of a bigger block: rows past N do not belong to it
            block = (torch.randn(((N + 7) * K,), generator=g, device=DEV) * 0.05).to(torch.bfloat16)
            mats[N, K] = block[7 * K:].view(N, K)
        for (N, K), w in mats.items():
            x = torch.randn((M, K), generator=g, device=DEV).to(torch.bfloat16)
            plain = ops.gemm_nt(x, w)
            for (N2, K2), w2 in mats.items():
                assert torch.equal(ops.gemm_nt(x, w, next_w=w2), plain), (M, N, K, N2, K2)
    torch.cuda.synchronize()
    # the registry form: a chain registered once, looked up by the weight's address
    ws = [mats[12288, 4096], mats[22016, 4096], mats[4096, 11008]]
    assert ops.gemm_nt_chain([ws[0], None, ws[1], ws[2]]) == 2
    try:
        x = torch.randn((65, 4096), generator=g, device=DEV).to(torch.bfloat16)
        assert ops._next_weight(ws[0]) is ws[1] and ops._next_weight(ws[2]) is None
        assert torch.equal(ops.gemm_nt(x, ws[0]), ops.gemm_nt(x, ws[0], next_w=ws[1]))
    finally:
        ops.gemm_nt_chain_clear()


# ------------------------------------------------------------------ bma_causal_attention (round 4)
def _causal_reference(q, k, v, scale, causal=True):
    """float64 attention of q (Lq,H,D) -- the LAST Lq positions -- against k, v (Lk,H,D)."""
    Lq, Lk = q.shape[0], k.shape[0]
    s = torch.einsum("qhd,khd->hqk", q, k) * scale
    if causal:
        hidden = torch.arange(Lk, device=q.device)[None, :] > (Lk - Lq + torch.arange(Lq, device=q.device))[:, None]
        s = s.masked_fill(hidden[None], float("-inf"))
    return torch.einsum("hqk,khd->qhd", s.softmax(-1), v)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_tower_attention_64_wide_heads_every_key_visible(dtype):
    """The same kernels at head width 64 and with every key visible (causal = 0): CLIP's 577 tokens x 16 heads, lengths on and
    off the block and chunk sizes, a causal 64-wide case; and at SigLIP's head width 72 (round 5: the real width in memory,
    96-wide images with zero columns in LDS) -- Gemma-3's 4096 tokens x 16 heads, short and odd lengths, a causal case;
    forward and backward against float64 on the same 16-bit operands, the gradients also against the library's pair;
    bit-equal over repeated launches."""
    from bimodalattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(14)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    # (72-wide heads of 1024 tokens and more run the two-tiles-per-wave forward: 128 rows per workgroup -- lengths on and off
    # that block, causal and not)
    for H, L, Dh, causal in ((16, 577, 64, False), (2, 64, 64, False), (2, 33, 64, False), (2, 130, 128, False), (4, 100, 64, True),
                             (2, 1, 64, False), (16, 4096, 72, False), (2, 100, 72, False), (3, 65, 72, False), (2, 33, 72, True),
                             (2, 1, 72, False), (1, 700, 72, True), (2, 1024, 72, True), (3, 1100, 72, False), (1, 1217, 72, True)):
        qkv = torch.randn((L, 3 * H * Dh), generator=g, device=DEV).to(dtype)
        q, k, v = (qkv[:, i * H * Dh:(i + 1) * H * Dh].view(L, H, Dh) for i in range(3))
        assert ops.causal_attention_ok(q, k, v)
        scale = Dh ** -0.5
        qd, kd, vd = (t.detach().double().requires_grad_() for t in (q, k, v))
        o_ref = _causal_reference(qd, kd, vd, scale, causal)
        do = torch.randn((L, H, Dh), generator=g, device=DEV).to(dtype)
        gq, gk, gv = torch.autograd.grad(o_ref, (qd, kd, vd), do.double())
        out, lse2 = ops.causal_attention(q, k, v, scale, causal)
        dq, dk, dv = ops.causal_attention_bwd(q, k, v, out, lse2, do, scale, causal=causal)
        rel = lambda a_, b_: float((a_.double() - b_).abs().max() / b_.abs().max().clamp_min(1e-2))      # noqa: E731
        assert rel(out, o_ref.detach()) <= 3 * eps, (H, L, Dh, causal, rel(out, o_ref.detach()))
        for name, mine, want in (("dq", dq, gq), ("dk", dk, gk), ("dv", dv, gv)):
            assert rel(mine, want) <= 6 * eps, (name, H, L, Dh, causal, rel(mine, want))
        if L > 1:
            ql, kl, vl = (t.detach().clone().requires_grad_() for t in (q, k, v))
            ol = torch.nn.functional.scaled_dot_product_attention(ql.transpose(0, 1)[None], kl.transpose(0, 1)[None], vl.transpose(0, 1)[None],
                                                                  is_causal=causal, scale=scale)[0].transpose(0, 1)
            lq, lk_, lv = torch.autograd.grad(ol, (ql, kl, vl), do)
            for name, mine, theirs, want in (("dq", dq, lq, gq), ("dk", dk, lk_, gk), ("dv", dv, lv, gv)):
                assert rel(mine, want) <= 1.5 * rel(theirs, want) + eps, (name, H, L, Dh, causal, rel(mine, want), rel(theirs, want))
        again = ops.causal_attention_bwd(q, k, v, out, lse2, do, scale, causal=causal)
        assert torch.equal(out, ops.causal_attention(q, k, v, scale, causal)[0]) and all(torch.equal(a_, b_) for a_, b_ in zip(again, (dq, dk, dv)))
    # through the tower's attention-interface function: (1, H, S, 64) views of a fused projection, with and without autograd
    from bimodalattack_amd import prefix_attention as pa
    H, S = 16, 577
    qkv = torch.randn((1, S, 3 * H * 64), generator=g, device=DEV).to(dtype).requires_grad_()
    heads = lambda x: tuple(t.transpose(1, 2) for t in x.view(1, S, 3, H, 64).unbind(2))      # noqa: E731
    do = torch.randn((1, S, H, 64), generator=g, device=DEV).to(dtype)
    out, _ = pa.padded_heads_attention(None, *heads(qkv), scaling=0.125)
    (g_own,) = torch.autograd.grad(out, qkv, do)
    try:
        ops.CAUSAL_ATTENTION = False
        out_l, _ = pa.padded_heads_attention(None, *heads(qkv), scaling=0.125)
        (g_lib,) = torch.autograd.grad(out_l, qkv, do)
    finally:
        ops.CAUSAL_ATTENTION = True
    assert out.shape == (1, S, H, 64)
    assert float((out.float() - out_l.float()).abs().max()) <= 4 * eps * float(out_l.float().abs().max())
    assert float((g_own.float() - g_lib.float()).abs().max()) <= 8 * eps * float(g_lib.float().abs().max())
    with torch.no_grad():
        out_n, _ = pa.padded_heads_attention(None, *heads(qkv), scaling=0.125)
    assert torch.equal(out_n, out.detach())
    # SigLIP's tower (4096 tokens x 16 heads of 72, views of the fused q/k/v product): the own pair against the library's
    # zero-padded route (pad to 128 with autograd / 96 without, slice the output)
    H, S, Dh = 16, 4096, 72
    qkv = torch.randn((1, S, 3 * H * Dh), generator=g, device=DEV).to(dtype).requires_grad_()
    heads72 = lambda x: tuple(t.transpose(1, 2) for t in x.view(1, S, 3, H, Dh).unbind(2))      # noqa: E731
    do = torch.randn((1, S, H, Dh), generator=g, device=DEV).to(dtype)
    calls = []
    keep = ops.causal_attention
    ops.causal_attention = lambda q_, *a_, **k_: (calls.append(tuple(q_.shape)), keep(q_, *a_, **k_))[1]
    try:
        out, _ = pa.padded_heads_attention(None, *heads72(qkv), scaling=Dh ** -0.5)
        (g_own,) = torch.autograd.grad(out, qkv, do)
        with torch.no_grad():
            out_n, _ = pa.padded_heads_attention(None, *heads72(qkv), scaling=Dh ** -0.5)
    finally:
        ops.causal_attention = keep
    assert calls == [(S, H, Dh)] * 2 and out.shape == (1, S, H, Dh) and torch.equal(out_n, out.detach())
    try:
        pa.OWN_TOWER_72 = False
        out_l, _ = pa.padded_heads_attention(None, *heads72(qkv), scaling=Dh ** -0.5)
        (g_lib,) = torch.autograd.grad(out_l, qkv, do)
    finally:
        pa.OWN_TOWER_72 = True
    assert float((out.float() - out_l.float()).abs().max()) <= 4 * eps * float(out_l.float().abs().max())
    assert float((g_own.float() - g_lib.float()).abs().max()) <= 8 * eps * float(g_lib.float().abs().max())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_causal_attention_forward_and_backward_match_float64(dtype):
    """bma_causal_attention / _bwd against float64 attention and its autograd on the same 16-bit operands: the image
    prompt's 643 and 599 tokens, 44 new rows behind 599 prefix keys (joint mode), lengths on and off the 64-row block and
    the 32-row chunk, a single query, 2 and 32 heads; q / k / v are strided views of one fused projection, as the decoder
    hands them over.  Bounds: the output within 3 roundings of the 16-bit type, the gradients within what the library's
    own flash pair shows against the same reference (P and dS rounded to 16 bits before their products) x 1.5.  Launches
    repeat bit for bit (no atomics)."""
    from bimodalattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(11)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for H, Lq, Lk in ((32, 643, 643), (32, 599, 599), (32, 44, 643), (2, 100, 100), (2, 1, 33), (4, 64, 64), (2, 65, 200), (2, 81, 81),
                      (2, 33, 32 + 33), (32, 1, 1), (40, 700, 700), (1, 4096, 4096), (3, 17, 2049)):
        qkv = torch.randn((Lk, 3 * H * 128), generator=g, device=DEV).to(dtype)
        q = qkv[Lk - Lq:, :H * 128].view(Lq, H, 128)
        k = qkv[:, H * 128:2 * H * 128].view(Lk, H, 128)
        v = qkv[:, 2 * H * 128:].view(Lk, H, 128)
        assert ops.causal_attention_ok(q, k, v)
        scale = 128 ** -0.5
        qd, kd, vd = (t.detach().double().requires_grad_() for t in (q, k, v))
        o_ref = _causal_reference(qd, kd, vd, scale)
        do = torch.randn((Lq, H, 128), generator=g, device=DEV).to(dtype)
        gq, gk, gv = torch.autograd.grad(o_ref, (qd, kd, vd), do.double())
        out, lse2 = ops.causal_attention(q, k, v, scale)
        dq, dk, dv = ops.causal_attention_bwd(q, k, v, out, lse2, do, scale)
        # (a single visible key has exactly zero dq / dk: measured against a floor instead of against nothing)
        rel = lambda a_, b_: float((a_.double() - b_).abs().max() / b_.abs().max().clamp_min(1e-2))      # noqa: E731
        assert rel(out, o_ref.detach()) <= 3 * eps, (H, Lq, Lk, rel(out, o_ref.detach()))
        for name, mine, want in (("dq", dq, gq), ("dk", dk, gk), ("dv", dv, gv)):
            assert rel(mine, want) <= 6 * eps, (name, H, Lq, Lk, rel(mine, want))
        if Lq == Lk and Lq > 1:
            ql, kl, vl = (t.detach().clone().requires_grad_() for t in (q, k, v))
            ol = torch.nn.functional.scaled_dot_product_attention(ql.transpose(0, 1)[None], kl.transpose(0, 1)[None], vl.transpose(0, 1)[None],
                                                                  is_causal=True, scale=scale)[0].transpose(0, 1)
            lq, lk_, lv = torch.autograd.grad(ol, (ql, kl, vl), do)
            for name, mine, theirs, want in (("dq", dq, lq, gq), ("dk", dk, lk_, gk), ("dv", dv, lv, gv)):
                assert rel(mine, want) <= 1.5 * rel(theirs, want) + eps, (name, H, Lq, Lk, rel(mine, want), rel(theirs, want))
        out2, lse2b = ops.causal_attention(q, k, v, scale)
        again = ops.causal_attention_bwd(q, k, v, out, lse2, do, scale)
        assert torch.equal(out, out2) and torch.equal(lse2, lse2b) and all(torch.equal(a_, b_) for a_, b_ in zip(again, (dq, dk, dv)))
        # the softmax denominator: lse2 = log2(sum exp(scaled score))
        ref_lse2 = torch.logsumexp((torch.einsum("qhd,khd->hqk", qd, kd) * scale).masked_fill(
            (torch.arange(Lk, device=DEV)[None, :] > (Lk - Lq + torch.arange(Lq, device=DEV))[:, None])[None], float("-inf")), -1) / 0.6931471805599453
        assert float((lse2.double() - ref_lse2.detach()).abs().max()) <= 1e-3
    # what it does not take
    q32 = torch.zeros((10, 2, 32), device=DEV, dtype=dtype)
    assert not ops.causal_attention_ok(q32, q32, q32)                         # head widths 64, 72, 128 and 256 only
    q3 = torch.zeros((10, 2, 128), device=DEV, dtype=dtype)
    assert not ops.causal_attention_ok(q3, q3[:5], q3[:5])                       # more queries than keys
    k3 = torch.zeros((10, 3, 128), device=DEV, dtype=dtype)
    assert not ops.causal_attention_ok(q3, k3, k3)                              # 2 query heads over 3 key/value heads
    assert not ops.causal_attention_ok(q3.float(), q3.float(), q3.float())
    assert not ops.causal_attention_ok(torch.zeros((10, 2, 256), device=DEV, dtype=dtype)[:, :, ::2], q3, q3)      # last dim not contiguous


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_causal_attention_grouped_queries_and_256_wide_heads(dtype):
    """The same three kernels with H query heads over Hkv key/value heads (HuggingFace's repeat_kv order: heads h*rep ..
    share key/value head h; dk / dv summed over the group INSIDE the launch) and at 256-wide heads (two 16-byte pieces per
    thread and chunk, the dk/dv launch splitting the output dims between its wave halves): Gemma-3's decoder in the gradient
    pass -- ~320 tokens x 8 heads over 4 of 256 -- lengths on and off the block and chunk sizes, a prefix in front, one
    key/value head for all, every key visible, grouped 64-, 72- and 128-wide heads; against float64 on the same 16-bit
    operands (k / v repeated, autograd doing the group's sum), and the gradients against the library's pair on repeated
    copies.  Bit-equal over repeated launches."""
    from bimodalattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(23)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for H, Hkv, Lq, Lk, Dh, causal in ((8, 4, 320, 320, 256, True), (8, 8, 643, 643, 256, True), (4, 1, 77, 77, 256, True),
                                       (8, 4, 44, 300, 256, True), (2, 2, 1, 1, 256, True), (4, 2, 65, 65, 256, False),
                                       (6, 2, 33, 64 + 33, 256, True), (8, 2, 200, 200, 128, True), (32, 8, 44, 643, 128, True),
                                       (4, 2, 130, 130, 64, False), (6, 3, 100, 100, 72, True), (8, 4, 1100, 1100, 256, True)):
        rep = H // Hkv
        q = torch.randn((Lq, H * Dh), generator=g, device=DEV).to(dtype).view(Lq, H, Dh)
        kv = torch.randn((Lk, 2 * Hkv * Dh), generator=g, device=DEV).to(dtype)
        k, v = kv[:, :Hkv * Dh].view(Lk, Hkv, Dh), kv[:, Hkv * Dh:].view(Lk, Hkv, Dh)
        assert ops.causal_attention_ok(q, k, v)
        scale = Dh ** -0.5
        qd, kd, vd = (t.detach().double().requires_grad_() for t in (q, k, v))
        o_ref = _causal_reference(qd, kd.repeat_interleave(rep, dim=1), vd.repeat_interleave(rep, dim=1), scale, causal)
        do = torch.randn((Lq, H, Dh), generator=g, device=DEV).to(dtype)
        gq, gk, gv = torch.autograd.grad(o_ref, (qd, kd, vd), do.double())
        out, lse2 = ops.causal_attention(q, k, v, scale, causal)
        dq, dk, dv = ops.causal_attention_bwd(q, k, v, out, lse2, do, scale, causal=causal)
        assert dq.shape == (Lq, H, Dh) and dk.shape == dv.shape == (Lk, Hkv, Dh)
        rel = lambda a_, b_: float((a_.double() - b_).abs().max() / b_.abs().max().clamp_min(1e-2))      # noqa: E731
        case = (H, Hkv, Lq, Lk, Dh, causal)
        assert rel(out, o_ref.detach()) <= 3 * eps, (case, rel(out, o_ref.detach()))
        for name, mine, want in (("dq", dq, gq), ("dk", dk, gk), ("dv", dv, gv)):
            # (the group's sum is taken in fp32 before the one rounding: no worse than a single head's)
            assert rel(mine, want) <= 6 * eps, (name, case, rel(mine, want))
        if Lq == Lk and Lq > 1:
            ql, kl, vl = (t.detach().clone().requires_grad_() for t in (q, k, v))
            ol = torch.nn.functional.scaled_dot_product_attention(
                ql.transpose(0, 1)[None], kl.repeat_interleave(rep, dim=1).transpose(0, 1)[None],
                vl.repeat_interleave(rep, dim=1).transpose(0, 1)[None], is_causal=causal, scale=scale)[0].transpose(0, 1)
            lq, lk_, lv = torch.autograd.grad(ol, (ql, kl, vl), do)
            for name, mine, theirs, want in (("dq", dq, lq, gq), ("dk", dk, lk_, gk), ("dv", dv, lv, gv)):
                assert rel(mine, want) <= 1.5 * rel(theirs, want) + eps, (name, case, rel(mine, want), rel(theirs, want))
        out2, lse2b = ops.causal_attention(q, k, v, scale, causal)
        again = ops.causal_attention_bwd(q, k, v, out, lse2, do, scale, causal=causal)
        assert torch.equal(out, out2) and torch.equal(lse2, lse2b) and all(torch.equal(a_, b_) for a_, b_ in zip(again, (dq, dk, dv)))
    # through the decoder's attention-interface function, as Gemma-3's attention block calls it: (1, H, S, 256) against
    # (1, Hkv, S, 256), under autograd; the library route (repeated k / v, flash kernels) as the yardstick
    from bimodalattack_amd import prefix_attention as pa
    H, Hkv, S, Dh = 8, 4, 323, 256
    q4 = torch.randn((1, S, H, Dh), generator=g, device=DEV).to(dtype).requires_grad_()
    k4 = torch.randn((1, S, Hkv, Dh), generator=g, device=DEV).to(dtype).requires_grad_()
    v4 = torch.randn((1, S, Hkv, Dh), generator=g, device=DEV).to(dtype).requires_grad_()
    do = torch.randn((1, S, H, Dh), generator=g, device=DEV).to(dtype)
    calls = []
    keep = ops.causal_attention
    ops.causal_attention = lambda q_, *a_, **k_: (calls.append(tuple(q_.shape)), keep(q_, *a_, **k_))[1]
    try:
        out, _ = pa.causal_b1_attention(None, q4.transpose(1, 2), k4.transpose(1, 2), v4.transpose(1, 2), scaling=Dh ** -0.5)
        g_own = torch.autograd.grad(out, (q4, k4, v4), do)
    finally:
        ops.causal_attention = keep
    assert calls == [(S, H, Dh)] and out.shape == (1, S, H, Dh)
    try:
        pa.OWN_WIDE_HEADS = False
        out_l, _ = pa.causal_b1_attention(None, q4.transpose(1, 2), k4.transpose(1, 2), v4.transpose(1, 2), scaling=Dh ** -0.5)
        g_lib = torch.autograd.grad(out_l, (q4, k4, v4), do)
    finally:
        pa.OWN_WIDE_HEADS = True
    assert float((out.float() - out_l.float()).abs().max()) <= 4 * eps * float(out_l.float().abs().max())
    for a_, b_ in zip(g_own, g_lib):
        assert a_.shape == b_.shape and float((a_.float() - b_.float()).abs().max()) <= 8 * eps * float(b_.float().abs().max())


def test_causal_attention_under_autograd_in_a_graph_and_through_the_interface():
    """CausalAttentionFn against autograd through the library on views of one fused projection; the pair captured into a
    hipGraph and replayed; and the two attention-interface functions of the gradient pass (prefix_attention.py) handing
    (1, H, L, 128) tensors to it -- full causal, and 44 rows behind keys/values with autograd history -- and stepping aside
    for grouped heads, head widths other than 64 / 128 and dropout."""
    from bimodalattack_amd import ops
    from bimodalattack_amd import prefix_attention as pa
    g = torch.Generator(device=DEV).manual_seed(12)
    H, S = 32, 643
    qkv = torch.randn((1, S, 3 * H * 128), generator=g, device=DEV).to(torch.bfloat16).requires_grad_()
    do = torch.randn((1, S, H, 128), generator=g, device=DEV).to(torch.bfloat16)

    def heads(x):
        q, k, v = x.view(1, S, 3, H, 128).unbind(2)
        return q.transpose(1, 2), k.transpose(1, 2), v.transpose(1, 2)        # (1, H, S, 128) views, as HuggingFace passes them
    q, k, v = heads(qkv)
    out, _ = pa.causal_b1_attention(None, q, k, v, scaling=128 ** -0.5)
    assert out.shape == (1, S, H, 128) and type(out.grad_fn).__name__ != "TransposeBackward0"
    (g_own,) = torch.autograd.grad(out, qkv, do)
    try:
        ops.CAUSAL_ATTENTION = False
        out_l, _ = pa.causal_b1_attention(None, *heads(qkv), scaling=128 ** -0.5)
        (g_lib,) = torch.autograd.grad(out_l, qkv, do)
    finally:
        ops.CAUSAL_ATTENTION = True
    assert float((out.float() - out_l.float()).abs().max()) <= 2 ** -6 * float(out_l.float().abs().max())
    assert float((g_own.float() - g_lib.float()).abs().max()) <= 2 ** -5 * float(g_lib.float().abs().max())
    # refusals: query heads that do not divide over the key/value heads (grouped heads are taken since round 5), 32-wide heads, dropout
    assert pa._own_causal(q, k[:, :5], v[:, :5], 0.1, 0.0) is None
    assert pa._own_causal(q, k[:, :8], v[:, :8], 0.1, 0.0) is not None
    assert pa._own_causal(q[..., :32], k[..., :32], v[..., :32], 0.1, 0.0) is None
    assert pa._own_causal(q, k, v, 0.1, 0.1) is None
    # the tail behind a prefix with history: gradients reach the prefix keys/values and the new rows alike
    P, L = 599, 44
    pk = torch.randn((1, H, P, 128), generator=g, device=DEV).to(torch.bfloat16).requires_grad_()
    pv = torch.randn((1, H, P, 128), generator=g, device=DEV).to(torch.bfloat16).requires_grad_()
    tq, tk, tv = (torch.randn((1, H, L, 128), generator=g, device=DEV).to(torch.bfloat16).requires_grad_() for _ in range(3))
    dt = torch.randn((1, L, H, 128), generator=g, device=DEV).to(torch.bfloat16)

    class KV:
        k, v = [pk], [pv]

        @staticmethod
        def bias(L_, dtype, device):
            b = torch.zeros((L_, P + L_), dtype=dtype, device=device)
            b[:, P:] = torch.full((L_, L_), float("-inf"), dtype=dtype, device=device).triu(1)
            return b

    class Mod:
        layer_idx = 0
    pa._ACTIVE.append(KV)
    try:
        o1, _ = pa.tail_grad_attention(Mod, tq, tk, tv, scaling=128 ** -0.5)
        g1 = torch.autograd.grad(o1, (tq, tk, tv, pk, pv), dt)
        ops.CAUSAL_ATTENTION = False
        o2, _ = pa.tail_grad_attention(Mod, tq, tk, tv, scaling=128 ** -0.5)
        g2 = torch.autograd.grad(o2, (tq, tk, tv, pk, pv), dt)
    finally:
        ops.CAUSAL_ATTENTION = True
        pa._ACTIVE.pop()
    assert float((o1.float() - o2.float()).abs().max()) <= 2 ** -6 * float(o2.float().abs().max())
    for a_, b_ in zip(g1, g2):
        assert float((a_.float() - b_.float()).abs().max()) <= 2 ** -5 * float(b_.float().abs().max())
    # captured and replayed
    q3, k3, v3 = (t.detach().squeeze(0).transpose(0, 1) for t in heads(qkv))
    xs = qkv.detach().clone()
    qs, ks, vs = (t.squeeze(0).transpose(0, 1) for t in (lambda x: (x.view(1, S, 3, H, 128).unbind(2)))(xs))
    qs, ks, vs = (t.transpose(0, 1).transpose(0, 1) for t in (qs, ks, vs))
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        o_g, l_g = ops.causal_attention(qs, ks, vs, 128 ** -0.5)
        grads_g = ops.causal_attention_bwd(qs, ks, vs, o_g, l_g, do[0], 128 ** -0.5)
    for kk in range(2):
        xs.copy_(qkv.detach() * (0.5 + kk))
        graph.replay()
        torch.cuda.synchronize()
        o_e, l_e = ops.causal_attention(qs, ks, vs, 128 ** -0.5)
        grads_e = ops.causal_attention_bwd(qs, ks, vs, o_e, l_e, do[0], 128 ** -0.5)
        assert torch.equal(o_g, o_e) and all(torch.equal(a_, b_) for a_, b_ in zip(grads_g, grads_e))


def test_rotary_causal_attention_block_equals_its_parts():
    """ops.RotaryCausalAttentionFn -- rotary, causal attention and the gradient written straight into the fused q/k/v
    projection's -- against the same steps as separate autograd functions (RoPE2Fn on views, CausalAttentionFn, autograd's
    own concatenation): the same bits forward and backward, at the image prompt's 643 tokens and at 100."""
    from bimodalattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(13)
    H = 32
    for S in (643, 100):
        qkv = torch.randn((1, S, 3 * H * 128), generator=g, device=DEV).to(torch.bfloat16)
        ang = torch.rand((S, 64), generator=g, device=DEV) * 6.28
        cos = torch.cat([ang.cos(), ang.cos()], -1).to(torch.bfloat16)
        sin = torch.cat([ang.sin(), ang.sin()], -1).to(torch.bfloat16)
        do = torch.randn((1, S, H * 128), generator=g, device=DEV).to(torch.bfloat16)
        assert ops.rotary_causal_attention_ok(qkv, cos.unsqueeze(0), H)
        a = qkv.clone().requires_grad_()
        out_a = ops.RotaryCausalAttentionFn.apply(a, cos, sin, H, 128 ** -0.5)
        (ga,) = torch.autograd.grad(out_a, a, do)
        b = qkv.clone().requires_grad_()
        x = b.view(1, S, 3, H, 128)
        qo, ko = ops.RoPE2Fn.apply(x[:, :, 0].transpose(1, 2), x[:, :, 1].transpose(1, 2), cos.unsqueeze(0), sin.unsqueeze(0))
        out_b = ops.CausalAttentionFn.apply(qo.squeeze(0).transpose(0, 1), ko.squeeze(0).transpose(0, 1), x[0, :, 2], 128 ** -0.5)
        (gb,) = torch.autograd.grad(out_b.reshape(1, S, H * 128), b, do)
        assert torch.equal(out_a, out_b.reshape(1, S, H * 128)) and torch.equal(ga, gb)
    assert not ops.rotary_causal_attention_ok(qkv[:, :80], cos[:80].unsqueeze(0), H)          # the one-launch kernel's range
    assert not ops.rotary_causal_attention_ok(qkv.float(), cos.unsqueeze(0).float(), H)


# ------------------------------------------------------------------ bma_gemm_mid (round 4)
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_gemm_mid_matches_fp32_reference(dtype, monkeypatch):
    """y = x W^T on the 224-row-tile kernel against the product in fp32 (one rounding of the fp32 sum to the 16-bit type):
    the seven products of a LLaVA-7B layer at the row counts of the pass with the image in the prompt (599 / 643 / 644:
    every tile plan -- K split 5 ways, 192- and 256-wide tiles unsplit, the last column of tiles split 8 ways behind 255
    whole ones), row counts on and off the 224-row tile and down to one row, N off the tile, off 64 and off 16, one and two
    units of K; bitwise equal over repeated launches (the second launch adds the partials in split order).  ops.gemm_mid_ok
    routes only the shapes where the kernel beats the library: lifted here."""
    import ctypes
    from bimodalattack_amd import ops
    from bimodalattack_amd.native import lib
    g = torch.Generator(device=DEV).manual_seed(321)
    monkeypatch.setattr(ops, "GEMM_MID_MIN_K_OVER_N", 0.0)
    monkeypatch.setattr(ops, "GEMM_MID_MIN_ROWS", 1)
    shapes = [(644, 4096, 4096), (644, 22016, 4096), (599, 4096, 22016), (643, 12288, 4096), (644, 4096, 11008), (644, 11008, 4096),
              (644, 4096, 12288), (672, 260, 64), (449, 132, 128), (1, 64, 64), (225, 200, 192), (448, 8, 64), (600, 1028, 4096),
              (17, 32064, 64), (224, 256, 1024)]
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    plan = (ctypes.c_int * 8)()
    seen = set()
    for M, N, K in shapes:
        x = (torch.randn((M, K), generator=g, device=DEV)).to(dtype)
        w = (torch.randn((N, K), generator=g, device=DEV) * 0.05).to(dtype)
        assert ops.gemm_mid_ok(x, w)
        assert lib.bma_gemm_mid_plan(M, N, K, plan) == 0
        seen.add((plan[2], plan[4] > 1, 0 < plan[7] < plan[1] * plan[3]))
        y = ops.gemm_mid(x, w)
        ref = (x.double() @ w.double().t()).float()               # float64: the fp32 sums' own error is priced below
        err = (y.float() - ref).abs()
        # one rounding to the 16-bit type (1 ulp where the fp32 sum sits on a rounding boundary) + what K fp32 additions
        # of terms of this size can have drifted (random walk, 8 sigma)
        tol = 2.02 * eps * ref.abs() + 8.0 * K ** 0.5 * 2.0 ** -24 * ref.abs().max() + (2.0 ** -24 if dtype == torch.float16 else 0.0)
        assert y.shape == (M, N) and bool((err <= tol).all()), (M, N, K, float(err.max()), float((err - tol).max()))
        for _ in range(3):
            assert torch.equal(ops.gemm_mid(x, w), y)
        lib_y = torch.nn.functional.linear(x, w)
        assert float((y.float() - lib_y.float()).abs().max()) <= 2.5 * eps * float(ref.abs().max())
    assert {(4, True, False), (3, False, False), (4, True, True)} <= seen          # every kind of plan ran
    # every pinned decomposition of one product gives the same rounding of the same sums up to the order of the K pieces
    x = (torch.randn((644, 4096), generator=g, device=DEV)).to(dtype)
    w = (torch.randn((1000, 4096), generator=g, device=DEV) * 0.05).to(dtype)
    ref = x.float() @ w.float().t()
    try:
        for nf, S, tail in ((3, 1, 0), (4, 1, 0), (4, 3, 0), (3, 8, 0), (4, 2, 1), (3, 5, 2)):
            lib.bma_gemm_mid_set_plan(nf, S, tail, -1)
            y = ops.gemm_mid(x, w)
            assert float((y.float() - ref).abs().max()) <= 2.5 * eps * float(ref.abs().max()), (nf, S, tail)
            lib.bma_gemm_mid_set_plan(nf, S, tail, 0)                 # without the XCD-contiguous order: the same bits
            assert torch.equal(ops.gemm_mid(x, w), y)
    finally:
        lib.bma_gemm_mid_set_plan(0, 0, -1, -1)
    # padded leading dimensions through the C ABI (a view of a wider buffer on either side, a wider output)
    from bimodalattack_amd.native import check
    xb = torch.randn((500, 4096 + 64), generator=g, device=DEV).to(dtype)
    wb = (torch.randn((300, 4096 + 128), generator=g, device=DEV) * 0.05).to(dtype)
    yb = torch.full((500, 320), 7.0, device=DEV, dtype=dtype)
    pair = ops.gemm_workspace(torch.device(DEV))
    check("bma_gemm_mid", lib.bma_gemm_mid(xb.data_ptr(), 4096 + 64, wb.data_ptr(), 4096 + 128, yb.data_ptr(), 320, 500, 300, 4096,
                                           1 if dtype == torch.bfloat16 else 2, pair[0].data_ptr(), pair[0].numel(),
                                           torch.cuda.current_stream().cuda_stream))
    ref = xb[:, :4096].float() @ wb[:, :4096].float().t()
    assert float((yb[:, :300].float() - ref).abs().max()) <= 2.5 * eps * float(ref.abs().max())
    assert bool((yb[:, 300:] == 7.0).all())                            # nothing written past N
    # a 3-D activation, as the decoder hands it over; what it does not take goes to the library
    x3 = torch.randn((1, 644, 11008), generator=g, device=DEV).to(dtype)
    w3 = (torch.randn((4096, 11008), generator=g, device=DEV) * 0.05).to(dtype)
    monkeypatch.undo()
    assert ops.gemm_mid_ok(x3, w3) and ops.linear_b1(x3, w3).shape == (1, 644, 4096)
    assert torch.equal(ops.linear_b1(x3, w3)[0], ops.gemm_mid(x3[0], w3))
    assert not ops.gemm_mid_ok(x3[:, :500], w3) and not ops.gemm_mid_ok(torch.zeros((700, 11008), device=DEV, dtype=dtype), w3)
    assert not ops.gemm_mid_ok(torch.zeros((644, 4096), device=DEV, dtype=dtype), torch.zeros((4096, 4096), device=DEV, dtype=dtype))
    assert not ops.gemm_mid_ok(x3.float(), w3.float())
    monkeypatch.setattr(ops, "MID_GEMM", False)
    assert not ops.gemm_mid_ok(x3, w3)


def test_gemm_mid_under_autograd_and_in_a_graph():
    """FrozenLinearFn at 644 rows (forward on the library or the kernel as routed, the input gradient through the transposed
    copy on the kernel) against autograd through the library, and the pair captured into a hipGraph and replayed."""
    from bimodalattack_amd import ops
    g = torch.Generator(device=DEV).manual_seed(8)
    x = torch.randn((1, 644, 4096), generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn((22016, 4096), generator=g, device=DEV) * 0.02).to(torch.bfloat16)
    wt = w.t().contiguous()
    dy = torch.randn((1, 644, 22016), generator=g, device=DEV).to(torch.bfloat16)
    assert ops.gemm_mid_ok(x, w) and ops.gemm_mid_ok(dy, wt)
    xa = x.clone().requires_grad_()
    ya = ops.FrozenLinearFn.apply(xa, w, wt)
    (ga,) = torch.autograd.grad(ya, xa, dy)
    xb = x.clone().requires_grad_()
    yb = torch.nn.functional.linear(xb, w)
    (gb,) = torch.autograd.grad(yb, xb, dy)
    assert float((ya.float() - yb.float()).abs().max()) <= 2 ** -7 * float(yb.float().abs().max())
    assert float((ga.float() - gb.float()).abs().max()) <= 2 ** -7 * float(gb.float().abs().max())
    assert ops.gemm_workspace_for_graphs(torch.device(DEV)) is not None
    xs = x.clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.gemm_mid(xs, w)
        out2 = ops.gemm_mid(out, wt)
    for k in range(3):
        xs.copy_(x * (k + 1))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ops.gemm_mid(xs, w)) and torch.equal(out2, ops.gemm_mid(out, wt))


def test_gemm_nt_under_autograd_and_in_a_graph(monkeypatch):
    """FrozenLinearFn on the skinny kernel (forward and the input gradient through the transposed copy) against autograd
    through the library, and the same pair captured into a hipGraph and replayed."""
    from bimodalattack_amd import ops
    monkeypatch.setattr(ops, "GEMM_NT_MIN_K_OVER_N", 0.0)        # every shape on the kernel
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn((1, 65, 4096), generator=g, device=DEV).to(torch.bfloat16)
    w = (torch.randn((11008, 4096), generator=g, device=DEV) * 0.02).to(torch.bfloat16)
    wt = w.t().contiguous()
    dy = torch.randn((1, 65, 11008), generator=g, device=DEV).to(torch.bfloat16)
    xa = x.clone().requires_grad_()
    ya = ops.FrozenLinearFn.apply(xa, w, wt)
    (ga,) = torch.autograd.grad(ya, xa, dy)
    xb = x.clone().requires_grad_()
    yb = torch.nn.functional.linear(xb, w)
    (gb,) = torch.autograd.grad(yb, xb, dy)
    assert float((ya.float() - yb.float()).abs().max()) <= 2 ** -7 * float(yb.float().abs().max())
    assert float((ga.float() - gb.float()).abs().max()) <= 2 ** -7 * float(gb.float().abs().max())
    xs = x.clone()
    side = torch.cuda.Stream(DEV)
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        ops.gemm_nt(xs, w)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    # captured launches use the device's pair for graphs (allocated outside the capture), eager ones a pair per stream:
    # an eager product on another stream cannot race a replay on partials and tickets
    assert ops.gemm_workspace_for_graphs(torch.device(DEV)) is not None
    main_pair = ops.gemm_workspace(torch.device(DEV))
    with torch.cuda.stream(side):
        side_pair = ops.gemm_workspace(torch.device(DEV))
    assert main_pair[0].data_ptr() != side_pair[0].data_ptr() != ops.gemm_workspace_for_graphs(torch.device(DEV))[0].data_ptr()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ops.gemm_nt(xs, w)
        out2 = ops.gemm_nt(out, wt)
    for k in range(3):
        xs.copy_(x * (k + 1))
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, ops.gemm_nt(xs, w)) and torch.equal(out2, ops.gemm_nt(out, wt))
    # ... and the engine's captured graphs, which all share that one pair, say so when a second stream replays one
    keep = dict(ops._GRAPH_REPLAY_STREAM)
    try:
        ops._GRAPH_REPLAY_STREAM.clear()
        ops.note_graph_replay(torch.device(DEV))
        ops.note_graph_replay(torch.device(DEV))
        with torch.cuda.stream(side), pytest.raises(RuntimeError, match="one stream"):
            ops.note_graph_replay(torch.device(DEV))
    finally:
        ops._GRAPH_REPLAY_STREAM.clear()
        ops._GRAPH_REPLAY_STREAM.update(keep)


@pytest.mark.gpu
def test_allgather_f32_world_of_one_through_rccl():
    """bma_allgather_f32 against a one-rank RCCL communicator made with the RCCL instance torch loaded (the entry point
    must find THAT instance, not link its own), in place and out of place, and without a communicator."""
    import ctypes
    import torch
    from bimodalattack_amd import native
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    x = torch.randn(1000, device=dev)
    out = torch.full((1000,), float("nan"), device=dev)
    st = torch.cuda.current_stream(dev).cuda_stream
    assert native.lib.bma_allgather_f32(x.data_ptr(), 1000, out.data_ptr(), 0, 1, None, st) == 0
    torch.cuda.synchronize()
    assert torch.equal(out, x)
    rccl = None
    for name in ("librccl.so.1", "librccl.so"):
        try:
            rccl = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", name))
            break
        except OSError:
            continue
    if rccl is None:
        pytest.skip("no RCCL next to torch")
    class UniqueId(ctypes.Structure):                              # ncclUniqueId: 128 opaque bytes, passed BY VALUE
        _fields_ = [("internal", ctypes.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0
    try:
        out.fill_(float("nan"))
        assert native.lib.bma_allgather_f32(x.data_ptr(), 1000, out.data_ptr(), 0, 1, comm, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(out, x)
        y = x.clone()                                                # in place: local = out + rank * n_local
        assert native.lib.bma_allgather_f32(y.data_ptr(), 1000, y.data_ptr(), 0, 1, comm, st) == 0
        torch.cuda.synchronize()
        assert torch.equal(y, x)
    finally:
        rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
        rccl.ncclCommDestroy(comm)


# ------------------------------------------------------------------ bma_b1_attention (round 4)
def _rope_tables(S, dtype, gen):
    ang = torch.rand((S, 64), generator=gen, device=DEV) * 6.28
    return torch.cat([ang.cos(), ang.cos()], -1).to(dtype), torch.cat([ang.sin(), ang.sin()], -1).to(dtype)


def _b1_reference(qkv, cos, sin, H, scale):
    """HuggingFace's route on the same numbers: apply_rotary_pos_emb in the 16-bit type (its three roundings), then causal
    attention in fp32 on the rotated 16-bit q / k -- with autograd through all of it."""
    S = qkv.shape[0]
    q, k, v = (t.view(S, H, 128).transpose(0, 1) for t in qkv.split(H * 128, dim=-1))       # (H,S,128)

    def rot(x):
        x1, x2 = x[..., :64], x[..., 64:]
        return (x * cos) + (torch.cat((-x2, x1), dim=-1) * sin)
    qr, kr = rot(q).float(), rot(k).float()
    s = (qr @ kr.transpose(1, 2)) * scale
    s = s.masked_fill(~torch.tril(torch.ones(S, S, dtype=torch.bool, device=qkv.device)), float("-inf"))
    lse = torch.logsumexp(s, dim=-1)
    out = torch.softmax(s, dim=-1) @ v.float()
    return out.transpose(0, 1).reshape(S, H * 128), lse


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_b1_attention_forward_and_backward(dtype):
    """Rotary + causal attention of one short sequence in one launch each way (the batch-1 gradient pass over a text-only
    prompt) against HuggingFace's formulation with fp32 attention: outputs and log-sum-exp to the rounding of the 16-bit
    probabilities, d(qkv) to a few percent of its scale element-wise and 1 % in norm; sequence lengths on and off the
    16-row tiles up to the 80-token limit, one head and 32; row strides wider than the row (a view into a larger
    buffer); same bits on repeated launches; beyond the limits the call refuses."""
    from bimodalattack_amd import ops
    from bimodalattack_amd.native import lib
    g = torch.Generator(device=DEV).manual_seed(7)
    eps = 2.0 ** -8 if dtype == torch.bfloat16 else 2.0 ** -11
    for S, H in ((65, 32), (44, 32), (1, 2), (16, 1), (17, 3), (80, 4), (33, 32)):
        scale = 128 ** -0.5
        big = (torch.randn((S, 3 * H * 128 + 64), generator=g, device=DEV) * 1.5).to(dtype)
        qkv = big[:, : 3 * H * 128]                            # row stride 3*H*128 + 64
        cos, sin = _rope_tables(S, dtype, g)
        assert ops.b1_attention_ok(qkv, cos, H, H, 128)
        leaf = qkv.detach().clone().requires_grad_()
        ref, ref_lse = _b1_reference(leaf, cos, sin, H, scale)
        dout = (torch.randn((S, H * 128), generator=g, device=DEV)).to(dtype)
        (ref_grad,) = torch.autograd.grad(ref, leaf, dout.float())
        mine = qkv.detach().requires_grad_()
        out = ops.B1AttentionFn.apply(mine, cos, sin, H, scale)
        (grad,) = torch.autograd.grad(out, mine, dout)
        out2, lse = ops.b1_attention(qkv, cos, sin, H, scale)
        assert torch.equal(out2, out.detach()) and out.shape == (S, H * 128) and grad.shape == (S, 3 * H * 128)
        err = (out.float() - ref).abs().max() / ref.abs().max()
        assert float(err) < 6 * eps, (S, H, float(err))
        assert float((lse - ref_lse).abs().max()) < 2e-3 * max(1.0, float(ref_lse.abs().max())), (S, H)
        gerr = (grad.float() - ref_grad.float()).abs().max() / ref_grad.abs().max()
        gnorm = (grad.float() - ref_grad.float()).norm() / ref_grad.float().norm()
        assert float(gerr) < 16 * eps and float(gnorm) < 4 * eps, (S, H, float(gerr), float(gnorm))
        (grad2,) = torch.autograd.grad(ops.B1AttentionFn.apply(mine, cos, sin, H, scale), mine, dout)
        assert torch.equal(grad, grad2)
    # what it does not take
    x = torch.zeros((81, 3 * 128), device=DEV, dtype=dtype)
    c = torch.zeros((81, 128), device=DEV, dtype=dtype)
    assert not ops.b1_attention_ok(x, c, 1, 1, 128)
    assert lib.bma_b1_attention(x.data_ptr(), 384, c.data_ptr(), c.data_ptr(), 81, 1, 1, 0.1, x.data_ptr(), 128, x.data_ptr(), None) == -5
    assert not ops.b1_attention_ok(x[:8], c[:8], 2, 1, 128) and not ops.b1_attention_ok(x[:8].float(), c[:8].float(), 1, 1, 128)
    assert lib.bma_b1_attention(x.data_ptr(), 384, c.data_ptr(), c.data_ptr(), 8, 1, 0, 0.1, x.data_ptr(), 128, x.data_ptr(), None) == -2
    assert lib.bma_b1_attention(x.data_ptr(), 100, c.data_ptr(), c.data_ptr(), 8, 1, 1, 0.1, x.data_ptr(), 128, x.data_ptr(), None) == -1
