"""End-to-end parity of the HIP engine with trajectories captured from the real
reference on tiny random models (tests/golden/g5_*), mirroring the reference's own
smoke matrix (run_tests.sh: PGD-only, GCG-only, PGD+GCG, joint) plus Gemma-3 and the
dynamic-width / n_replace / buffer paths.  fp32 models, randoms drawn on the CPU
generator as the reference's CPU path draws them.

Bars: sampled ids, filter survivors, winner strings exact; losses <= 1e-4 rel; the
PGD image bit-exact except where a device-vs-CPU rounding difference flips the sign of
an (almost) zero pixel gradient -- at most 0.5 % of pixels, each by exactly one step.
"""

import json
import os
import tempfile

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

with open(os.path.join(os.path.dirname(__file__), "golden", "g5_meta.json")) as _f:
    META = json.load(_f)


def run_case(name, **engine):
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    m = META["cases"][name]
    model, tok, proc, image = S.tiny_case(m["kind"], device=DEV)
    tmp = tempfile.mkdtemp(prefix="bma_gpu_")
    cfg = BimodalAttackConfig(seed=1, verbosity="ERROR", optim_str_init=m["optim_str_init"], images_folder=tmp,
                              **m["config"])
    trace = []
    res = run(model, tok, proc, m["goal"], m["goal"], m["target"], image, cfg,
              normalize=S.Normalize(S.CLIP_MEAN, S.CLIP_STD), rng_device="cpu", trace=trace, **engine)
    return m, res, trace, tmp


def check_against_golden(golden_dir, name, m, res, trace, tmp):
    z = np.load(os.path.join(golden_dir, f"g5_{name}.npz"))
    assert len(trace) == m["steps"]
    eps, alpha = m["config"].get("eps", 0), m["config"].get("alpha", 0)
    for i, st in enumerate(trace):
        assert st["n_grad"] == int(z[f"s{i}_n_grad"])
        assert np.array_equal(st["optim_ids_in"], z[f"s{i}_optim_ids_in"]), f"step {i}: optim ids"
        if st["grad_tok"]:
            want = z[f"s{i}_grad_tok0"]
            fin = np.isfinite(want)                  # the reference masked its copy with +inf in place
            np.testing.assert_allclose(st["grad_tok"][-1][fin], want[fin], rtol=2e-3, atol=2e-6)
        for j, g in enumerate(st["grad_img"]):
            np.testing.assert_allclose(g, z[f"s{i}_grad_img{j}"], rtol=5e-3, atol=1e-6)
        if "image_after_pgd" in st:
            want = z[f"s{i}_image_after_pgd"]
            diff = st["image_after_pgd"] != want
            assert diff.mean() <= 0.005, f"step {i}: {diff.sum()} pixels differ"
            assert np.abs(st["image_after_pgd"] - want).max() <= 2 * alpha * eps + 1e-6
        if "sampled" in st:
            assert np.array_equal(st["sampled"], z[f"s{i}_sampled"]), f"step {i}: sampled ids"
        if "filtered" in st:
            assert np.array_equal(st["filtered"], z[f"s{i}_filtered"]), f"step {i}: filter survivors"
        for j, l in enumerate(st["losses"]):
            np.testing.assert_allclose(l, z[f"s{i}_loss{j}"], rtol=1e-4)
    np.testing.assert_allclose(res.losses, z["losses"], rtol=1e-4)
    np.testing.assert_allclose(res.best_loss, float(z["best_loss"]), rtol=1e-4)
    assert res.strings == m["strings"] and res.best_string == m["best_string"]
    assert res.adversarial_suffixes == m["adversarial_suffixes"]
    assert [len(getattr(res, k)) for k in ("gradient_times", "sampling_times", "loss_times", "pgd_times",
                                           "total_times")] == m["n_timing"]
    assert res.model_outputs == [""] * m["steps"]
    png = os.path.join(golden_dir, f"g5_{name}_png0.npz")
    if os.path.exists(png):
        from PIL import Image
        got = np.array(Image.open(os.path.join(tmp, "0.png")))
        want = np.load(png)["png"]
        assert got.shape == want.shape and (got != want).mean() <= 0.005
        assert sorted(os.listdir(tmp)) == sorted(f"{i}.png" for i in range(m["steps"]))


@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_trajectory_matches_reference(golden_dir, name):
    m, res, trace, tmp = run_case(name)
    check_against_golden(golden_dir, name, m, res, trace, tmp)


@pytest.mark.parametrize("name", ["llava_joint", "opt_gcg", "gemma3_joint", "llava_pgd_gcg"])
@pytest.mark.parametrize("engine", [dict(prefix_reuse=False), dict(prefix_reuse=False, target_rows_only=False),
                                    dict(chunk=5)])
def test_restructurings_do_not_change_results(golden_dir, name, engine):
    """Full-sequence forward / full logits (the reference's call shape) and odd chunk
    sizes give the same trajectory as prefix reuse + target rows only."""
    m, res, trace, tmp = run_case(name, **engine)
    check_against_golden(golden_dir, name, m, res, trace, tmp)


def test_device_rng_mode_and_bf16_run():
    """Default mode draws the randoms on the device like the reference does on a GPU; a
    bf16 model exercises the 16-bit kernels end to end.  No CPU golden exists for either,
    so: same seed -> same run; losses finite; candidates differ from the parent in
    exactly n_replace positions; every top-k id is allowed."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    out = []
    for _ in range(2):
        model, tok, proc, image = S.tiny_case("llava", dtype=torch.bfloat16, device=DEV)
        trace = []
        cfg = BimodalAttackConfig(num_steps=3, search_width=32, topk=16, pgd_attack=True, gcg_attack=True,
                                  joint_eval=True, eps=64 / 255, alpha=4 / 255, seed=7, verbosity="ERROR",
                                  optim_str_init=S.TINY_OPTIM_INIT, images_folder=tempfile.mkdtemp())
        res = run(model, tok, proc, "tell me a story", "tell me a story", "Sure here is a story", image, cfg,
                  normalize=S.Normalize(S.CLIP_MEAN, S.CLIP_STD), trace=trace)
        out.append((res, trace))
    (r0, t0), (r1, t1) = out
    assert r0.losses == r1.losses and r0.strings == r1.strings
    assert all(np.isfinite(r0.losses))
    from oracle import kernels as K
    tok = S.build_tokenizer(S.TINY_WORDS, S.TINY_NONASCII, S.TINY_UNRT)
    na = set(K.nonascii_tokens(tok).tolist())
    for st in t0:
        assert np.array_equal(st["sampled"], t1[t0.index(st)]["sampled"])
        assert ((st["sampled"] != st["optim_ids_in"]).sum(1) <= 1).all()
        assert not (set(st["topk_idx"].reshape(-1).tolist()) & na)


def test_early_stop_and_errors():
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    model, tok, proc, _ = S.tiny_case("opt", device=DEV)
    base = dict(seed=1, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT, images_folder=tempfile.mkdtemp())
    with pytest.raises(TypeError):
        run(model, tok, proc, "a", "a", "Sure", None, BimodalAttackConfig(pgd_after_gcg=True, **base))
    with pytest.raises(ValueError, match="needs an image"):
        run(model, tok, proc, "a", "a", "Sure", None, BimodalAttackConfig(pgd_attack=True, **base))
    # every candidate contains an un-roundtrippable token -> the filter keeps nothing
    with pytest.raises(RuntimeError, match="No token sequences are the same"):
        run(model, tok, proc, "a", "a", "Sure", None,
            BimodalAttackConfig(num_steps=1, search_width=4, topk=4, **dict(base, optim_str_init=["ab0 cd"] * 1)),
            rng_device="cpu")
    cpu_model, _, _, _ = S.tiny_case("opt")
    with pytest.raises(RuntimeError, match="AMD GPU only"):
        run(cpu_model, tok, proc, "a", "a", "Sure", None, BimodalAttackConfig(num_steps=1, **base))
    # early_stop wiring: runs, and stop_flag can only shorten the run
    res = run(model, tok, proc, "tell me", "tell me", "Sure here", None,
              BimodalAttackConfig(num_steps=3, search_width=8, topk=8, early_stop=True, **base), rng_device="cpu")
    assert 1 <= len(res.losses) <= 3
