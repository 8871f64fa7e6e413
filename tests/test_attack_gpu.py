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


def run_case(name, cfg_over=None, **engine):
    """The golden case `name` through the engine; `cfg_over`: config fields changed for runs that are compared with each
    other instead of with the golden (the many-rank rehearsal)."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    m = META["cases"][name]
    model, tok, proc, image = S.tiny_case(m["kind"], device=DEV)
    tmp = tempfile.mkdtemp(prefix="bma_gpu_")
    cfg = BimodalAttackConfig(seed=1, verbosity="ERROR", optim_str_init=m["optim_str_init"], images_folder=tmp,
                              **dict(m["config"], **(cfg_over or {})))
    trace = []
    engine.setdefault("strict", True)        # a fast path that silently gives up on these models is a failure
    res = run(model, tok, proc, m["goal"], m["goal"], m["target"], image, cfg,
              normalize=S.Normalize(S.CLIP_MEAN, S.CLIP_STD), rng_device="cpu", trace=trace, **engine)
    return m, res, trace, tmp


def _ambiguous_tokens(grad_row, allowed, k, tol):
    """Token ids whose rank among the top-(k+1) allowed values is decided by a gap
    smaller than `tol` (the device-vs-CPU gradient discrepancy): either order is right."""
    vals = np.where(allowed, grad_row, np.inf)
    order = np.argsort(vals, kind="stable")[: k + 1]
    v = vals[order]
    close = np.where(np.diff(v) <= tol)[0]
    return set(order[close].tolist()) | set(order[close + 1].tolist())


def check_against_golden(golden_dir, name, m, res, trace, tmp, png=True):
    """Step-by-step comparison with the reference's trajectory.  Two layers:
    (1) in situ, on the engine's OWN intermediate values: the kernels' outputs equal the
        oracle's, exactly -- always required;
    (2) against the golden trajectory: exact, except where the golden token gradient holds
        a near-tie (gap below the measured device-vs-CPU gradient difference) inside the
        top-k -- then the two orders are both right, only the tied tokens may differ, and
        if such a candidate wins the step the comparison ends there."""
    from bimodalattack_amd import synthetic as S
    from oracle import kernels as K
    z = np.load(os.path.join(golden_dir, f"g5_{name}.npz"))
    cfg = m["config"]
    assert len(trace) == m["steps"]
    eps, alpha, k = cfg.get("eps", 0), cfg.get("alpha", 0), cfg.get("topk", 256)
    na = K.nonascii_tokens(S.build_tokenizer(S.TINY_WORDS, S.TINY_NONASCII, S.TINY_UNRT))
    diverged = False
    diverged_at = None           # the step whose near-tied winner differs from the golden's: nothing after it is compared
    for i, st in enumerate(trace):
        # ---- (1) in-situ kernel parity -------------------------------------------------
        if "sampled" in st:
            g = st["grad_tok"][-1]
            assert np.array_equal(st["topk_idx"], K.mask_topk(g, na, k)), f"step {i}: top-k vs oracle"
            assert np.array_equal(st["sampled"], K.sample_scatter(st["optim_ids_in"][0], st["topk_idx"], st["pos"],
                                                                  st["rank"])), f"step {i}: scatter vs oracle"
        if diverged:
            continue
        # ---- (2) the reference's trajectory ---------------------------------------------
        assert st["n_grad"] == int(z[f"s{i}_n_grad"])
        assert np.array_equal(st["optim_ids_in"], z[f"s{i}_optim_ids_in"]), f"step {i}: optim ids"
        tol = 0.0
        if st["grad_tok"]:
            want = z[f"s{i}_grad_tok0"]
            # device-vs-CPU GEMM reduction order: absolute error scales with the row's magnitude
            np.testing.assert_allclose(st["grad_tok"][-1], want, rtol=2e-3, atol=1e-4 * float(np.abs(want).max()))
            tol = 4 * float(np.abs(st["grad_tok"][-1] - want).max())
        for j, g in enumerate(st["grad_img"]):
            wg = z[f"s{i}_grad_img{j}"]
            np.testing.assert_allclose(g, wg, rtol=5e-3, atol=1e-4 * float(np.abs(wg).max()))
        if "image_after_pgd" in st:
            want = z[f"s{i}_image_after_pgd"]
            diff = st["image_after_pgd"] != want
            assert diff.mean() <= 0.005, f"step {i}: {diff.sum()} pixels differ"
            assert np.abs(st["image_after_pgd"] - want).max() <= 2 * alpha * eps + 1e-6
        exact = True
        if "sampled" in st:
            want = z[f"s{i}_sampled"]
            assert st["sampled"].shape == want.shape
            if not np.array_equal(st["sampled"], want):
                exact = False
                allowed = np.ones(z[f"s{i}_grad_tok0"].shape[1], bool)
                allowed[na] = False
                for b, p in zip(*np.where(st["sampled"] != want)):
                    amb = _ambiguous_tokens(z[f"s{i}_grad_tok0"][p], allowed, k, tol)
                    assert st["sampled"][b, p] in amb and want[b, p] in amb, \
                        f"step {i}: candidate {b} position {p} differs outside a gradient near-tie"
        if exact:
            if "filtered" in st:
                assert np.array_equal(st["filtered"], z[f"s{i}_filtered"]), f"step {i}: filter survivors"
            for j, l in enumerate(st["losses"]):
                np.testing.assert_allclose(l, z[f"s{i}_loss{j}"], rtol=1e-4)
        nxt = z[f"s{i + 1}_optim_ids_in"] if i + 1 < m["steps"] else None
        if not exact and nxt is not None and not np.array_equal(trace[i + 1]["optim_ids_in"], nxt):
            diverged = True            # a near-tied candidate won: both continuations are valid
            diverged_at = i
        if not diverged:
            np.testing.assert_allclose(st["current_loss"], z["losses"][i], rtol=1e-4)
    if not diverged:
        np.testing.assert_allclose(res.losses, z["losses"], rtol=1e-4)
        np.testing.assert_allclose(res.best_loss, float(z["best_loss"]), rtol=1e-4)
        assert res.strings == m["strings"] and res.best_string == m["best_string"]
        assert res.adversarial_suffixes == m["adversarial_suffixes"]
    assert [len(getattr(res, k_)) for k_ in ("gradient_times", "sampling_times", "loss_times", "pgd_times",
                                             "total_times")] == m["n_timing"]
    assert res.model_outputs == [""] * m["steps"]
    check_png = png
    png = os.path.join(golden_dir, f"g5_{name}_png0.npz")
    if check_png and os.path.exists(png):
        from PIL import Image
        got = np.array(Image.open(os.path.join(tmp, "0.png")))
        want = np.load(png)["png"]
        assert got.shape == want.shape and (got != want).mean() <= 0.005
        assert sorted(os.listdir(tmp)) == sorted(f"{i}.png" for i in range(m["steps"]))
    _ALL_RUNS.append((name, diverged_at, len(trace)))
    return diverged_at


# how each base trajectory ended on THIS device: None = compared to its last step, i = the escape hatch opened at step i
# (a near-tied candidate won and every later step went unchecked); audited by the test behind the parametrised one
def _audit_line(line: str) -> None:
    """The audit's outcome where a log tail keeps it: as a warning (pytest's summary) and in conftest's terminal summary."""
    import warnings
    import conftest
    conftest.AUDIT_LINES.append(line)
    warnings.warn("golden audit: " + line, UserWarning, stacklevel=2)


_DIVERGED_AT = {}
_ALL_RUNS = []          # (case, diverged_at, steps) of EVERY check_against_golden call of the session (the last test of the file audits it)


@pytest.mark.parametrize("name", sorted(META["cases"]))
def test_trajectory_matches_reference(golden_dir, name):
    m, res, trace, tmp = run_case(name)
    _DIVERGED_AT[name] = check_against_golden(golden_dir, name, m, res, trace, tmp)


def test_most_base_trajectories_are_compared_to_their_last_step():
    """VERDICT r4 "What's weak" 2: `check_against_golden` stops comparing once a near-tied candidate wins a step
    (legitimate: SURVEY.md 7 "argmin flips") -- so count how often that happens on this device.  At least 10 of the 13
    base trajectories must be held to the reference to their LAST step; otherwise the goldens need regenerating with
    margin.  (Runs behind the parametrised test above; skipped when that was deselected.)"""
    if len(_DIVERGED_AT) < len(META["cases"]):
        pytest.skip("the base trajectories were not all run in this session")
    ended_early = {k: v for k, v in _DIVERGED_AT.items() if v is not None}
    line = (f"trajectories compared to their last step: {len(_DIVERGED_AT) - len(ended_early)} of {len(_DIVERGED_AT)}; "
            f"diverged (near-tied winner) at: {ended_early}")
    print(line)
    _audit_line(line)                                  # (the driver's log tail must carry this)
    assert len(_DIVERGED_AT) - len(ended_early) >= min(10, len(_DIVERGED_AT)), ended_early


@pytest.mark.parametrize("name", ["llava_joint", "opt_gcg", "gemma3_joint", "llava_pgd_gcg"])
@pytest.mark.parametrize("engine", [dict(prefix_reuse=False), dict(prefix_reuse=False, target_rows_only=False),
                                    dict(chunk=5),
                                    # the HF-cache path (KV concat) instead of shared-prefix attention
                                    dict(shared_prefix_attention=False),
                                    # shared-prefix attention on the padded block (every suffix token computed)
                                    dict(ragged_suffix=False),
                                    # the reference's separate batch-1 re-score of the joint winner
                                    dict(joint_winner_from_batch=False),
                                    # the full 1-sequence gradient pass instead of scoring prefix + tail (joint mode)
                                    dict(grad_prefix_reuse=False),
                                    # HuggingFace's own decoder-layer forward (aten residual adds, one rotary launch per tensor)
                                    dict(fuse_add_norm=False),
                                    # the host reads each step's outcome before the next gradient pass is launched
                                    dict(gradient_ahead=False),
                                    # ... or launches it ahead but plans the ragged forward from the sampled ids
                                    dict(early_plan=False),
                                    # everything eager and unfused
                                    dict(graph_scoring=False, graph_gradient=False,
                                         fused_elementwise=False, gemm_tuning="off")])
def test_restructurings_do_not_change_results(golden_dir, name, engine):
    """Full-sequence forward / full logits (the reference's call shape) and odd chunk
    sizes give the same trajectory as prefix reuse + target rows only."""
    m, res, trace, tmp = run_case(name, **engine)
    check_against_golden(golden_dir, name, m, res, trace, tmp)


@pytest.mark.parametrize("name", ["llava_gcg_early", "llava_pgd_gcg_early", "llava_joint_early", "gemma3_pgd",
                                  "gemma3_joint_dyn"])
@pytest.mark.parametrize("engine", [dict(graph_scoring=False, graph_gradient=False),
                                    dict(joint_winner_from_batch=False),
                                    dict(fuse_pgd_only=False),
                                    dict(gradient_ahead=False),
                                    dict(early_plan=False),
                                    dict(grad_prefix_reuse=False),
                                    dict(ragged_suffix=False, chunk=7)])
def test_early_stop_and_gemma_orders_under_restructurings(golden_dir, name, engine):
    """early_stop runs that stop mid-run (reference :1300-1306, :785-787) and the Gemma-3 segment orders
    (:1150-1163) give the reference's trajectory whether the winner re-score is a hipGraph replay or eager,
    whether the joint winner's loss comes from the batch or from a re-score, with and without the PGD-only
    fusion, and through odd chunks."""
    m, res, trace, tmp = run_case(name, **engine)
    assert len(res.losses) == m["steps"] < m["config"]["num_steps"] or "early" not in name
    check_against_golden(golden_dir, name, m, res, trace, tmp)


@pytest.mark.parametrize("name", ["opt_gcg", "llava_gcg_early"])
def test_gradient_queued_ahead_gives_the_same_run(name):
    """GCG-only without a trace (nothing but the packed read-back stops the host): the run whose next gradient pass
    is queued behind the scoring forward -- with the ragged plan made from the random draws while that pass runs, or
    from the sampled ids afterwards -- returns what the run that reads every step's outcome first returns --
    losses, strings, the step an early stop ends it at, one timing entry per phase and step."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    m = META["cases"][name]
    out = []
    for ahead, early in ((True, True), (True, False), (False, False)):
        model, tok, proc, image = S.tiny_case(m["kind"], device=DEV)
        cfg = BimodalAttackConfig(seed=1, verbosity="ERROR", optim_str_init=m["optim_str_init"],
                                  images_folder=tempfile.mkdtemp(prefix="bma_gpu_"), **m["config"])
        out.append(run(model, tok, proc, m["goal"], m["goal"], m["target"], image, cfg,
                       normalize=S.Normalize(S.CLIP_MEAN, S.CLIP_STD), rng_device="cpu", strict=True,
                       gradient_ahead=ahead, early_plan=early))
    a = out[0]
    for b in out[1:]:
        # (the ragged plan made from the draws may compute a few rows more than the one made from the ids -- a
        # candidate that happens to equal its parent -- which moves no loss: rows are independent)
        assert a.losses == b.losses and a.strings == b.strings and a.adversarial_suffixes == b.adversarial_suffixes
        assert a.best_loss == b.best_loss and a.best_string == b.best_string
    assert len(a.losses) == m["steps"]
    for r in out:
        assert len(r.gradient_times) == len(r.losses) and len(r.loss_times) == len(r.losses)
        assert len(r.sampling_times) == len(r.losses) and all(t >= 0.0 for t in r.gradient_times + r.loss_times)


def test_early_draws_keep_the_reference_order_of_the_generator():
    """the draws of step i+1 made inside step i (early_plan) must not reorder what the run takes from the
    global generator.  (a) a run that stops early leaves the generator where the plain order leaves it -- the draws of
    the step that never ran are given back; (b) with debug_output and a SAMPLING generation_config, generate(i)
    consumes random numbers between the draws of step i and of step i+1 (:745-777 against :150-160): same losses,
    strings and generated text with the pipelined step boundary as without."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    m = META["cases"]["llava_gcg_early"]
    after = []
    for early in (True, False):
        model, tok, proc, image = S.tiny_case(m["kind"], device=DEV)
        cfg = BimodalAttackConfig(seed=1, verbosity="ERROR", optim_str_init=m["optim_str_init"],
                                  images_folder=tempfile.mkdtemp(prefix="bma_gpu_"), **m["config"])
        res = run(model, tok, proc, m["goal"], m["goal"], m["target"], image, cfg,
                  normalize=S.Normalize(S.CLIP_MEAN, S.CLIP_STD), rng_device="cpu", strict=True, early_plan=early)
        assert len(res.losses) == m["steps"] < m["config"]["num_steps"]          # it did stop early
        after.append((res.losses, torch.rand(4).tolist()))
    assert after[0] == after[1]
    out = []
    for early in (True, False):
        model, tok, proc, _ = S.tiny_case("opt", device=DEV)
        model.generation_config.do_sample = True
        model.generation_config.top_k = 8
        cfg = BimodalAttackConfig(num_steps=12, search_width=8, topk=8, seed=3, verbosity="ERROR", debug_output=True,
                                  optim_str_init=S.TINY_OPTIM_INIT, images_folder=tempfile.mkdtemp())
        res = run(model, tok, proc, "tell me", "tell me", "Sure here", None, cfg, rng_device="cpu", strict=True,
                  early_plan=early)
        out.append((res.losses, res.strings, res.model_outputs, torch.rand(4).tolist()))
    assert out[0] == out[1]
    assert out[0][2][0] != "" or out[0][2][10] != ""


@pytest.mark.parametrize("kind,dtype", [("llava", torch.bfloat16), ("llava", torch.float16),
                                        ("gemma3", torch.bfloat16)])
def test_device_rng_mode_and_16bit_run(kind, dtype):
    """Default mode draws the randoms on the device like the reference does on a GPU; a
    bf16 / fp16 model exercises the 16-bit kernels end to end.  No CPU golden exists for either,
    so: same seed -> same run; losses finite; candidates differ from the parent in
    exactly n_replace positions; every top-k id is allowed."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    out = []
    for _ in range(2):
        model, tok, proc, image = S.tiny_case(kind, dtype=dtype, device=DEV)
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
    for st, st1 in zip(t0, t1):
        assert np.array_equal(st["sampled"], st1["sampled"])
        assert ((st["sampled"] != st["optim_ids_in"]).sum(1) <= 1).all()
        assert not (set(st["topk_idx"].reshape(-1).tolist()) & na)


def test_gemm_shapes_are_touched_before_the_first_step():
    """EngineOptions.warm_gemms: a GCG run on a llama-family model touches, in set-up, every row count its ragged forwards
    can meet (so that the library's first-use loading of a kernel does not land inside a step); the run itself is the
    one without the warm-up."""
    from bimodalattack_amd import BimodalAttackConfig, synthetic as S
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import expected_row_counts
    out = {}
    for warm in (True, False):
        model, tok, proc, _ = S.tiny_case("llava", dtype=torch.bfloat16, device=DEV)
        cfg = BimodalAttackConfig(num_steps=3, search_width=24, topk=16, seed=5, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT,
                                  images_folder=tempfile.mkdtemp())
        atk = BimodalAttack(model, tok, proc, cfg, None, EngineOptions.from_env(rng_device="cpu", strict=True, warm_gemms=warm))
        res = atk.run("tell me a story", "tell me a story", "Sure here is a story", None)
        out[warm] = (res.losses, res.strings, atk.engine_state()["warmed_row_counts"], atk)
    assert out[True][:2] == out[False][:2] and out[False][2] == []
    atk = out[True][3]
    n_opt = len(tok(S.TINY_OPTIM_INIT, add_special_tokens=False)["input_ids"])
    L = n_opt + atk.seg["after"].shape[1] + atk.seg["target_in"].shape[1]
    assert out[True][2] == expected_row_counts(24, n_opt, L, 1, 16, (1,)) and len(out[True][2]) >= 1
    assert atk.score_stats["ragged_calls"] == 3


def test_early_stop_and_errors():
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    model, tok, proc, _ = S.tiny_case("opt", device=DEV)
    base = dict(seed=1, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT, images_folder=tempfile.mkdtemp())
    with pytest.raises(TypeError):
        run(model, tok, proc, "a", "a", "Sure", None, BimodalAttackConfig(pgd_after_gcg=True, **base))
    with pytest.raises(ValueError, match="needs an image"):
        run(model, tok, proc, "a", "a", "Sure", None, BimodalAttackConfig(pgd_attack=True, **base))
    cpu_model, _, _, _ = S.tiny_case("opt")
    with pytest.raises(RuntimeError, match="AMD GPU only"):
        run(cpu_model, tok, proc, "a", "a", "Sure", None, BimodalAttackConfig(num_steps=1, **base))
    # early_stop wiring: runs, and stop_flag can only shorten the run
    res = run(model, tok, proc, "tell me", "tell me", "Sure here", None,
              BimodalAttackConfig(num_steps=3, search_width=8, topk=8, early_stop=True, **base), rng_device="cpu")
    assert 1 <= len(res.losses) <= 3


def test_oom_halving_recovers_and_remembers(golden_dir):
    """The out-of-memory safety net (reference utils.py:57-115: halve and retry), exercised for real: every
    scoring forward with more than 5 candidates raises the allocator's error text.  The engine halves until
    a chunk fits -- leaving ragged mode on the way --, remembers the size for the following steps (the
    reference restarts from the full batch every step), and the trajectory is still the reference's."""
    from bimodalattack_amd import BimodalAttackConfig, synthetic as S
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    name = "llava_joint"
    m = META["cases"][name]
    model, tok, proc, image = S.tiny_case(m["kind"], device=DEV)
    tmp = tempfile.mkdtemp(prefix="bma_gpu_")
    cfg = BimodalAttackConfig(seed=1, verbosity="ERROR", optim_str_init=m["optim_str_init"], images_folder=tmp, **m["config"])
    trace = []
    atk = BimodalAttack(model, tok, proc, cfg, S.Normalize(S.CLIP_MEAN, S.CLIP_STD),
                        EngineOptions.from_env(rng_device="cpu", trace=trace, strict=True))
    raised = []

    def guard(fn, count):
        def wrapped(x, *a, **kw):
            n = count(x, *a)
            if n > 5:
                raised.append(n)
                raise RuntimeError("HIP out of memory. Tried to allocate 2.00 GiB (test)")
            return fn(x, *a, **kw)
        return wrapped

    hf = atk.hf
    hf.target_logits = guard(hf.target_logits, lambda x, *a: x.shape[0])
    hf.target_logits_shared_prefix = guard(hf.target_logits_shared_prefix, lambda x, *a: x.shape[0])
    hf.target_logits_ragged = guard(hf.target_logits_ragged, lambda rows, T, cache, maps: maps.m_out)
    res = atk.run(m["goal"], m["goal"], m["target"], image)
    assert raised and raised[0] > 5 and atk._chunk_cap is not None and atk._chunk_cap <= 5
    assert len([r for r in raised]) <= 4                      # halved a few times in the FIRST step, never again
    check_against_golden(golden_dir, name, m, res, trace, tmp)


@pytest.mark.parametrize("mode", ["gcg", "joint"])
def test_filter_first_policy_with_a_tokenizer_that_rejects_a_third(mode):
    """VERDICT r4 item 5 / "What's missing" 3: the reference filters BEFORE it scores (:166-186, :930-941).  With a
    tokenizer whose round trip rejects a real share of the candidates (80 of 256 words contain a space: ~30 % of the
    sampled candidates fail) the engine must not spend the scoring phase on them.  Three runs of one attack on the tiny
    LLaVA (fp32, CPU draws): filter beside the forward + mask afterwards (`filter_first=False`), the reference's order
    (`True`), and the default policy (None: score-everything at the first step, filter-first from then on because the
    survivor rate sits far below the break-even).  Same sampled ids, survivors, losses, strings in all three -- and in
    the oracle loop, which filters first like the reference -- while the candidates actually scored drop by the rejected
    share."""
    from bimodalattack_amd import BimodalAttackConfig, synthetic as S
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from oracle.attack_loop import run_oracle
    joint = mode == "joint"
    steps = 5
    kw = dict(num_steps=steps, search_width=32, topk=64, seed=3, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT,
              pgd_attack=joint, gcg_attack=True, joint_eval=joint, eps=64 / 255, alpha=4 / 255)
    norm = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)
    n_unrt = 80
    out = {}
    for ff in (False, True, None):
        model, _, proc, image = S.tiny_case("llava", device=DEV)
        tok = S.build_tokenizer(S.TINY_WORDS, S.TINY_NONASCII, n_unrt)
        proc = S.SyntheticProcessor(tok)
        trace = []
        atk = BimodalAttack(model, tok, proc, BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw), norm,
                            EngineOptions.from_env(rng_device="cpu", trace=trace, strict=True, filter_first=ff, save_images=False))
        res = atk.run("tell me a story", "tell me a story", "Sure here is", image if joint else None)
        assert not atk.fallbacks, atk.fallbacks
        out[ff] = (res, trace, atk.score_stats["candidates"], list(atk.filter_first_steps), list(atk._keep_rates))
    cmodel, _, _, cimage = S.tiny_case("llava")
    ctok = S.build_tokenizer(S.TINY_WORDS, S.TINY_NONASCII, n_unrt)
    want, wtrace, _ = run_oracle(cmodel, ctok, S.SyntheticProcessor(ctok), "tell me a story", "tell me a story", "Sure here is",
                                 cimage if joint else None, BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw), normalize=norm)
    base = out[False]
    rates = base[4]
    assert len(rates) == steps and 0.4 < sum(rates) / steps < 0.9, rates           # a real share is rejected every step
    for ff, (res, trace, scored, first_steps, _) in out.items():
        for a, b, c in zip(trace, base[1], wtrace):
            assert np.array_equal(a["sampled"], b["sampled"]) and np.array_equal(a["sampled"], c["sampled"])
            assert np.array_equal(a["filtered"], b["filtered"]) and np.array_equal(a["filtered"], c["filtered"])
            np.testing.assert_allclose(a["losses"][0], b["losses"][0], rtol=1e-5)
            np.testing.assert_allclose(a["losses"][0], c["losses"][0], rtol=1e-4)
        assert res.strings == base[0].strings == want["strings"]
        np.testing.assert_allclose(res.losses, base[0].losses, rtol=1e-5)
        assert [st["n_scored"] for st in trace] == [st["n_scored"] for st in base[1]]
    sampled_total = sum(st["sampled"].shape[0] for st in base[1])
    kept_total = sum(st["n_scored"] for st in base[1])

    assert out[False][3] == [] and out[True][3] == list(range(steps)) and out[None][3] == list(range(1, steps))
    assert out[False][2] == sampled_total and out[True][2] == kept_total, (out[False][2], out[True][2], sampled_total, kept_total)
    assert kept_total < out[None][2] < sampled_total and kept_total < 0.9 * sampled_total


def test_long_suffix_matches_oracle():
    """A 70-token suffix (the reference takes any optim_str_init; round 1's position sampler stopped at 64):
    two GCG steps on the tiny OPT, HIP engine against the oracle loop, same CPU draws."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    from oracle.attack_loop import run_oracle
    init = " ".join(["x", "y", "z", "w", "!"] * 14)
    kw = dict(num_steps=2, search_width=16, topk=8, n_replace=3, seed=1, verbosity="ERROR", optim_str_init=init)
    model, tok, proc, _ = S.tiny_case("opt", device=DEV)
    trace = []
    res = run(model, tok, proc, "tell me", "tell me", "Sure here", None,
              BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw), rng_device="cpu", trace=trace, strict=True)
    cmodel, ctok, cproc, _ = S.tiny_case("opt")
    want, wtrace, _ = run_oracle(cmodel, ctok, cproc, "tell me", "tell me", "Sure here", None,
                                 BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw))
    assert trace[0]["sampled"].shape == (16, 70)
    for a, b in zip(trace, wtrace):
        assert np.array_equal(a["sampled"], b["sampled"]) and np.array_equal(a["filtered"], b["filtered"])
        np.testing.assert_allclose(a["losses"][0], b["losses"][0], rtol=1e-4)
    assert res.strings == want["strings"]


@pytest.mark.parametrize("name", ["llava_joint_dyn", "gemma3_joint_dyn"])
def test_hf_cache_route_with_padded_last_chunk(golden_dir, name):
    """Decaying widths that are not a multiple of the chunk quantum (24, 18, 12, 8; 24, 19, 14, 9, 8) through the
    HF-cache route (KV concat, shared_prefix_attention=False): the last chunk is padded to the quantum and the
    prefix cache must be expanded to the PADDED row count (it was expanded to b and the concat failed)."""
    m, res, trace, tmp = run_case(name, shared_prefix_attention=False)
    check_against_golden(golden_dir, name, m, res, trace, tmp)


def test_hf_cache_route_odd_width_matches_oracle():
    """A model outside the shared-prefix families (OPT) at search_width 13: 13 > quantum 8 and 13 % 8 != 0, so the
    one chunk is padded to 16 rows on the HF-cache route; against the oracle loop, same CPU draws."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    from oracle.attack_loop import run_oracle
    kw = dict(num_steps=3, search_width=13, topk=8, seed=1, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT)
    model, tok, proc, _ = S.tiny_case("opt", device=DEV)
    trace = []
    res = run(model, tok, proc, "tell me", "tell me", "Sure here", None,
              BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw), rng_device="cpu", trace=trace, strict=True)
    cmodel, ctok, cproc, _ = S.tiny_case("opt")
    want, wtrace, _ = run_oracle(cmodel, ctok, cproc, "tell me", "tell me", "Sure here", None,
                                 BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw))
    assert trace[0]["sampled"].shape[0] == 13
    for a, b in zip(trace, wtrace):
        assert np.array_equal(a["sampled"], b["sampled"]) and np.array_equal(a["filtered"], b["filtered"])
        np.testing.assert_allclose(a["losses"][0], b["losses"][0], rtol=1e-4)
    assert res.strings == want["strings"]


def test_library_ragged_route_on_a_16bit_model(monkeypatch):
    """BMA_FUSED_RAGGED_ATTENTION=0 (the documented A/B switch) on a bf16 model: the planner builds the padded-block
    maps the library attention route needs (it decided from dtype/head size alone, the route then raised
    and ragged scoring was disabled), ragged scoring stays ON, and the losses agree with the kernel route."""
    from bimodalattack_amd import BimodalAttackConfig, prefix_attention as pa, synthetic as S
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    out = {}
    for fused in (True, False):
        monkeypatch.setattr(pa, "FUSED_RAGGED_ATTENTION", fused)
        model, tok, proc, image = S.tiny_case("llava", dtype=torch.bfloat16, device=DEV)
        trace = []
        cfg = BimodalAttackConfig(num_steps=2, search_width=24, topk=16, pgd_attack=False, gcg_attack=True, seed=5,
                                  verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT, images_folder=tempfile.mkdtemp())
        atk = BimodalAttack(model, tok, proc, cfg, None, EngineOptions.from_env(rng_device="cpu", trace=trace, strict=True,
                                                                                  loss_in_model_dtype=False))
        atk.run("tell me a story", "tell me a story", "Sure here is a story", None)
        assert atk.fallbacks == {} and atk.hf.ragged_ok is True and atk.score_stats["ragged_calls"] == 2
        out[fused] = trace
    a, b = out[True][0], out[False][0]
    assert np.array_equal(a["sampled"], b["sampled"])
    np.testing.assert_allclose(a["losses"][0], b["losses"][0], rtol=2e-2)      # bf16: two attention implementations


def test_list_form_optim_str_init_matches_oracle():
    """optim_str_init as a LIST of strings (reference :857-866: one buffer entry per string, a warning when the count
    differs from buffer_size): HIP engine against the oracle loop."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    from oracle.attack_loop import run_oracle
    inits = ["x x x x x x x x", "y x y x y x y x", "! ! x x ! ! x x"]
    kw = dict(num_steps=2, search_width=16, topk=8, buffer_size=3, seed=1, verbosity="ERROR", optim_str_init=inits,
              pgd_attack=True, gcg_attack=True, joint_eval=True, eps=64 / 255, alpha=4 / 255)
    norm = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)
    model, tok, proc, image = S.tiny_case("llava", device=DEV)
    trace = []
    res = run(model, tok, proc, "tell me", "tell me", "Sure here", image,
              BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw), normalize=norm, rng_device="cpu", trace=trace,
              strict=True)
    cmodel, ctok, cproc, cimage = S.tiny_case("llava")
    want, wtrace, orc = run_oracle(cmodel, ctok, cproc, "tell me", "tell me", "Sure here", cimage,
                                   BimodalAttackConfig(images_folder=tempfile.mkdtemp(), **kw), normalize=norm)
    for a, b in zip(trace, wtrace):
        assert np.array_equal(a["optim_ids_in"], b["optim_ids_in"]) and np.array_equal(a["sampled"], b["sampled"])
        np.testing.assert_allclose(a["losses"][0], b["losses"][0], rtol=1e-4)
    np.testing.assert_allclose(res.losses, want["losses"], rtol=1e-4)
    assert res.strings == want["strings"]


def test_debug_output_generates_every_tenth_step():
    """debug_output=True (reference :745-777): a greedy generation from the first candidate's prompt at steps 0, 10, ...;
    model_outputs carries the decoded text there and "" elsewhere; the trajectory is the one without it."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    norm = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)
    out = {}
    for dbg in (False, True):
        model, tok, proc, image = S.tiny_case("llava", device=DEV)
        cfg = BimodalAttackConfig(num_steps=11, search_width=8, topk=8, seed=2, verbosity="ERROR", debug_output=dbg,
                                  pgd_attack=True, gcg_attack=True, joint_eval=True, eps=64 / 255, alpha=4 / 255,
                                  optim_str_init=S.TINY_OPTIM_INIT, images_folder=tempfile.mkdtemp())
        out[dbg] = run(model, tok, proc, "tell me", "tell me", "Sure here", image, cfg, normalize=norm, rng_device="cpu",
                       strict=True)
    assert out[True].losses == out[False].losses and out[True].strings == out[False].strings
    mo = out[True].model_outputs
    assert len(mo) == 11 and all(isinstance(t, str) for t in mo)
    assert all(mo[i] == "" for i in range(11) if i % 10) and out[False].model_outputs == [""] * 11


def test_weights_changed_between_runs_are_picked_up():
    """The engine keeps derived copies of the decoder weights per MODEL (transposed, concatenated q/k/v, interleaved
    gate/up: fused._CopyCache).  A caller who changes the weights between two run() calls -- in place -- must get the
    attack on the NEW weights: every copy is stamped with its sources' version counters and rebuilt when they moved."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    norm = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)

    def attack(model, tok, proc, image):
        cfg = BimodalAttackConfig(num_steps=2, search_width=16, topk=8, seed=4, verbosity="ERROR", pgd_attack=True,
                                  gcg_attack=True, joint_eval=True, eps=64 / 255, alpha=4 / 255,
                                  optim_str_init=S.TINY_OPTIM_INIT, images_folder=tempfile.mkdtemp())
        return run(model, tok, proc, "tell me", "tell me", "Sure here", image.clone(), cfg, normalize=norm, rng_device="cpu",
                   strict=True)

    def perturb(model):
        g = torch.Generator(device=DEV).manual_seed(99)
        with torch.no_grad():
            for name, p in model.named_parameters():
                if p.dim() == 2 and "proj" in name:
                    p.mul_(1.0 + 0.2 * torch.rand(p.shape, generator=g, device=DEV).to(p.dtype))

    model, tok, proc, image = S.tiny_case("llava", dtype=torch.bfloat16, device=DEV)
    first = attack(model, tok, proc, image)
    perturb(model)
    second = attack(model, tok, proc, image)
    fresh, tok2, proc2, image2 = S.tiny_case("llava", dtype=torch.bfloat16, device=DEV)
    perturb(fresh)
    want = attack(fresh, tok2, proc2, image2)
    assert second.losses == want.losses and second.strings == want.strings
    assert first.losses != second.losses


# ------------------------------------------------------------------ sharded engine, 2 ranks on one GPU
def _sharded_worker(rank, world, port, name, out, backend="gloo", cfg_over=None, engine=None):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if backend == "nccl":                                                 # RCCL: one GPU per rank
        global DEV
        DEV = f"cuda:{rank}"
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(DEV))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)      # gloo moves the GPU tensors through the host
    try:
        if backend != "nccl":
            torch.cuda.set_device(0)
        m, res, trace, tmp = run_case(name, cfg_over, **(engine or {}))
        out.put((rank, res.losses, res.strings, [st["n_scored"] for st in trace],
                 [st["losses"][0].tolist() for st in trace if st["losses"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["llava_joint", "llava_joint_dyn", "opt_gcg", "llava_pgd_gcg_early", "gemma3_joint_dyn"])
def test_sharded_engine_two_ranks_equal_single(golden_dir, name):
    """Candidate scoring sharded over 2 ranks (rehearsed with gloo, both ranks on cuda:0 --
    the driver's 8-GPU run uses RCCL): every rank returns the single-process result."""
    import socket
    import torch.multiprocessing as mp
    m, res1, trace1, _ = run_case(name)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, losses, strings, n_scored, cand_losses in got:
        assert strings == res1.strings, f"rank {rank}"
        np.testing.assert_allclose(losses, res1.losses, rtol=1e-5)
        assert n_scored == [st["n_scored"] for st in trace1]
        for a, b in zip(cand_losses, [st["losses"][0] for st in trace1 if st["losses"]]):
            np.testing.assert_allclose(a, b, rtol=1e-5)
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2]          # ranks agree bit for bit


def _collectives_worker(rank, world, port, name, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        m, res, trace, tmp = run_case(name)
        out.put((rank, [st["collectives"] for st in trace] + [trace[-1]["collectives_end"]]))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name,per_step", [
    # GCG-only: rank 0's sampled ids in one broadcast + one gather of the losses (the early-stop hits ride in it) + the NEXT
    # step's random draws, made ahead of the gradient pass (early_plan: not behind the last step of the schedule)
    ("llava_gcg", [3, 3, 3, 2]), ("llava_gcg_early", [3, 3, 3, 3, 3]),
    # joint: ids + PGD image in ONE packed broadcast, the gather, the early draws
    ("llava_joint", [3, 3, 2]), ("llava_joint_early", [3, 3, 3]),
    # Gemma-3 (suffix in front of the image: padded blocks, nothing is planned ahead): the two and nothing else
    ("gemma3_joint", [2, 2, 2]),
    # PGD+GCG, candidates scored without the image: the image goes out on its own in front of the second gradient pass (which
    # caches what it derives from the image by tensor identity), then the ids, then the gather
    ("llava_pgd_gcg", [3, 3, 3]),
    # PGD-only: the image, nothing to gather
    ("llava_pgd", [1, 1, 1, 1]),
])
def test_sharded_engine_issues_exactly_its_collectives_per_step(name, per_step):
    """at world 2 (gloo, both ranks on the one GPU) every mode issues EXACTLY the data-path collectives
    DESIGN.md 8 lists -- two per step (one packed broadcast of rank 0's state, one all-gather of the losses) plus the early
    draws' broadcast where the engine plans ahead -- counted by dist.CandidateSharder.n_collectives at every step's start;
    one more in front of the loop (the initial suffix's loss).  A collective that crept into the step loop would scale with
    the 8-GPU run's 600 steps; this holds the count."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_collectives_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, counts in got:
        assert counts[0] == 1, (rank, counts)                              # the initial suffix's loss gather (image broadcast for PGD-only)
        assert [b - a for a, b in zip(counts[:-1], counts[1:])] == per_step, (rank, counts)


# Ranks of the many-rank rehearsal.  The driver's node has 8; a GPU box of this pool lets at most 6 processes use its one
# card at a time and the test runner is one of them, so the rehearsal runs 4 ranks on a width that decays BELOW 4: the
# same code paths as 8 ranks on a width decayed to 8 and thinned by the filter -- `per` = 1, empty
# shares, the contiguous partition taking over from the dealt one.  The 8-rank partition / gather arithmetic itself runs
# on the CPU (tests/test_dist_gloo.py::test_sharded_scoring_gloo[8]).
MANY_RANKS = 4


@pytest.mark.parametrize("name,engine", [("llava_joint_dyn", {}), ("gemma3_joint_dyn", {}), ("llava_gcg", {}),
                                         # ... and with the filter run BEFORE scoring on every rank (the reference's order): the
                                         # partition is then made over the survivors, not over the sampled candidates
                                         ("llava_joint_dyn", {"filter_first": True}), ("llava_gcg", {"filter_first": True})])
def test_sharded_engine_more_ranks_than_candidates(name, engine):
    """BASELINE configs[3]/[4]'s tail in miniature: a dynamic width that decays to fewer candidates than there are ranks
    (12, 10, 8, 6, 4, 2 over 4 ranks; the retokenisation filter thins it further): every rank returns the single-process
    run -- strings, losses, candidates scored per step, every candidate's loss."""
    import socket
    import torch.multiprocessing as mp
    over = dict(num_steps=6, search_width=12, dynamic_search=True, min_search_width=2, early_stop=False)
    m, res1, trace1, _ = run_case(name, over)          # (the single process runs the default policy: same winners either way)
    widths = [st["sampled"].shape[0] for st in trace1]
    assert widths == [12, 10, 8, 6, 4, 2] and min(st["n_scored"] for st in trace1) <= 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, MANY_RANKS, port, name, out, "gloo", over, engine)) for r in range(MANY_RANKS)]
    for p in procs:
        p.start()
    got = [out.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, losses, strings, n_scored, cand_losses in got:
        assert strings == res1.strings, f"rank {rank}"
        np.testing.assert_allclose(losses, res1.losses, rtol=1e-5)
        assert n_scored == [st["n_scored"] for st in trace1]
        for a, b in zip(cand_losses, [st["losses"][0] for st in trace1 if st["losses"]]):
            np.testing.assert_allclose(a, b, rtol=1e-5)
    assert all(g[1] == got[0][1] and g[2] == got[0][2] for g in got)      # ranks agree bit for bit


def _tp_worker(rank, world, port, name, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        golden_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
        m, res, trace, tmp = run_case(name, tp_gradient=True)
        check_against_golden(golden_dir, name, m, res, trace, tmp, png=rank == 0)     # (only rank 0 writes the images)
        out.put((rank, res.losses, res.strings))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["llava_joint", "llava_pgd_gcg", "gemma3_joint"])
def test_sharded_engine_with_tensor_parallel_gradient_pass(golden_dir, name):
    """EngineOptions.tp_gradient on two ranks (gloo, both on cuda:0): the batch-1 gradient pass is cut over the ranks --
    each computes its heads / its share of the MLP width, two all-reduces per layer and direction -- while candidate
    scoring is sharded as before; every rank reproduces the reference trajectory (token and pixel gradients to the
    goldens' tolerance, ids, survivors, losses) and the ranks agree bit for bit."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_tp_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0][1] == got[1][1] and got[0][2] == got[1][2]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL cannot put two ranks on one)")
@pytest.mark.parametrize("name", ["llava_joint", "llava_pgd_gcg_early"])
def test_sharded_engine_two_ranks_rccl(golden_dir, name):
    """The same equality over RCCL/xGMI, one GPU per rank (skipped on a one-GPU box)."""
    import socket
    import torch.multiprocessing as mp
    m, res1, trace1, _ = run_case(name)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_sharded_worker, args=(r, 2, port, name, out, "nccl")) for r in range(2)]
    for p in procs:
        p.start()
    got = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, losses, strings, n_scored, cand_losses in got:
        assert strings == res1.strings, f"rank {rank}"
        np.testing.assert_allclose(losses, res1.losses, rtol=1e-5)
    assert got[0][2] == got[1][2]


def test_bench_self_launch_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` as the driver calls it (no launcher): the script starts its two ranks itself
    (rehearsed with gloo on the one GPU of this box; the driver's node runs RCCL), prints ONE JSON line with
    n_gpus = 2, candidates sharded, two data-path collectives per step plus the early draws' broadcast."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BMA_DIST_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2",
                        "--warmup", "1", "--profile-steps", "1", "--search-width", "64", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0] == r.stdout.strip().splitlines()[-1]
    # the driver keeps the last 8000 characters of stdout: the line, `rccl` block included, must fit and be strict JSON
    assert len(lines[0]) <= 6000, len(lines[0])
    d = json.loads(lines[0], parse_constant=lambda c: pytest.fail(f"non-JSON constant {c} in the bench line"))
    assert d["finite"] is True and d["rccl"]["world"] == 2 and d["rccl"]["backend"] == "gloo"
    assert d["detail_file"] and os.path.exists(os.path.join(repo, d["detail_file"]))
    assert d["n_gpus"] == 2 and d["config"]["sharding"] == "candidates/2" and d["scaling"] == "strong"
    # the line carries its own A/B of the tensor-parallel gradient pass: both legs ran, one was chosen
    rc = d["rccl"]
    assert rc["tp_off_ms"] > 0 and rc["tp_on_ms"] > 0 and rc["chosen"] in ("off", "on") and "tp_error" not in rc, rc
    assert rc.get("tp_graph") is False                     # (gloo: its collectives run on the host, nothing to capture)
    assert d["ms_per_step"] == pytest.approx(min(rc["tp_off_ms"], rc["tp_on_ms"]) if rc["chosen"] == "on" else rc["tp_off_ms"], rel=1e-3)
    if rc["chosen"] == "off":
        # one loss gather for the initial suffix, then two collectives per step (ids broadcast + loss gather), 4 steps;
        # from the second step on the draws made ahead of the gradient pass are broadcast as well (early_plan: 8 KB)
        assert d["engine"]["collectives"] == 1 + 2 * (1 + 2 + 1) + 3
    assert not d["engine"]["fallbacks"]
    assert d["roofline"]["bound"] in ("mfma", "hbm") and d["value"] > 0
    # north_star's table for this N without post-processing: steps/s and candidates/s on N GPUs, the run's
    # OWN one-GPU leg (every rank ran the whole job by itself, no collectives), the efficiency against it, the dominant
    # kernel's roofline fraction on every rank
    t = d["scaling_table"]
    assert t["n_gpus"] == 2 and t["candidate_forwards_per_sec"] == pytest.approx(d["value"]) and t["attack_steps_per_sec"] > 0
    n1 = t["own_n1_leg"]
    assert n1["finite"] is True and n1["candidate_forwards_per_sec"] > 0 and n1["steps"] == 2
    assert len(t["own_n1_leg_candidate_forwards_per_sec_per_rank"]) == 2 and all(v > 0 for v in t["own_n1_leg_candidate_forwards_per_sec_per_rank"])
    assert t["efficiency_vs_own_n1"] == pytest.approx(d["value"] / (2 * n1["candidate_forwards_per_sec"]), rel=1e-3)
    assert len(t["dominant_kernel"]["frac_per_rank"]) == 2 and all(f is not None and f > 0 for f in t["dominant_kernel"]["frac_per_rank"])


def test_bench_prints_the_first_leg_when_the_tensor_parallel_leg_does_not_return():
    """bench.py's multi-GPU A/B must never cost the run its line: with the tensor-parallel leg's time limit set to a
    millisecond (as if a collective never returned) the watchdog prints the FIRST leg's complete line -- n_gpus 2, the
    replicated pass's figures, `rccl.tp_note` saying what happened -- and every rank leaves with bench.TP_HUNG_STATUS (4),
    never 0: a hung collective is a deadlock the launcher must see; "status 4 + one JSON line" = the replicated
    leg is valid, the tensor-parallel leg hung."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BMA_DIST_BACKEND="gloo", BMA_TP_LEG_LIMIT_S="0.001")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2",
                        "--warmup", "1", "--profile-steps", "1", "--search-width", "64", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and r.returncode != 0, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["finite"] is True and d["value"] > 0
    rc = d["rccl"]
    assert rc["chosen"] == "off" and rc["tp_on_ms"] is None and rc["tp_off_ms"] == pytest.approx(d["ms_per_step"], rel=1e-3)
    assert "did not finish" in rc["tp_note"]


def test_bench_line_survives_rank_0_dying_inside_the_tensor_parallel_leg():
    """... and when rank 0 is KILLED inside that leg (what a faulting collective does -- no Python handler runs), the child
    process that held the first leg's line prints it: one line on stdout, complete, saying what happened; the launcher's
    exit status is the crash's."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BMA_DIST_BACKEND="gloo", BMA_BENCH_TP_CRASH="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2", "--layers", "2", "--steps", "2",
                        "--warmup", "1", "--profile-steps", "1", "--search-width", "64", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert r.returncode != 0 and len(lines) == 1, (r.returncode, r.stdout[-500:], r.stderr[-2000:])
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["finite"] is True and d["value"] > 0 and d["rccl"]["chosen"] == "off"
    assert "process ended inside the tensor-parallel leg" in d["rccl"]["tp_note"]


# ------------------------------------------------------------------ BASELINE-size parity of the scoring path
def _reference_call_shape_losses(model, atk, cand, order, feats, chunk=8):
    """What the reference computes for these candidates (:1112-1225, :1278-1310): emb(ids) + repeated
    segments concatenated, a full-sequence forward, full (B,S,V) logits, torch cross-entropy in fp32."""
    E = atk.embedding_layer
    want = []
    n = cand.shape[0]
    for s in range(0, n, chunk):
        b = min(chunk, n - s)
        parts = [E(cand[s:s + b]) if nm == "optim" else (feats.to(E.weight.dtype) if nm == "image" else atk.seg[nm]).expand(b, -1, -1)
                 for nm in order]
        x = torch.cat(parts, dim=1)
        logits = model(inputs_embeds=x, use_cache=False).logits
        T = atk.T
        sl = logits[:, x.shape[1] - T - 1:-1, :].float()
        l = torch.nn.functional.cross_entropy(sl.reshape(-1, sl.shape[-1]), atk.labels.repeat(b), reduction="none")
        want.append(l.view(b, T).mean(-1))
        del logits, sl, x
    return torch.cat(want).cpu().numpy()


@pytest.mark.parametrize("workload,n", [("gcg", 48), ("joint", 48), ("gcg", 512), ("joint", 512)])
def test_7b_scoring_equals_reference_call_shape(workload, n):
    """LLaVA-1.5-7B shape, bf16.  The optimised scoring path (target rows only, last token
    dropped, shared-prefix keys/values or shared-prefix attention, fused RMSNorm/SwiGLU/RoPE,
    tuned GEMM selection, HIP splice + CE) against the reference's call shape on the same
    model: full-sequence forward, full (B,S,V) logits, torch cross-entropy in fp32.  Both are
    bf16 computations of the same function; they differ by bf16 rounding noise only.
    n = 512 is BASELINE's search_width in ONE chunk, candidates drawn like the sampler draws them
    (a position, one of 256 tokens for it): ~26 exact duplicates, the coarse row grid and the
    `keep` maps of ragged scoring are asserted at full width."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import segment_order

    dev = torch.device(DEV)
    model, tok, proc, messages, goal, target, image, norm = build_plugins(workload, dev, torch.bfloat16, 32)
    joint = workload == "joint"
    cfg = BimodalAttackConfig(num_steps=1, search_width=n, seed=1, verbosity="ERROR", pgd_attack=joint,
                              gcg_attack=True, joint_eval=joint, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, loss_in_model_dtype=False,
                                                                            strict=True))
    atk._prepare_prompt(messages, target)
    g = torch.Generator(device=DEV).manual_seed(0)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    n_opt = ids.shape[1]
    cand = ids.repeat(n, 1)
    pos = torch.randint(0, n_opt, (n,), generator=g, device=DEV)
    pool = torch.randint(5, 32000, (n_opt, 256), generator=g, device=DEV)      # a "top-k" table per position
    cand[torch.arange(n, device=DEV), pos] = pool[pos, torch.randint(0, 256, (n,), generator=g, device=DEV)]
    order = segment_order("pgd", "llava", single=True) if joint else segment_order("gcg", "llava", no_joint_eval=True)
    with torch.no_grad():
        feats = atk.hf.image_features(image) if joint else None
        cand[7] = cand[3]                                   # an exact duplicate and a copy of the parent
        cand[11] = ids[0]
        distinct = len(np.unique(cand.cpu().numpy(), axis=0))
        got = atk.score_candidates(cand.contiguous(), order, feats).float().cpu().numpy()
        # the same candidates through ragged scoring (rows from the first replaced position on, duplicates
        # once, attention in the one-launch MFMA kernel): what the attack loop runs
        ragged = atk.score_candidates(cand.contiguous(), order, feats, parent=ids).float().cpu().numpy()
        st = atk.score_stats
        assert st["ragged_calls"] == 1 and st["padded_calls"] == 1 and not atk.fallbacks
        L = n_opt + 25
        assert st["rows_needed"] < 2 * n * L * 0.95
        # ragged leg alone: exactly the rows of the distinct candidates from their first replaced position on
        first = np.where((cand != ids).any(1).cpu().numpy(), (cand != ids).int().argmax(1).cpu().numpy(), n_opt - 1)
        _, keep_first = np.unique(cand.cpu().numpy(), axis=0, return_index=True)
        need = n_opt + int((L - first[keep_first]).sum())
        assert st["rows_needed"] - n * L == need, (st["rows_needed"] - n * L, need)
        if n == 512:
            assert distinct < n - 5                          # the draw has duplicates to remove
            assert (st["rows"] - n * L) % 256 == 0 and 0 <= (st["rows"] - n * L) - need < 256     # coarse grid
        want = _reference_call_shape_losses(model, atk, cand, order, feats)
        # the yardstick for "bf16 rounding noise" on THIS model, measured instead of assumed: the same reference call
        # shape in fp32 on the same weights (first 48 candidates).  The engine -- prefix reuse, ragged rows, the two MFMA
        # attention kernels, fused elementwise kernels -- must be as close to fp32 as the reference's own bf16 forward is.
        ny = min(n, 48)
        segs16 = atk.seg
        model.float()
        atk.seg = {k_: v_.float() for k_, v_ in segs16.items()}
        want32 = _reference_call_shape_losses(model, atk, cand[:ny], order, None if feats is None else feats.float())
        atk.seg = segs16
    rel = np.abs(got - want) / np.abs(want)
    rel_r = np.abs(ragged - want) / np.abs(want)
    noise = (np.abs(want[:ny] - want32) / np.abs(want32)).max()
    err = (np.abs(got[:ny] - want32) / np.abs(want32)).max()
    err_r = (np.abs(ragged[:ny] - want32) / np.abs(want32)).max()
    print(f"{workload} n={n}: vs fp32 (first {ny}): reference-bf16 {noise:.2e}, engine padded {err:.2e}, engine ragged {err_r:.2e}")
    assert noise < 1e-2, noise                              # the reference's own bf16 computation
    assert err <= 1.5 * noise + 2e-3, (err, noise)
    assert err_r <= 1.5 * noise + 2e-3, (err_r, noise)
    # bf16 has 8 significand bits; 32 layers of rounding noise land well under 1 %
    assert rel.max() < 1e-2, rel.max()
    assert rel_r.max() < 1e-2, rel_r.max()
    assert ragged[7] == ragged[3]                           # computed once
    cn = cand.cpu().numpy()
    _, inv = np.unique(cn, axis=0, return_inverse=True)
    for u in np.unique(inv):                                # EVERY duplicate group gets one value
        grp = np.where(np.asarray(inv).reshape(-1) == u)[0]
        assert (ragged[grp] == ragged[grp[0]]).all()
    # and the ranking the attack cares about is the same where it is not a near-tie
    gap = np.sort(want)[1] - np.sort(want)[0]
    if gap > 4 * np.abs(got - want).max():
        assert int(got.argmin()) == int(want.argmin())
    if gap > 4 * np.abs(ragged - want).max():
        assert int(ragged.argmin()) == int(want.argmin())
    print(f"{workload} n={n}: max rel diff {rel.max():.2e} (ragged {rel_r.max():.2e}), mean {rel.mean():.2e} "
          f"(ragged {rel_r.mean():.2e}), loss range [{want.min():.4f}, {want.max():.4f}], {distinct} distinct")


@pytest.mark.parametrize("workload", ["gcg", "joint"])
def test_7b_scoring_fp32_within_1e_4(workload):
    """north_star's "fp32 losses within 1e-4" at BASELINE width: LLaVA-1.5-7B width (D = 4096, 32 heads x 128, FFN 11008,
    V = 32064; the 576-token CLIP image prefix in joint mode, S = 644), fp32, 4 decoder layers, search_width 512 drawn
    like the sampler draws it.  The engine's scoring -- prefix keys/values reused, ragged rows with duplicates computed
    once, target rows only, the padded shared-prefix path, and padded chunks of 100 -- against the reference's call
    shape (:1112-1225, :1282-1299: emb(ids) + repeated segments, full-sequence forward, full (B,S,V) logits, torch
    cross-entropy): every per-candidate loss within 1e-4 relative, same argmin."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import segment_order

    dev = torch.device(DEV)
    n = 512
    model, tok, proc, messages, goal, target, image, norm = build_plugins(workload, dev, torch.float32, 4)
    tc = model.config.text_config
    assert (tc.hidden_size, tc.num_attention_heads, tc.intermediate_size, tc.num_hidden_layers) == (4096, 32, 11008, 4)
    assert model.get_input_embeddings().num_embeddings == 32064 and model.dtype == torch.float32
    joint = workload == "joint"
    cfg = BimodalAttackConfig(num_steps=1, search_width=n, seed=1, verbosity="ERROR", pgd_attack=joint,
                              gcg_attack=True, joint_eval=joint, images_folder=tempfile.mkdtemp())
    order = segment_order("pgd", "llava", single=True) if joint else segment_order("gcg", "llava", no_joint_eval=True)
    g = torch.Generator(device=DEV).manual_seed(0)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    n_opt = ids.shape[1]
    cand = ids.repeat(n, 1)
    pos = torch.randint(0, n_opt, (n,), generator=g, device=DEV)
    pool = torch.randint(5, 32000, (n_opt, 256), generator=g, device=DEV)      # a "top-k" table per position
    cand[torch.arange(n, device=DEV), pos] = pool[pos, torch.randint(0, 256, (n,), generator=g, device=DEV)]
    cand[7] = cand[3]                                       # an exact duplicate and a copy of the parent
    cand[11] = ids[0]
    cand = cand.contiguous()
    got = {}
    for name, kw in (("ragged", {}), ("padded", dict(ragged_suffix=False)), ("chunks of 100", dict(ragged_suffix=False, chunk=100))):
        atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, loss_in_model_dtype=False,
                                                                                strict=True, **kw))
        atk._prepare_prompt(messages, target)
        with torch.no_grad():
            feats = atk.hf.image_features(image) if joint else None
            got[name] = atk.score_candidates(cand, order, feats, parent=ids).float().cpu().numpy()
        assert not atk.fallbacks, atk.fallbacks
        st = atk.score_stats
        assert (st["ragged_calls"], st["padded_calls"]) == {"ragged": (1, 0), "padded": (0, 1), "chunks of 100": (0, 6)}[name]
        assert feats is None or feats.shape[1] == 576
    with torch.no_grad():
        want = _reference_call_shape_losses(model, atk, cand, order, feats, chunk=8)
    seq = sum((n_opt if nm == "optim" else (576 if nm == "image" else atk.seg[nm].shape[1])) for nm in order)
    assert atk.T == 20 and seq == (644 if joint else 66)                        # the BASELINE sequence lengths
    for name, v in got.items():
        rel = np.abs(v - want) / np.abs(want)
        print(f"fp32 {workload} {name}: max rel {rel.max():.2e} mean {rel.mean():.2e}; loss range [{want.min():.5f}, {want.max():.5f}]")
        np.testing.assert_allclose(v, want, rtol=1e-4, atol=0)                  # north_star's bar
        gap = np.sort(want)[1] - np.sort(want)[0]
        if gap > 4 * np.abs(v - want).max():
            assert int(v.argmin()) == int(want.argmin())
    assert got["ragged"][7] == got["ragged"][3]


def test_gemma3_4b_scoring_fp32_within_1e_4():
    """The same bar at BASELINE configs[4]'s width: Gemma-3-4b-it shape (D = 2560, 8 query heads on 4 key/value heads x
    256, FFN 10240, V = 262208, scaled embedding, q/k norms, sandwich norms, suffix in FRONT of the 256 image tokens:
    S = 324), fp32, 3 decoder layers, 64 candidates: ragged rows, the padded block and padded chunks of 24 against the
    reference's call shape, every loss within 1e-4 relative."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import segment_order

    dev = torch.device(DEV)
    n = 64
    model, tok, proc, messages, goal, target, image, norm = build_plugins("gemma_joint", dev, torch.float32, 3)
    tc = model.config.text_config
    assert (tc.hidden_size, tc.num_attention_heads, tc.num_key_value_heads, tc.head_dim, tc.num_hidden_layers) == (2560, 8, 4, 256, 3)
    cfg = BimodalAttackConfig(num_steps=1, search_width=n, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=True,
                              joint_eval=True, images_folder=tempfile.mkdtemp())
    order = segment_order("pgd", "gemma3", single=True)
    g = torch.Generator(device=DEV).manual_seed(0)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    n_opt = ids.shape[1]
    cand = ids.repeat(n, 1)
    pos = torch.randint(0, n_opt, (n,), generator=g, device=DEV)
    cand[torch.arange(n, device=DEV), pos] = torch.randint(5, 262144, (n,), generator=g, device=DEV)
    cand[7] = cand[3]
    cand[11] = ids[0]
    cand = cand.contiguous()
    got = {}
    for name, kw in (("ragged", {}), ("padded", dict(ragged_suffix=False)), ("chunks of 24", dict(ragged_suffix=False, chunk=24))):
        atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, loss_in_model_dtype=False,
                                                                                strict=True, **kw))
        atk._prepare_prompt(messages, target)
        assert atk.hf.emb_scale != 1.0 and atk.embedding_layer.num_embeddings == 262208
        with torch.no_grad():
            feats = atk.hf.image_features(image)
            got[name] = atk.score_candidates(cand, order, feats, parent=ids).float().cpu().numpy()
        assert not atk.fallbacks, atk.fallbacks
    with torch.no_grad():
        want = _reference_call_shape_losses(model, atk, cand, order, feats, chunk=4)
    assert sum((n_opt if nm == "optim" else (256 if nm == "image" else atk.seg[nm].shape[1])) for nm in order) == 324
    for name, v in got.items():
        rel = np.abs(v - want) / np.abs(want)
        print(f"fp32 gemma3-4b {name}: max rel {rel.max():.2e} mean {rel.mean():.2e}; loss range [{want.min():.5f}, {want.max():.5f}]")
        np.testing.assert_allclose(v, want, rtol=1e-4, atol=0)
    assert got["ragged"][7] == got["ragged"][3]


@pytest.mark.parametrize("workload", ["gcg", "joint"])
def test_7b_gradient_fp32_matches_reference_call_shape(workload):
    """The gradient pass at BASELINE width in fp32 (LLaVA-1.5-7B width, 4 decoder layers, the CLIP tower in joint mode)
    against the reference's own formulation (:953-1028): a one-hot leaf times the embedding table, the llava segment
    order, FULL (1,S,V) logits, torch cross-entropy over the shifted target slice, autograd to the one-hot and the
    pixels.  The engine's pass (embedding leaf + E^T product, target rows only, HIP cross-entropy with its own backward,
    fused norm / MLP / rotary backward halves, mask-free attention) must give the same token gradient (19 x 32064) and
    pixel gradient to 1e-4 of their scale, the same loss to 1e-5, and -- what the attack consumes -- the same top-256
    tokens per position wherever the reference's own gradient is not tied within that tolerance."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig, ops
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)
    joint = workload == "joint"
    model, tok, proc, messages, goal, target, image, norm = build_plugins(workload, dev, torch.float32, 4)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=joint, gcg_attack=True,
                              joint_eval=joint, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, graph_gradient=False, strict=True))
    atk._prepare_prompt(messages, target)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    img = image.detach().clone().requires_grad_() if joint else None
    g_tok, g_img, loss = atk._gradient_eager(ids, img)
    # the reference's formulation
    E = atk.embedding_layer
    V = E.num_embeddings
    onehot = torch.nn.functional.one_hot(ids, num_classes=V).to(model.dtype).requires_grad_()
    optim_embeds = onehot @ E.weight
    img_r = image.detach().clone().requires_grad_() if joint else None
    if joint:
        feats = atk.hf.image_features(img_r)
        parts = [atk.seg["before_img"], feats.to(model.dtype), atk.seg["before_suffix"], optim_embeds, atk.seg["after"], atk.seg["target"]]
    else:
        parts = [atk.seg["before"], optim_embeds, atk.seg["after"], atk.seg["target"]]
    x = torch.cat(parts, dim=1)
    logits = model(inputs_embeds=x, use_cache=False).logits
    T = atk.T
    shift = x.shape[1] - T
    ref_loss = torch.nn.functional.cross_entropy(logits[0, shift - 1:-1, :], atk.target_ids[0])
    grads = torch.autograd.grad(ref_loss, [onehot] + ([img_r] if joint else []))
    assert x.shape[1] == (644 if joint else 66)
    np.testing.assert_allclose(float(loss), float(ref_loss), rtol=1e-5)
    ref_tok = grads[0][0]
    err = float((g_tok[0] - ref_tok).abs().max() / ref_tok.abs().max())
    print(f"fp32 gradient {workload}: loss {float(loss):.6f}, token-gradient max err / scale {err:.2e}"
          + (f", pixel-gradient {float((g_img - grads[1]).abs().max() / grads[1].abs().max()):.2e}" if joint else ""))
    assert err < 1e-4
    if joint:
        assert float((g_img - grads[1]).abs().max() / grads[1].abs().max()) < 1e-4
    # the selection the attack makes from it
    mine = ops.mask_topk(g_tok[0].contiguous(), atk.mask_bits, 256).cpu().numpy()
    theirs = ops.mask_topk(ref_tok.contiguous(), atk.mask_bits, 256).cpu().numpy()
    tol = 4 * float((g_tok[0] - ref_tok).abs().max())
    rt = ref_tok.cpu().numpy()
    for p_ in range(mine.shape[0]):
        if not np.array_equal(mine[p_], theirs[p_]):
            diff = set(mine[p_].tolist()) ^ set(theirs[p_].tolist())
            edge = np.sort(rt[p_][theirs[p_]])[-1]
            # only tokens whose gradient sits within the tolerance of the 256th value, or near-tied neighbours in order
            assert all(abs(rt[p_][t_] - edge) <= tol for t_ in diff), (p_, diff)
            vals_m, vals_t = rt[p_][mine[p_]], rt[p_][theirs[p_]]
            assert np.abs(np.sort(vals_m) - np.sort(vals_t)).max() <= tol


def _reference_gradient(model, atk, ids, image):
    """The reference's gradient pass (:953-1028) as it calls the model: one-hot @ E.weight (the UNSCALED table, whatever
    the model family), the llava segment order, full (1,S,V) logits, torch's mean cross-entropy on the target slice."""
    E = atk.embedding_layer
    dt = E.weight.dtype
    onehot = torch.nn.functional.one_hot(ids, num_classes=E.num_embeddings).to(dt).requires_grad_()
    optim_embeds = onehot @ E.weight
    img = image.detach().clone().requires_grad_()
    feats = atk.hf.image_features(img)
    parts = [atk.seg["before_img"].to(dt), feats.to(dt), atk.seg["before_suffix"].to(dt), optim_embeds, atk.seg["after"].to(dt),
             atk.seg["target"].to(dt)]
    x = torch.cat(parts, dim=1)
    logits = model(inputs_embeds=x, use_cache=False).logits
    shift = x.shape[1] - atk.T
    loss = torch.nn.functional.cross_entropy(logits[0, shift - 1:-1, :], atk.target_ids[0])
    g_tok, g_img = torch.autograd.grad(loss, [onehot, img])
    return g_tok[0].detach(), g_img.detach(), float(loss.detach())


def test_gemma3_4b_gradient_matches_reference_call_shape():
    """VERDICT r3 item 2(a).  Gemma-3-4b-it width (D = 2560, 256-wide heads, grouped K/V, V = 262208, SigLIP tower of
    4096 patches with 72-wide heads pooled to 256 image tokens).  (1) fp32, 4 text layers: the engine's gradient pass
    (target rows only, fused norms / gates / rotary, padded vision heads, transposed weight copies, its own
    cross-entropy) against the reference's call shape -- llava segment order, unscaled table (:968, :981-991) --
    token and pixel gradients within 1e-4 of the gradient's scale.  (2) the full-depth bf16 model: both gradients
    finite, and as close to the fp32 model's as the reference's own bf16 pass is (x1.5 + 2e-3, relative L2)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)

    def engine(dtype, layers, **opts):
        model, tok, proc, messages, goal, target, image, norm = build_plugins("gemma_joint", dev, dtype, layers)
        cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=True,
                                  joint_eval=True, images_folder=tempfile.mkdtemp())
        atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True, **opts))
        atk._prepare_prompt(messages, target)
        ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
        return model, atk, ids, image

    # ---- (1) fp32, 4 layers ----------------------------------------------------------------------------------
    model, atk, ids, image = engine(torch.float32, 4, graph_gradient=False)
    assert atk.hf.model_type == "gemma3" and atk.hf.emb_scale != 1.0
    g_tok, g_img, loss = atk._gradient_eager(ids, image.detach().clone().requires_grad_())
    r_tok, r_img, r_loss = _reference_gradient(model, atk, ids, image)
    np.testing.assert_allclose(float(loss), r_loss, rtol=1e-5)
    e_tok = float((g_tok[0] - r_tok).abs().max() / r_tok.abs().max())
    e_img = float((g_img - r_img).abs().max() / r_img.abs().max())
    print(f"gemma3-4b fp32 gradient (4 layers): loss {float(loss):.6f}, token-gradient max err / scale {e_tok:.2e}, pixel-gradient {e_img:.2e}")
    assert e_tok < 1e-4 and e_img < 1e-4
    del model, atk
    torch.cuda.empty_cache()

    # ---- (2) bf16, full depth: the replayed pass the attack runs ------------------------------------------------
    model, atk, ids, image = engine(torch.bfloat16, 34)
    img = image.detach().clone().requires_grad_()
    with _KernelSpy() as spy, torch.enable_grad():
        first = [t.clone() for t in atk.compute_gradient(ids, img)]          # eager warm-up + capture + first replay
    with torch.enable_grad():
        again = atk.compute_gradient(ids, img)                                # a replay
    assert "gradient" in atk.graphs_captured and not atk.fallbacks
    # round 5: every attention of the pass is on the hand-written pair -- SigLIP's 27 layers at 4096 tokens x 16 heads of 72
    # with every key visible, the decoder's 34 at 8 query heads over 4 key/value heads of 256, causal -- forward and backward
    # (warm-up + capture: two passes counted); no atomics left, so a replay reproduces the first run bit for bit
    seen = spy.seen
    tower = [c for c in seen["causal_fwd"] if c[3] == 72 and not c[4]]
    dec = [c for c in seen["causal_fwd"] if c[3] == 256 and c[4]]
    assert len(tower) in (26 * 2, 27 * 2) and all(c[:3] == (4096, 4096, 16) for c in tower), tower[:2]
    # (the pass's rows: prompt, 256 image tokens, suffix, target without its last token -- one sequence of ~320)
    assert len(dec) == 34 * 2 and len(set(dec)) == 1 and dec[0][2] == 8 and dec[0][0] == dec[0][1] and 300 <= dec[0][0] <= 340, dec[:2]
    assert len([c for c in seen["causal_bwd"] if c[3] == 256]) == 34 * 2
    assert len([c for c in seen["causal_bwd"] if c[3] == 72]) in (26 * 2, 27 * 2)
    for a, b in zip(first, again):
        # (the hand-written attention has no atomics; the library's stream-K products of this model's shapes may: two
        # replays agree to rounding at least)
        assert bool(torch.isfinite(a.float()).all()) and bool(torch.isfinite(b.float()).all())
        assert float((a.float() - b.float()).norm() / b.float().norm()) < 2e-2
    print("gemma3-4b replay bit-equal to the first run:", all(torch.equal(a, b) for a, b in zip(first, again)))
    g_tok, g_img = first[0][0].float(), first[1].float()
    # the reference's formulation on the UNPATCHED HuggingFace modules (stock attention, aten norms) in bf16: the engine
    # must sit as close to it as two bf16 computations of one function do
    del atk._grad_graph
    atk._grad_graph = None
    with torch.enable_grad():
        p_tok, p_img, p_loss = _plain_hf_gradient(model, atk, ids, image, atk.hf.normalize)
    b_tok, b_img, b_loss = _reference_gradient(model, atk, ids, image)         # the reference's own bf16 computation
    segs16 = atk.seg
    model.float()
    atk.embedding_layer.embed_scale.fill_(atk.hf.emb_scale)                    # (keeps the bf16-rounded sqrt(D))
    atk.seg = {k_: v_.float() for k_, v_ in segs16.items()}
    f_tok, f_img, f_loss = _reference_gradient(model, atk, ids, image)
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())              # noqa: E731
    noise_tok, noise_img = rel(b_tok, f_tok), rel(b_img, f_img)
    err_tok, err_img = rel(g_tok, f_tok), rel(g_img, f_img)
    print(f"gemma3-4b bf16 gradient (34 layers): loss engine {float(first[2]):.4f} reference-bf16 {b_loss:.4f} fp32 {f_loss:.4f}; "
          f"token gradient rel-L2 vs fp32: engine {err_tok:.3e}, reference-bf16 {noise_tok:.3e}; pixel gradient: engine "
          f"{err_img:.3e}, reference-bf16 {noise_img:.3e}")
    assert abs(float(first[2]) - f_loss) <= 1.5 * abs(b_loss - f_loss) + 2e-2 * abs(f_loss)
    assert err_tok <= 1.5 * noise_tok + 2e-3 and err_img <= 1.5 * noise_img + 2e-3
    # ... and against the plain-modules bf16 pass as the yardstick, with the SIGN of the pixel gradient (the PGD step, :1033)
    plain_tok, plain_img = rel(p_tok, f_tok), rel(p_img, f_img)
    big = f_img.abs() > 0.05 * f_img.abs().max()
    agree = lambda a, b: float((torch.sign(a[big]) == torch.sign(b[big])).float().mean())   # noqa: E731
    s_ref, s_eng = agree(p_img.float(), f_img), agree(g_img, f_img)
    print(f"gemma3-4b plain-modules bf16 pass: loss {p_loss:.4f}; rel-L2 vs fp32 token {plain_tok:.3e} pixel {plain_img:.3e}; sign "
          f"agreement with fp32 on {int(big.sum())} pixels > 5 % of max: engine {s_eng:.4f}, plain bf16 {s_ref:.4f}")
    assert err_tok <= 1.5 * plain_tok + 2e-3 and err_img <= 1.5 * plain_img + 2e-3, (err_tok, plain_tok, err_img, plain_img)
    assert s_eng >= s_ref - 0.02, (s_eng, s_ref)


def _plain_hf_gradient(model, atk, ids, image, norm):
    """The reference's gradient pass (:953-1028) on the UNPATCHED HuggingFace modules: `model.get_image_features` called
    as the reference calls it (:972-979) with the stock attention (no engine context of any kind), one-hot @ E.weight,
    the llava segment order, full (1,S,V) logits, torch's mean cross-entropy on the shifted target slice."""
    from bimodalattack_amd.hf_adapter import features_tensor
    assert not any("forward" in m.__dict__ for m in model.modules()), "an engine patch is still on the model"
    E = atk.embedding_layer
    dt = E.weight.dtype
    onehot = torch.nn.functional.one_hot(ids, num_classes=E.num_embeddings).to(dt).requires_grad_()
    optim_embeds = onehot @ E.weight
    img = image.detach().clone().requires_grad_()
    if atk.hf.is_gemma_processor:
        feats = features_tensor(model.get_image_features(pixel_values=norm(img)))
    else:
        feats = features_tensor(model.get_image_features(pixel_values=norm(img), vision_feature_layer=-2,
                                                         vision_feature_select_strategy="default"))
    parts = [atk.seg["before_img"].to(dt), feats.to(dt), atk.seg["before_suffix"].to(dt), optim_embeds, atk.seg["after"].to(dt),
             atk.seg["target"].to(dt)]
    x = torch.cat(parts, dim=1)
    logits = model(inputs_embeds=x, use_cache=False).logits
    shift = x.shape[1] - atk.T
    loss = torch.nn.functional.cross_entropy(logits[0, shift - 1:-1, :], atk.target_ids[0])
    g_tok, g_img = torch.autograd.grad(loss, [onehot, img])
    return g_tok[0].detach().float(), g_img.detach().float(), float(loss.detach())


class _KernelSpy:
    """Counts what reaches the hand-written kernels' torch-side entry points while a block runs."""

    def __init__(self):
        from bimodalattack_amd import ops
        self.ops, self.seen = ops, dict(causal_fwd=[], causal_bwd=[], gemm_mid=[], gemm_nt=[])

    def __enter__(self):
        ops, seen = self.ops, self.seen
        self._fwd, self._bwd = ops.causal_attention, ops.causal_attention_bwd

        def fwd(q, k, v, scale, causal=True):
            seen["causal_fwd"].append((q.shape[0], k.shape[0], q.shape[1], q.shape[2], bool(causal)))
            return self._fwd(q, k, v, scale, causal)

        def bwd(q, k, v, *a, **kw):
            seen["causal_bwd"].append((q.shape[0], k.shape[0], q.shape[1], q.shape[2], bool(kw.get("causal", True))))
            return self._bwd(q, k, v, *a, **kw)

        ops.causal_attention, ops.causal_attention_bwd = fwd, bwd
        ops.GEMM_MID_HOOK = lambda x, w: seen["gemm_mid"].append((x.numel() // x.shape[-1], w.shape[0], w.shape[1]))
        ops.GEMM_NT_HOOK = lambda x, w: seen["gemm_nt"].append((x.numel() // x.shape[-1], w.shape[0], w.shape[1]))
        return self

    def __exit__(self, *exc):
        ops = self.ops
        ops.causal_attention, ops.causal_attention_bwd = self._fwd, self._bwd
        ops.GEMM_MID_HOOK = ops.GEMM_NT_HOOK = None
        return False


@pytest.mark.parametrize("workload", ["pgd", "joint"])
def test_7b_image_gradient_bf16_matches_reference_call_shape(workload):
    """the round-4 hot-path kernels IN COMPOSITION at BASELINE size (configs[1] and [3]) against the
    reference's own formulation of the gradient pass (:953-1028) -- LLaVA-1.5-7B shape, bf16, 32 decoder layers, the
    24-layer CLIP tower, the 643-row image prompt.  Engine = `compute_gradient` as the attack runs it (strict; hipGraph
    replay; PGD-only: the one-sequence pass, joint: scoring prefix with history + tail), with the launches counted:
    `bma_gemm_mid` (the 599-644-row products), `bma_causal_attention` at 128-wide heads causal (decoder) AND at 64-wide
    heads with every key visible (the tower), `bma_gemm_nt` (joint: the 44-row tail).  Reference = the unpatched
    HuggingFace modules called as the reference calls them, in bf16 and -- after `model.float()` -- in fp32.
    Bars (measured yardstick, as for scoring): relative L2 of engine-vs-fp32 <= 1.5 x reference-bf16-vs-fp32 + 2e-3 for
    the pixel gradient (and the token gradient in joint mode), the loss likewise, and the SIGN of the pixel gradient --
    which is the PGD step (:1033) -- agrees with fp32's on the pixels that matter at least as often as the reference's
    own bf16 pass does (-2 %), and with the reference-bf16 pass's on > 90 % of them."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig, native
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)
    joint = workload == "joint"
    model, tok, proc, messages, goal, target, image, norm = build_plugins(workload, dev, torch.bfloat16, 32)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=joint,
                              joint_eval=joint, eps=64 / 255, alpha=4 / 255, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True))
    atk._prepare_prompt(messages, target)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    img = image.detach().clone().requires_grad_()

    # ---- the engine, as the attack runs it: first call = eager warm-up (counted) + capture + replay; second = a replay
    native.profile_enable(True)
    with _KernelSpy() as spy, torch.enable_grad():
        # (joint: the pass makes its own scoring prefix current for this image -- what step t's scoring call does in a run)
        first = [None if t is None else t.detach().clone() for t in atk.compute_gradient(ids, img)]
    torch.cuda.synchronize()
    prof = native.profile_read()
    native.profile_enable(False)
    with torch.enable_grad():
        again = atk.compute_gradient(ids, img)
    assert not atk.fallbacks, atk.fallbacks
    assert ({"grad_prefix", "grad_tail"} if joint else {"gradient"}) <= set(atk.graphs_captured), atk.graphs_captured
    for a, b in zip(first, again):
        if a is not None:                                   # no atomics on these routes: a replay reproduces the first run
            assert bool(torch.isfinite(a.float()).all()) and torch.equal(a, b)
    seen = spy.seen
    dec_fwd = [c for c in seen["causal_fwd"] if c[3] == 128 and c[4]]
    tower_fwd = [c for c in seen["causal_fwd"] if c[3] == 64 and not c[4]]
    dec_bwd = [c for c in seen["causal_bwd"] if c[3] == 128 and c[4]]
    tower_bwd = [c for c in seen["causal_bwd"] if c[3] == 64 and not c[4]]
    # (eager warm-up + capture each run the pass once: counts are per pass x 2; the tower runs its first 23 layers --
    # vision_feature_layer = -2 -- or all 24, depending on the transformers version)
    n_pass = 2
    assert len(tower_fwd) in (23 * n_pass, 24 * n_pass) and all(c[:3] == (577, 577, 16) for c in tower_fwd), tower_fwd[:3]
    assert len(tower_bwd) == 23 * n_pass, len(tower_bwd)      # (the features come from layer -2: the last layer has no backward)
    if joint:
        assert len(dec_fwd) == 2 * 32 * n_pass and sorted(set(c[:2] for c in dec_fwd)) == [(44, 643), (599, 599)]
        assert len([c for c in seen["gemm_nt"] if c[0] == 44]) >= 4 * 32 * n_pass
        assert len([c for c in seen["gemm_mid"] if c[0] == 599]) >= 3 * 32 * n_pass
    else:
        assert len(dec_fwd) == 32 * n_pass and set(c[:3] for c in dec_fwd) == {(643, 643, 32)}
        assert len([c for c in seen["gemm_mid"] if c[0] == 643]) >= 3 * 32 * n_pass
    # (joint: the prefix rows' LAST-layer attention output feeds nothing -- the tail reads that layer's keys/values only --
    # so it has no backward)
    assert len(dec_bwd) == len(dec_fwd) - (n_pass if joint else 0), (len(dec_bwd), len(dec_fwd))
    assert prof["gemm_mid"]["launches"] > 0 and prof["causal_attn"]["launches"] > 0
    assert (prof["gemm_nt"]["launches"] > 0) == joint
    g_tok = None if first[0] is None else first[0][0].float()
    g_img, g_loss = first[1].float(), float(first[2])
    del atk._grad_graph, atk._gp
    atk._grad_graph = atk._gp = None

    # ---- the reference's formulation on the same bf16 model, then on the same weights in fp32 ----------------------
    with torch.enable_grad():
        b_tok, b_img, b_loss = _plain_hf_gradient(model, atk, ids, image, norm)
        segs16 = atk.seg
        model.float()
        atk.seg = {k_: v_.float() for k_, v_ in segs16.items()}
        f_tok, f_img, f_loss = _plain_hf_gradient(model, atk, ids, image, norm)
        atk.seg = segs16
    rel = lambda a, b: float((a - b).norm() / b.norm())              # noqa: E731
    big = f_img.abs() > 0.05 * f_img.abs().max()
    agree = lambda a, b: float((torch.sign(a[big]) == torch.sign(b[big])).float().mean())   # noqa: E731
    noise_img, err_img = rel(b_img, f_img), rel(g_img, f_img)
    s_ref, s_eng, s_cross = agree(b_img, f_img), agree(g_img, f_img), agree(g_img, b_img)
    msg = (f"7B image gradient pass, bf16, {workload}: loss engine {g_loss:.4f} reference-bf16 {b_loss:.4f} fp32 {f_loss:.4f}; pixel "
           f"gradient rel-L2 vs fp32: engine {err_img:.3e}, reference-bf16 {noise_img:.3e}; sign agreement on {int(big.sum())} "
           f"pixels > 5 % of max: engine-vs-fp32 {s_eng:.4f}, reference-bf16-vs-fp32 {s_ref:.4f}, engine-vs-reference-bf16 {s_cross:.4f}")
    if joint:
        noise_tok, err_tok = rel(b_tok, f_tok), rel(g_tok, f_tok)
        msg += f"; token gradient rel-L2 vs fp32: engine {err_tok:.3e}, reference-bf16 {noise_tok:.3e}"
    print(msg)
    assert abs(g_loss - f_loss) <= 1.5 * abs(b_loss - f_loss) + 2e-2 * abs(f_loss)
    assert err_img <= 1.5 * noise_img + 2e-3, (err_img, noise_img)
    if joint:
        assert err_tok <= 1.5 * noise_tok + 2e-3, (err_tok, noise_tok)
    assert s_eng >= s_ref - 0.02 and s_cross > 0.90, (s_eng, s_ref, s_cross)
    if s_ref > 0.99:
        assert s_eng > 0.98, (s_eng, s_ref)


def test_7b_pgd_only_steps():
    """VERDICT r4 item 1, second half: BASELINE configs[1]'s own settings (PGD-only, eps 64/255, alpha 4/255) for six
    steps at FULL size -- (A) the default engine (`bma_gemm_mid`, `bma_causal_attention` for decoder and tower, fused
    PGD-only loop, hipGraph) against (B) the same engine with every round-4 kernel and the graph switched off
    (library products, library attention, eager) and (C) the reference's loop itself on the plain HuggingFace modules
    (:953-1037: gradient, x <- clamp(clamp(x - a*e*sign(g), x0 +- e), 0, 1), loss of the updated image).  Three bf16
    computations of one function: A must sit as close to C as B does (x1.5 + margin) in the per-step losses and in the
    share of pixels that took a different turn, every image stays within eps of the original and inside [0, 1]."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)
    steps, eps, alpha = 6, 64 / 255, 4 / 255
    model, tok, proc, messages, goal, target, image, norm = build_plugins("pgd", dev, torch.bfloat16, 32)
    x0 = image.detach().clone()
    runs = {}
    for name, opts in (("default", {}),
                       ("library", dict(own_b1_kernels=False, graph_gradient=False))):
        cfg = BimodalAttackConfig(num_steps=steps, search_width=8, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=False,
                                  eps=eps, alpha=alpha, images_folder=tempfile.mkdtemp())
        atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True, **opts))
        with _KernelSpy() as spy:
            res = atk.run(messages, goal, target, x0.clone())
        assert not atk.fallbacks, atk.fallbacks
        n_own = len(spy.seen["gemm_mid"]) + len(spy.seen["causal_fwd"])
        assert (n_own > 0) == (name == "default"), (name, n_own)
        assert ("gradient" in atk.graphs_captured) == (name == "default")
        runs[name] = (np.asarray(res.losses, dtype=np.float64), atk.final_image.detach().clone())
        ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
        keep = atk
        del atk
    # (C) the reference's loop on the plain modules: loss_i = loss of the image AFTER step i's update (:605-612)
    x = x0.clone()
    ref_losses = []
    with torch.enable_grad():
        for i in range(steps + 1):
            _, g, loss = _plain_hf_gradient(model, keep, ids, x, norm)
            if i > 0:
                ref_losses.append(float(torch.tensor(loss).to(torch.bfloat16)))
            if i < steps:
                x = torch.clamp(torch.clamp(x - alpha * eps * torch.sign(g), x0 - eps, x0 + eps), 0, 1)
    ref_losses = np.asarray(ref_losses)
    la, lb = runs["default"][0], runs["library"][0]
    xa, xb = runs["default"][1], runs["library"][1]
    frac = lambda a, b: float((a != b).float().mean())              # noqa: E731
    d_ac, d_bc, d_ab = frac(xa, x), frac(xb, x), frac(xa, xb)
    e_a, e_b = np.abs(la - ref_losses).max(), np.abs(lb - ref_losses).max()
    print(f"7B PGD-only, {steps} steps: losses default {la.round(4).tolist()} library {lb.round(4).tolist()} reference {ref_losses.round(4).tolist()}; "
          f"pixels that differ after {steps} steps: default-vs-reference {d_ac:.4f}, library-vs-reference {d_bc:.4f}, default-vs-library {d_ab:.4f}")
    for xi in (xa, xb):
        assert bool(torch.isfinite(xi).all()) and float(xi.min()) >= 0.0 and float(xi.max()) <= 1.0
        assert float((xi - x0).abs().max()) <= eps + 1e-6
    assert np.isfinite(la).all() and np.isfinite(lb).all() and la.shape == (steps,)
    assert e_a <= 1.5 * e_b + 2e-2 * np.abs(ref_losses).max(), (e_a, e_b)
    assert d_ac <= 1.5 * d_bc + 0.005, (d_ac, d_bc)
    # the trajectory goes somewhere: six steps of sign descent do not raise the loss (bf16 losses near 10: one unit in the last place is 0.0625)
    assert la[-1] <= la[0] + 0.0625 and ref_losses[-1] <= ref_losses[0] + 0.0625
    assert d_ab > 0.0 or np.array_equal(la, lb)      # (two different kernel sets: identical images would mean the switch did nothing)


@pytest.mark.parametrize("width", [64, 512])
def test_gemma3_4b_joint_steps_finite(width):
    """VERDICT r3 item 2(b).  bench.py's Gemma-3-4b joint workload (BASELINE configs[4]: dynamic search width, suffix in
    front of the image, hipGraphs on), a few steps with NO step trace -- the configuration rounds 2 and 3 shipped with
    NaN losses: the image features replayed from a hipGraph held NaN rows at every step whose width was not 512 or 256
    (ATen's strided mean inside the projector's RMSNorm; ``BimodalAttack.image_features``).  Every loss finite and in
    the range the CPU oracle reports for this model (14.4 at the start, slowly falling), the image inside [0, 1] and
    within eps of the original, features of the final image finite and equal to the eager call's."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import dynamic_width

    dev = torch.device(DEV)
    model, tok, proc, messages, goal, target, image, norm = build_plugins("gemma_joint", dev, torch.bfloat16, 34)
    steps = 3 if width == 64 else 6
    # the widths of a 600-step run sampled evenly (bench.py): 460, 358, ... -- none of them 512 or 256 until late
    sched = (lambda i: dynamic_width(min(int(round((i + 0.5) * 600 / steps)), 599), width, 600, min(128, width), True)) \
        if width == 512 else None
    eps = 64 / 255
    cfg = BimodalAttackConfig(num_steps=steps, search_width=width, topk=256, seed=1, verbosity="ERROR", pgd_attack=True,
                              gcg_attack=True, joint_eval=True, eps=eps, alpha=4 / 255, dynamic_search=True,
                              min_search_width=min(128, width) if width == 512 else 16, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True, width_override=sched))
    x0 = image.detach().clone()
    res = atk.run(messages, goal, target, image.detach().clone())
    assert len(res.losses) == steps and all(np.isfinite(res.losses)), res.losses
    assert all(12.0 < l < 15.5 for l in res.losses), res.losses
    assert {"gradient", "image_features"} <= set(atk.graphs_captured) and not atk.fallbacks
    final = atk.final_image.detach()
    assert bool(torch.isfinite(final).all()) and float(final.min()) >= 0.0 and float(final.max()) <= 1.0
    assert float((final - x0).abs().max()) <= eps + 1e-6
    with torch.no_grad():
        replayed = atk.scoring_features(final).clone()
        eager = atk.image_features(final)
    assert bool(torch.isfinite(replayed.float()).all()) and torch.equal(replayed, eager)
    assert len(set(atk.n_scored)) > 1                      # the width did change from step to step


def test_gemma3_4b_joint_first_40_steps_of_the_600_step_schedule_hold_their_pace():
    """BASELINE configs[4] changes its search width ~385 times over its 600 steps (reference :919-923) --
    new GEMM row counts, new ragged grids, the library's first-sight cost of an unseen shape.  The first 40 CONSECUTIVE steps
    of the real schedule (widths 512 -> 478, a new width every step or two), timed by the host clock at the engine's step
    hook (one packed read-back per step: consecutive hooks are a step apart): no step from the third on may take more than
    1.3 x the median (the widths themselves differ by 7 %), every loss finite, nothing fell back.  The whole 600 steps are in
    profiles/r6_full_gemma600.json (tools/full_length.py): ms = 64.7 + 1.687 x width, no step over 1.15 x that line."""
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import dynamic_width

    dev = torch.device(DEV)
    model, tok, proc, messages, goal, target, image, norm = build_plugins("gemma_joint", dev, torch.bfloat16, 34)
    steps = 40
    sched = lambda i: dynamic_width(i, 512, 600, 128, True)             # noqa: E731  (the 600-step run's own widths)
    stamps = []
    cfg = BimodalAttackConfig(num_steps=steps, search_width=512, topk=256, seed=1, verbosity="ERROR", pgd_attack=True,
                              gcg_attack=True, joint_eval=True, eps=64 / 255, alpha=4 / 255, dynamic_search=True,
                              min_search_width=128, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True, width_override=sched,
                                                                            step_hook=lambda i: stamps.append(time.perf_counter())))
    res = atk.run(messages, goal, target, image.detach().clone())
    assert len(res.losses) == steps and all(np.isfinite(res.losses)) and not atk.fallbacks
    ms = [1e3 * (b - a) for a, b in zip(stamps[:-1], stamps[1:])]
    assert len(ms) == steps and len(set(atk.n_scored)) >= 15           # (the width did change: 512 ... 478 less what the filter took)
    med = float(np.median(ms[2:]))
    worst = max(range(2, steps), key=lambda i: ms[i])
    assert ms[worst] <= 1.3 * med, (worst, ms[worst], med, [round(v, 1) for v in ms])


def test_gemma3_4b_scoring_equals_reference_call_shape():
    """Gemma-3-4b-it shape, bf16 (BASELINE configs[4]): 34 layers, 256-wide heads, grouped K/V heads, V = 262208,
    suffix in FRONT of the image (:1150-1163), scaled embedding (:1142).  The engine's scoring path for the joint
    layout against the reference's call shape on the same model."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.layout import segment_order

    dev = torch.device(DEV)
    n = 40
    model, tok, proc, messages, goal, target, image, norm = build_plugins("gemma_joint", dev, torch.bfloat16, 34)
    cfg = BimodalAttackConfig(num_steps=1, search_width=n, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=True,
                              joint_eval=True, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, loss_in_model_dtype=False,
                                                                            strict=True))
    atk._prepare_prompt(messages, target)
    assert atk.hf.model_type == "gemma3" and atk.hf.emb_scale != 1.0 and atk.embedding_layer.num_embeddings == 262208
    g = torch.Generator(device=DEV).manual_seed(0)
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    n_opt = ids.shape[1]
    cand = ids.repeat(n, 1)
    pos = torch.randint(0, n_opt, (n,), generator=g, device=DEV)
    cand[torch.arange(n, device=DEV), pos] = torch.randint(5, 262144, (n,), generator=g, device=DEV)
    cand[7] = cand[3]
    cand[11] = ids[0]
    order = segment_order("pgd", "gemma3", single=True)
    assert order.index("optim") < order.index("image")
    with torch.no_grad():
        feats = atk.hf.image_features(image)
        got = atk.score_candidates(cand.contiguous(), order, feats, parent=ids).float().cpu().numpy()
        chunked = atk.score_candidates(cand.contiguous(), order, feats).float().cpu().numpy()
        assert not atk.fallbacks, atk.fallbacks
        want = _reference_call_shape_losses(model, atk, cand, order, feats, chunk=4)
        # the yardstick for "bf16 rounding noise" on THIS model: the same reference call shape in fp32.  The
        # engine must be as close to it as the reference's own bf16 forward is (Gemma-3's sqrt(D)-scaled
        # embeddings and 34 layers make that noise ~1 %, larger than on the Llama shape).
        segs16 = atk.seg
        model.float()
        # same function in fp32: the embedding scale keeps the bf16-rounded value the bf16 model multiplies by
        atk.embedding_layer.embed_scale.fill_(atk.hf.emb_scale)
        atk.seg = {k_: v_.float() for k_, v_ in segs16.items()}
        want32 = _reference_call_shape_losses(model, atk, cand, order, feats.float(), chunk=4)
        atk.seg = segs16
    rel = np.abs(got - want) / np.abs(want)
    noise = (np.abs(want - want32) / np.abs(want32)).max()
    err = (np.abs(got - want32) / np.abs(want32)).max()
    err_chunked = (np.abs(chunked - want32) / np.abs(want32)).max()
    print(f"gemma3-4b joint: engine vs fp32 {err:.2e} (chunked {err_chunked:.2e}), reference-bf16 vs fp32 {noise:.2e}, "
          f"engine vs reference-bf16 max {rel.max():.2e} mean {rel.mean():.2e}, loss range [{want.min():.4f}, {want.max():.4f}]")
    assert noise < 3e-2                                      # the reference's own bf16 computation
    assert err < 1.5 * noise + 2e-3 and err_chunked < 1.5 * noise + 2e-3
    assert rel.max() < 3e-2, rel.max()
    assert got[7] == got[3]
    gap = np.sort(want32)[1] - np.sort(want32)[0]
    if gap > 4 * np.abs(got - want32).max():
        assert int(got.argmin()) == int(want32.argmin())


def test_text_gradient_pass_with_one_launch_attention_at_7b_width(monkeypatch):
    """the text-only gradient pass at LLaVA-1.5-7B width (bf16, 65 rows, 32 heads of 128; 4
    layers here) with rotary + attention between the fused q/k/v projection and o_proj as ONE launch each way
    (bma_b1_attention) against the same pass through HuggingFace's rotary + the library's attention: same loss and token
    gradient to bf16 noise, both as close to the fp32 model's gradient as each other; the kernel really runs (two
    launches per layer, counted by the library's own profiler) and the pass still replays from a hipGraph."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig, native
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)
    layers = 4
    model, tok, proc, messages, goal, target, image, norm = build_plugins("gcg", dev, torch.bfloat16, layers)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=False, gcg_attack=True,
                              images_folder=tempfile.mkdtemp())
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    got = {}
    from bimodalattack_amd import ops
    for one_launch in (True, False):
        monkeypatch.setitem(ops.OWN_KERNELS, "b1_attention", one_launch)       # (an A/B switch of ONE of the own kernels: ops.OWN_KERNELS)
        atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True))
        atk._prepare_prompt(messages, target)
        assert atk.engine_state()["fusions"]["b1_attention_blocks"] == (layers if one_launch else 0)
        native.profile_enable(True)
        g_tok, _, loss = atk._gradient_eager(ids, None)
        torch.cuda.synchronize()
        n = native.profile_read()["b1_attn"]["launches"]
        native.profile_enable(False)
        assert n == (2 * layers if one_launch else 0), n
        with torch.enable_grad():
            atk.compute_gradient(ids, None)
            replay = atk.compute_gradient(ids, None)
        assert "gradient" in atk.graphs_captured and not atk.fallbacks
        assert torch.equal(replay[0], g_tok) and torch.equal(replay[2], loss)          # (no atomics on this route: bit for bit)
        got[one_launch] = (g_tok[0].float().clone(), float(loss))
        del atk
    model.float()
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, graph_gradient=False,
                                                                            fused_elementwise=False))
    atk._prepare_prompt(messages, target)
    ref, ref_loss = atk._gradient_eager(ids, None)[0][0].float(), float(atk._gradient_eager(ids, None)[2])
    rel = lambda a, b: float((a - b).norm() / b.norm())              # noqa: E731
    e1, e0, cross = rel(got[True][0], ref), rel(got[False][0], ref), rel(got[True][0], got[False][0])
    print(f"text gradient pass, 7B width, {layers} layers: loss one-launch {got[True][1]:.4f} library {got[False][1]:.4f} fp32 {ref_loss:.4f}; "
          f"token gradient rel-L2 vs fp32: one-launch {e1:.3e}, library {e0:.3e}; one-launch vs library {cross:.3e}")
    assert abs(got[True][1] - ref_loss) <= 1.5 * abs(got[False][1] - ref_loss) + 5e-3 * abs(ref_loss)
    assert e1 <= 1.5 * e0 + 2e-3 and cross <= 2.0 * e0 + 2e-3


def test_image_gradient_pass_forms_each_projection_once():
    """The one-launch attention (<= 80 tokens) must step aside for the 643-row image pass BEFORE it has formed the fused
    q/k/v product -- a refusal behind it ran that projection twice per layer (round 4's first PGD-only measurement:
    35.4 -> 37.5 ms).  Counted at the dispatcher: one forward product of width 3 x 4096 per layer, one input-gradient
    product of that shape per layer (on bma_gemm_mid since round 4: counted through its hook)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from torch.utils._python_dispatch import TorchDispatchMode
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)
    layers = 2
    model, tok, proc, messages, goal, target, image, norm = build_plugins("pgd", dev, torch.bfloat16, layers)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=False,
                              images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True, graph_gradient=False))
    atk._prepare_prompt(messages, target)
    assert atk.engine_state()["fusions"]["b1_attention_blocks"] == layers
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    img = image.detach().clone().requires_grad_()
    atk._gradient_eager(ids, img)                       # derived weight copies made outside the count
    seen = []

    class Count(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = getattr(getattr(func, "overloadpacket", None), "__name__", "")
            if name in ("mm", "addmm") and torch.is_tensor(args[-1]):
                a, b = args[-2], args[-1]
                seen.append((a.shape[0], b.shape[1], a.shape[1]))
            return func(*args, **(kwargs or {}))

    from bimodalattack_amd import ops
    ops.GEMM_MID_HOOK = lambda x, w: seen.append((x.numel() // x.shape[-1], w.shape[0], w.shape[1]))   # products on bma_gemm_mid never reach aten
    try:
        with Count():
            atk._gradient_eager(ids, img.detach().clone().requires_grad_())
    finally:
        ops.GEMM_MID_HOOK = None
    rows = 643
    fwd = [s_ for s_ in seen if s_ == (rows, 3 * 4096, 4096)]
    bwd = [s_ for s_ in seen if s_ == (rows, 4096, 3 * 4096)]
    # (a forward product may reach the dispatcher under another name -- one layer's does -- but never more than once per
    # layer; the doubled projection showed as 2 x layers here)
    assert 1 <= len(fwd) <= layers and len(bwd) == layers, (len(fwd), len(bwd))
    n_qkv_shaped = len([s_ for s_ in seen if s_[1] == 3 * 4096 and s_[2] == 4096])
    assert n_qkv_shaped <= layers, seen


def test_maskless_b1_attention_gradient_matches_masked(monkeypatch):
    """The gradient pass of the image prompt (643 rows, LLaVA-1.5-7B width, 2 layers here) with the library
    attention asked for `is_causal` against the same pass handed HuggingFace's mask tensor: token and pixel
    gradients agree to bf16 noise (same maths, different library kernels)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig, prefix_attention as pa
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)
    model, tok, proc, messages, goal, target, image, norm = build_plugins("joint", dev, torch.bfloat16, 2)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=True,
                              joint_eval=True, images_folder=tempfile.mkdtemp())
    out, calls = {}, []
    orig = pa.causal_b1_attention
    from bimodalattack_amd import attack as attack_mod
    for maskless in (True, False):
        monkeypatch.setattr(attack_mod, "MASKLESS_B1_ATTENTION", maskless)      # (a module constant since round 5)
        atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, graph_gradient=False, strict=True))
        atk._prepare_prompt(messages, target)
        ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
        img = image.detach().clone().requires_grad_()
        g_tok, g_img, loss = atk._gradient_eager(ids, img)
        out[maskless] = (g_tok.float(), g_img.float(), float(loss))
        calls.append(len(atk.hf.shared_prefix_configs(643)) > 0)
    assert all(calls) and callable(orig)
    (t1, i1, l1), (t0, i0, l0) = out[True], out[False]
    assert abs(l1 - l0) <= 2e-2 * abs(l0)
    assert float((t1 - t0).abs().max()) <= 5e-2 * float(t0.abs().max())
    assert float((i1 - i0).abs().max()) <= 5e-2 * float(i0.abs().max())
    big = i0.abs() > 0.05 * i0.abs().max()
    assert float((torch.sign(i1[big]) == torch.sign(i0[big])).float().mean()) > 0.98


def test_run_experiment_writes_reference_artifacts(tmp_path):
    """The harness loop (reference experiments.py:54-285) on this engine: two prompts, PNGs per
    step under images_<run>/, the seven artefact files, losses.csv consistent with the result."""
    import csv
    from bimodalattack_amd import synthetic as S
    from bimodalattack_amd.artifacts import run_experiment
    model, tok, proc, image = S.tiny_case("llava", device=DEV)
    kwargs = {"num_steps": 2, "search_width": 8, "topk": 16, "dynamic_search": False, "min_search_width": 8,
              "pgd_attack": True, "gcg_attack": True, "alpha": 4 / 255, "eps": 64 / 255, "debug_output": False,
              "alpha_str": "4/255", "eps_str": "64/255", "joint_eval": True, "model": "llava",
              "optim_str_init": S.TINY_OPTIM_INIT}
    pairs = [("tell me a story", "Sure here is"), ("write a plan", "Sure here is a plan")]
    folder = run_experiment("tiny", kwargs, pairs, model, tok, proc, image, S.Normalize(S.CLIP_MEAN, S.CLIP_STD),
                            base=str(tmp_path / "experiments"), rng_device="cpu")
    assert os.path.basename(folder) == "exp1"
    for fn in ("prompts.csv", "losses.csv", "details.csv", "times.csv", "parameters.csv", "best_strings.txt", "summary.csv"):
        assert os.path.getsize(os.path.join(folder, fn)) > 0
    assert sorted(os.listdir(os.path.join(folder, "images_1"))) == ["0.png", "1.png"]
    assert sorted(os.listdir(os.path.join(folder, "images_2"))) == ["0.png", "1.png"]
    rows = list(csv.reader(open(os.path.join(folder, "losses.csv"))))
    assert rows[0] == ["Iteration", "Run 1", "Run 2"] and len(rows) == 3
    assert all(np.isfinite(float(c)) for r in rows[1:] for c in r[1:])
    params = dict(csv.reader(open(os.path.join(folder, "parameters.csv"))))
    assert params["alpha"] == "4/255" and params["num_prompts"] == "2" and params["seed"] == "1" and "alpha_str" not in params


def test_rccl_single_rank_collectives():
    """The exact torch.distributed calls of dist.py / bench.py on the RCCL backend (a 1-rank
    group: two ranks cannot share one GPU under RCCL; multi-rank semantics are covered by the
    gloo tests).  Catches API / dtype / device misuse before the driver's 8-GPU run."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = r"""
import os, torch, torch.distributed as dist
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
from bimodalattack_amd.dist import CandidateSharder
sh = CandidateSharder()
sh.enabled, sh.world = True, 1          # force the collective code path
local = torch.arange(7, dtype=torch.float32, device=dev)
full = sh.gather(local, 7)
assert torch.equal(full, local)
m = sh.gather(torch.ones(7, device=dev), 7, pad=0.0)
assert float(m.sum()) == 7
import numpy as np
order = np.array([3, 0, 4, 1, 2])
vals = torch.tensor([30., 0., 40., 10., 20.], device=dev)      # values of candidates 3, 0, 4, 1, 2
assert sh.gather_dealt(vals, order).tolist() == [0., 10., 20., 30., 40.]
full, hits = sh.gather2(local, torch.ones(7, device=dev), 7)
assert torch.equal(full, local) and float(hits.sum()) == 7
a, b = sh.gather_dealt(vals, order, extra=-vals)
assert a.tolist() == [0., 10., 20., 30., 40.] and b.tolist() == [-0., -10., -20., -30., -40.]
ids = torch.arange(12, device=dev).view(4, 3)
img = torch.rand(1, 3, 8, 8, device=dev).requires_grad_()
keep = img.detach().clone()
sh.sync_state(ids, img)
assert torch.equal(ids, torch.arange(12, device=dev).view(4, 3)) and torch.equal(img.detach(), keep)
both = torch.tensor([1.25, 1.0], device=dev)
sh.broadcast_(both)
t = torch.tensor([1.5], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
dist.destroy_process_group()
print("rccl-ok")
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "rccl-ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("reuse", [True, False])
def test_graphs_are_really_captured_with_a_list_style_normalize(reuse):
    """The hipGraph paths must not silently fall back to eager: with a torchvision-style normalize
    (mean/std as Python lists, rebuilt as tensors on every call -- a host-to-device copy that aborts a
    capture) the engine still captures the gradient pass, the vision tower, the prefix pass and the
    winner re-score -- or, with the scoring prefix reused by the gradient pass (the default in joint mode), the
    prefix-with-history and tail+backward pair -- and the run equals the all-eager run."""
    from bimodalattack_amd import BimodalAttackConfig, synthetic as S
    from bimodalattack_amd.config import EngineOptions
    from bimodalattack_amd.attack import BimodalAttack, _GradientGraph, _GradPrefix, _ReplayGraph

    class ListNormalize:
        mean, std = list(S.CLIP_MEAN), list(S.CLIP_STD)

        def __call__(self, t):
            m = torch.as_tensor(self.mean, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            s = torch.as_tensor(self.std, dtype=t.dtype, device=t.device).view(-1, 1, 1)
            return (t - m) / s

    out = {}
    for eager in (False, True):
        model, tok, proc, image = S.tiny_case("llava", device=DEV)
        cfg = BimodalAttackConfig(num_steps=3, search_width=16, topk=8, pgd_attack=True, gcg_attack=True, joint_eval=True,
                                  eps=64 / 255, alpha=4 / 255, seed=3, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT,
                                  images_folder=tempfile.mkdtemp())
        kw = dict(graph_gradient=False, graph_scoring=False) if eager else {}
        kw["joint_winner_from_batch"] = False           # keep the batch-1 winner re-score (and its graph) in play
        attack = BimodalAttack(model, tok, proc, cfg, ListNormalize(),
                               EngineOptions.from_env(rng_device="cpu", grad_prefix_reuse=reuse, strict=True, **kw))
        res = attack.run("tell me a story", "tell me a story", "Sure here is a story", image)
        out[eager] = res
        assert attack.fallbacks == {}
        assert isinstance(attack._gp, _GradPrefix) == reuse
        if not eager:
            if reuse:
                assert isinstance(attack._gp.g1, torch.cuda.CUDAGraph) and isinstance(attack._gp.g2, torch.cuda.CUDAGraph)
                assert attack._grad_graph is None        # (the buffer initialisation, before step 0, still scores through the plain prefix graph)
            else:
                assert isinstance(attack._grad_graph, _GradientGraph)
                assert isinstance(attack._feat_graph, _ReplayGraph)
                assert attack._prefix_graphs and all(isinstance(g, _ReplayGraph) for g in attack._prefix_graphs.values())
            assert attack._rescore_graphs and all(isinstance(g, _ReplayGraph) for g in attack._rescore_graphs.values())
            assert attack.hf.normalize.ok is True
    np.testing.assert_allclose(out[False].losses, out[True].losses, rtol=1e-5)
    assert out[False].strings == out[True].strings


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_16bit_engine_paths_agree(dtype):
    """The 16-bit-only machinery (fused q/k/v projection, transposed-weight backward, MFMA ragged
    attention where the head size allows) against the plainer paths of the same engine on a 16-bit
    model: same candidates (CPU draws), first-step candidate losses equal to 16-bit rounding noise."""
    from bimodalattack_amd import BimodalAttackConfig, run, synthetic as S
    # one gradient pass for all (the full 1-sequence pass): the prefix-reusing pass of joint mode rounds differently in
    # 16 bits and two of the variants switch it off by construction, which would change the top-k picks under
    # comparison; it rides along as the last variant, where its gradient is held to the same tolerance
    full = dict(grad_prefix_reuse=False)
    variants = [dict(full), dict(full, ragged_suffix=False), dict(full, shared_prefix_attention=False),
                dict(full, prefix_reuse=False),
                dict(full, derived_weight_copies=False, graph_gradient=False),
                dict()]
    out = []
    for eng in variants:
        model, tok, proc, image = S.tiny_case("llava", dtype=dtype, device=DEV)
        trace = []
        cfg = BimodalAttackConfig(num_steps=1, search_width=24, topk=8, pgd_attack=True, gcg_attack=True, joint_eval=True,
                                  eps=64 / 255, alpha=4 / 255, seed=5, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT,
                                  images_folder=tempfile.mkdtemp())
        run(model, tok, proc, "tell me a story", "tell me a story", "Sure here is a story", image, cfg,
            normalize=S.Normalize(S.CLIP_MEAN, S.CLIP_STD), trace=trace, rng_device="cpu", **eng)
        out.append(trace[0])
    tol = 4e-2 if dtype == torch.bfloat16 else 6e-3
    base = out[0]
    for eng, st in zip(variants[1:], out[1:]):
        # the gradient may differ in the last bits between variants; compare scoring on identical candidates only
        if np.array_equal(st["sampled"], base["sampled"]):
            np.testing.assert_allclose(st["losses"][0], base["losses"][0], rtol=tol, err_msg=str(eng))
        np.testing.assert_allclose(st["grad_tok"][-1], base["grad_tok"][-1], rtol=0.2,
                                   atol=0.05 * float(np.abs(base["grad_tok"][-1]).max()), err_msg=str(eng))
    assert sum(np.array_equal(st["sampled"], base["sampled"]) for st in out[1:5]) >= 3


def test_padded_vision_heads_same_features_and_pixel_gradient():
    """SigLIP-So400m-shaped tower (4096 patches, 16 heads x 72; 3 layers here): image features and the pixel gradient
    through (a) the hand-written attention pair at the REAL head width (round 5: 72 in memory, 96-wide images in LDS; no
    padded copies, no library call at all), (b) the padded-head library route it replaces (72 -> 96 forward-only, -> 128
    with a backward) and (c) the library's own 72-wide route -- same attention, three sets of kernels, bf16 noise apart."""
    from bimodalattack_amd import ops, prefix_attention as pa, synthetic as S
    from bimodalattack_amd.hf_adapter import HFAdapter

    dev = torch.device(DEV)
    model = S._gemma3(1024, 256, 512, 1, 4, 2, 64, 1152, 4304, 3, 16, 896, 14, 256, 1024, torch.bfloat16, dev, 0, "sdpa")
    tok = S.build_tokenizer(256, 0, 0)
    hf = HFAdapter(model, S.Gemma3Processor(tok, S.GEMMA_TEMPLATE), S.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)))
    assert len(hf.vision_configs()) == 1 and pa.padded_width(72, True) == 128
    image = S.synthetic_image(896, 896, seed=0, device=dev)
    w = torch.randn(256, 256, device=dev, dtype=torch.bfloat16, generator=torch.Generator(device=DEV).manual_seed(1))
    seen, own = [], []
    orig = torch.nn.functional.scaled_dot_product_attention
    keep_fwd, keep_bwd = ops.causal_attention, ops.causal_attention_bwd

    def spy(q, *a, **k):
        seen.append(int(q.shape[-1]))
        return orig(q, *a, **k)

    def own_fwd(q, *a, **k):
        own.append(("fwd",) + tuple(q.shape))
        return keep_fwd(q, *a, **k)

    def own_bwd(q, *a, **k):
        own.append(("bwd",) + tuple(q.shape))
        return keep_bwd(q, *a, **k)

    out = {}
    torch.nn.functional.scaled_dot_product_attention = spy
    ops.causal_attention, ops.causal_attention_bwd = own_fwd, own_bwd
    try:
        for route in ("own", "padded", "plain"):
            hf.pad_vision_heads = route != "plain"
            pa.OWN_TOWER_72 = route == "own"
            seen.clear()
            own.clear()
            img = image.clone().requires_grad_()
            feats = hf.image_features(img)
            (g,) = torch.autograd.grad((feats[0].to(torch.bfloat16) * w).sum().float(), img)
            with torch.no_grad():
                f2 = hf.image_features(image)
            out[route] = (feats.detach().float(), g.float(), f2.float(), list(seen), list(own))
    finally:
        torch.nn.functional.scaled_dot_product_attention = orig
        ops.causal_attention, ops.causal_attention_bwd = keep_fwd, keep_bwd
        pa.OWN_TOWER_72 = True
    assert out["own"][3] == [] and out["own"][4] == [("fwd", 4096, 16, 72)] * 3 + [("bwd", 4096, 16, 72)] * 3 + [("fwd", 4096, 16, 72)] * 3
    assert out["padded"][3] == [128] * 3 + [96] * 3 and out["plain"][3] == [72] * 6 and out["padded"][4] == out["plain"][4] == []
    assert model.config.vision_config._attn_implementation == "sdpa"            # restored
    f0, g0, n0 = out["plain"][:3]
    for route in ("own", "padded"):
        f1, g1, n1 = out[route][:3]
        for a, b in ((f1, f0), (n1, n0), (g1, g0)):
            assert float((a - b).abs().max()) <= 4e-2 * float(b.abs().max()), (route, float((a - b).abs().max()) / float(b.abs().max()))
        big = g0.abs() > 0.1 * g0.abs().max()
        assert float((torch.sign(g1[big]) == torch.sign(g0[big])).float().mean()) > 0.98, route


@pytest.mark.parametrize("graphs", [False, True])
def test_gradient_pass_reusing_the_scoring_prefix_equals_the_full_pass(graphs):
    """Joint mode, LLaVA layout: the prefix pass of candidate scoring run with autograd + the 44 tokens behind it
    (attack._GradPrefix) against the full 1-sequence forward/backward -- token gradients, pixel gradients and loss;
    eagerly and as two hipGraphs sharing one autograd graph, over three images and suffixes in a row (a replay must
    see the new image, not the captured one).  fp32 tiny model: the two are the same maths."""
    from bimodalattack_amd import BimodalAttackConfig, synthetic as S
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions
    model, tok, proc, image = S.tiny_case("llava", device=DEV)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, topk=16, seed=1, verbosity="ERROR", pgd_attack=True,
                              gcg_attack=True, joint_eval=True, optim_str_init=S.TINY_OPTIM_INIT,
                              images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, S.Normalize(S.CLIP_MEAN, S.CLIP_STD),
                        EngineOptions.from_env(save_images=False, graph_gradient=graphs, graph_scoring=graphs, strict=True))
    atk._prepare_prompt("tell me a story", "Sure here is")
    assert atk._gp_enabled()
    g = torch.Generator().manual_seed(3)
    ids0 = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(DEV)
    for it in range(3):
        ids = ids0.clone()
        ids[0, it] = 17 + it
        img = (image.to(DEV) + 0.05 * it * torch.rand(image.shape, generator=g).to(DEV)).clamp(0, 1)
        want_tok, want_img, want_loss = atk._gradient_eager(ids, img.detach().clone().requires_grad_())
        got_tok, got_img, got_loss = atk.compute_gradient(ids, img)
        assert atk._gp not in (None, False) and (atk._gp.g2 is not None) == graphs
        assert abs(float(got_loss) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
        for got, want in ((got_tok, want_tok), (got_img, want_img)):        # fp32 sums in another order: 1e-5 of the scale
            assert float((got.float() - want.float()).abs().max()) <= 1e-4 * float(want.float().abs().max())
        # the token gradient alone (second pass of a non-joint step): prefix detached, same numbers
        tok_only, none_img, loss_only = atk.compute_gradient(ids, img, tokens_only=True)
        assert none_img is None and abs(float(loss_only) - float(want_loss)) <= 1e-5 * abs(float(want_loss))
        assert float((tok_only.float() - want_tok.float()).abs().max()) <= 1e-4 * float(want_tok.float().abs().max())
        assert (atk._gp.g3 is not None) == graphs
        # scoring on the next image goes through the same object and serves the prefix of the pass after it
        nxt = (img + 0.01).clamp(0, 1)
        feats = atk.scoring_features(nxt)
        assert atk._gp.serves(("before_img", "image", "before_suffix"), feats) and atk._gp.current is nxt
        with torch.no_grad():
            assert torch.allclose(feats, atk.hf.image_features(nxt), rtol=1e-5, atol=1e-6)
    assert atk.fallbacks == {}
    assert ("grad_tail" in atk.graphs_captured) == graphs


def test_prefix_reusing_gradient_pass_at_7b_width_bf16():
    """LLaVA-1.5-7B width, 2 layers, bf16, the 643-token image prompt: scoring prefix (599 rows, with history) + the
    44 tokens behind it against the full pass -- same function, other GEMM shapes and one library attention over
    [prefix ; tail] keys: loss, token gradient and pixel gradient agree to bf16 noise, as two hipGraphs."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import build_plugins
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.config import EngineOptions

    dev = torch.device(DEV)
    model, tok, proc, messages, goal, target, image, norm = build_plugins("joint", dev, torch.bfloat16, 2)
    cfg = BimodalAttackConfig(num_steps=1, search_width=8, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=True,
                              joint_eval=True, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, strict=True))
    atk._prepare_prompt(messages, target)
    assert atk._gp_enabled()
    ids = tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(dev)
    t0, i0, l0 = atk._gradient_eager(ids, image.detach().clone().requires_grad_())
    for _ in range(2):                                        # capture, then a replay
        t1, i1, l1 = atk.compute_gradient(ids, image)
    assert atk._gp.g2 is not None and atk._gp.P == 599 and atk.fallbacks == {}
    t0, i0, t1, i1 = t0.float(), i0.float(), t1.float(), i1.float()
    assert abs(float(l1) - float(l0)) <= 2e-2 * abs(float(l0))
    assert float((t1 - t0).abs().max()) <= 5e-2 * float(t0.abs().max())
    assert float((i1 - i0).abs().max()) <= 5e-2 * float(i0.abs().max())
    big = i0.abs() > 0.05 * i0.abs().max()
    assert float((torch.sign(i1[big]) == torch.sign(i0[big])).float().mean()) > 0.98


def test_tensor_parallel_gradient_pass_replays_from_a_hipgraph_under_rccl():
    """VERDICT r3 item 6(b).  ``tp_gradient`` (the batch-1 gradient pass cut over the ranks, two all-reduces per decoder
    layer and direction) captured into a hipGraph WITH its RCCL collectives inside, and replayed: one rank (RCCL cannot
    put two on one GPU) with the sharder's collective paths forced on, so every all-reduce is a real RCCL call on a
    live communicator.  The graph is captured (opt-in: `tp_graph`; its outcome is agreed on by an all-reduce(MIN) over the
    ranks), no fallback is taken, and the run equals the same attack without a process group.  Still off by default: its
    speed has never met xGMI."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = r"""
import json, os, tempfile, torch, torch.distributed as dist
from bimodalattack_amd import BimodalAttackConfig, synthetic as S
from bimodalattack_amd.attack import BimodalAttack
from bimodalattack_amd.config import EngineOptions
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
out = {}
for grouped in (False, True):
    if grouped:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        t = torch.ones(4, device=dev); dist.all_reduce(t)
    model, tok, proc, image = S.tiny_case("llava", device="cuda:0")
    cfg = BimodalAttackConfig(num_steps=4, search_width=16, topk=8, pgd_attack=True, gcg_attack=True, joint_eval=False,
                              eps=64 / 255, alpha=4 / 255, seed=3, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT,
                              early_stop=False, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, S.Normalize(S.CLIP_MEAN, S.CLIP_STD),
                        EngineOptions.from_env(rng_device="cpu", strict=True, tp_gradient="graph" if grouped else False))
    if grouped:
        atk.shard.enabled = True
    res = atk.run("tell me a story", "tell me a story", "Sure here is a story", image)
    out[grouped] = dict(losses=res.losses, strings=res.strings, graphs=atk.graphs_captured, fallbacks=atk.fallbacks,
                        tp=bool(atk._tp_checked))
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps({str(k): v for k, v in out.items()}))
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RESULT " in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    out = json.loads(r.stdout.split("RESULT ", 1)[1].splitlines()[0])
    a, b = out["False"], out["True"]
    assert a["fallbacks"] == {} and b["fallbacks"] == {} and b["tp"] is True
    assert "gradient_tp" in b["graphs"] and "gradient" not in b["graphs"]
    assert a["strings"] == b["strings"]
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=1e-4)


def test_engine_with_hipgraphs_under_a_live_rccl_group():
    """What the driver's multi-GPU run does that no other test does: hipGraph captures while an RCCL process group is
    alive (its watchdog thread polls events of earlier collectives) and real RCCL collectives every step.  One rank
    (two cannot share a GPU under RCCL) with the sharder's collective paths forced on: a joint attack on the tiny
    LLaVA captures its graphs, takes no fallback, and equals the same attack without a process group."""
    import socket
    import subprocess
    import sys
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    code = r"""
import json, os, tempfile, torch, torch.distributed as dist
from bimodalattack_amd import BimodalAttackConfig, synthetic as S
from bimodalattack_amd.attack import BimodalAttack
from bimodalattack_amd.config import EngineOptions
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
out = {}
for grouped in (False, True):
    if grouped:
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        t = torch.ones(4, device=dev); dist.all_reduce(t)                 # the communicator and its watchdog are live
    model, tok, proc, image = S.tiny_case("llava", device="cuda:0")
    cfg = BimodalAttackConfig(num_steps=4, search_width=16, topk=8, pgd_attack=True, gcg_attack=True, joint_eval=True,
                              eps=64 / 255, alpha=4 / 255, seed=3, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT,
                              early_stop=False, images_folder=tempfile.mkdtemp())
    atk = BimodalAttack(model, tok, proc, cfg, S.Normalize(S.CLIP_MEAN, S.CLIP_STD),
                        EngineOptions.from_env(rng_device="cpu", strict=True))
    if grouped:
        atk.shard.enabled = True                                           # world 1, collectives on: broadcast + all-gather per step
    res = atk.run("tell me a story", "tell me a story", "Sure here is a story", image)
    out[grouped] = dict(losses=res.losses, strings=res.strings, graphs=atk.graphs_captured, fallbacks=atk.fallbacks,
                        collectives=atk.shard.n_collectives)
if dist.is_initialized():
    dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps({str(k): v for k, v in out.items()}))
"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1",
               PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RESULT " in r.stdout, (r.stdout[-1000:], r.stderr[-3000:])
    import json
    out = json.loads(r.stdout.split("RESULT ", 1)[1].splitlines()[0])
    a, b = out["False"], out["True"]
    assert a["fallbacks"] == {} and b["fallbacks"] == {}
    assert "grad_tail" in b["graphs"] and "grad_prefix" in b["graphs"] and b["collectives"] >= 2 * 4
    assert a["strings"] == b["strings"]
    np.testing.assert_allclose(a["losses"], b["losses"], rtol=1e-5)


def test_zz_every_golden_comparison_of_the_session_is_audited():
    """The audit of `check_against_golden`'s escape hatch over the WHOLE session (the 13 base trajectories and the ~90
    parametrised restructurings / sharded runs of this file): how many comparisons ran to their last step, how many ended
    at a near-tied winner and where.  At least 80 % must run to the end, and the steps actually compared must be at least
    85 % of the steps run."""
    if len(_ALL_RUNS) < 20:
        pytest.skip("too few golden comparisons in this session to audit")
    early = [(n, d, s_) for n, d, s_ in _ALL_RUNS if d is not None]
    steps = sum(s_ for _, _, s_ in _ALL_RUNS)
    compared = sum((s_ if d is None else d + 1) for _, d, s_ in _ALL_RUNS)
    by_case = {}
    for n, d, _ in early:
        by_case.setdefault(n, []).append(d)
    line = (f"golden comparisons: {len(_ALL_RUNS)}, to the last step: {len(_ALL_RUNS) - len(early)}; steps compared {compared} of {steps}; "
            f"ended at a near-tied winner: {by_case}")
    print(line)
    _audit_line(line)
    assert len(early) <= 0.2 * len(_ALL_RUNS), by_case
    assert compared >= 0.85 * steps
