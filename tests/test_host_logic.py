"""Host-side logic of the product (no GPU): API surface, segment layouts, width
schedule, tokenizer helpers, chunk planning, HF adapter restructurings."""

import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import kernels as K


def test_api_surface_matches_reference_fields():
    import bimodalattack_amd as pkg
    from bimodalattack_amd.config import BimodalAttackConfig, BimodalAttackResult, GCGConfig
    assert pkg.__all__[:3] == ["BimodalAttackConfig", "run", "BimodalAttackResult"]
    assert GCGConfig is BimodalAttackConfig
    cfg = {f.name: f.default for f in dataclasses.fields(BimodalAttackConfig)}
    # reference bimodal_attack.py:42-70, in order
    assert list(cfg) == ["num_steps", "optim_str_init", "search_width", "batch_size", "topk", "n_replace",
                         "buffer_size", "use_mellowmax", "mellowmax_alpha", "early_stop", "allow_non_ascii",
                         "filter_ids", "add_space_before_target", "seed", "verbosity", "dynamic_search",
                         "min_search_width", "alpha", "eps", "pgd_attack", "gcg_attack", "debug_output",
                         "joint_eval", "experiment_folder", "images_folder", "pgd_after_gcg", "model"]
    assert (cfg["num_steps"], cfg["search_width"], cfg["topk"], cfg["n_replace"], cfg["buffer_size"]) == (250, 512, 256, 1, 0)
    assert (cfg["alpha"], cfg["eps"], cfg["min_search_width"], cfg["filter_ids"], cfg["gcg_attack"]) == (0.01, 0.1, 32, True, True)
    assert cfg["optim_str_init"] == " ".join(["x"] * 19) and cfg["seed"] is None and cfg["batch_size"] is None
    assert [f.name for f in dataclasses.fields(BimodalAttackResult)] == [
        "best_loss", "best_string", "losses", "strings", "adversarial_suffixes", "model_outputs",
        "gradient_times", "sampling_times", "loss_times", "pgd_times", "total_times"]
    import inspect
    assert list(inspect.signature(pkg.run).parameters)[:9] == [
        "model", "tokenizer", "processor", "messages", "goal", "target", "image", "config", "normalize"]


def test_layout_equals_oracle_for_every_flag_combination():
    from bimodalattack_amd.layout import dynamic_width, segment_order, split_at_suffix
    for mt in ("llava", "gemma3", "opt", "llama"):
        for mode in ("pgd", "gcg", "gcg_pgd"):
            for single in (False, True):
                for nj in (False, True):
                    for nt in (False, True):
                        kw = dict(single=single, no_joint_eval=nj, no_target=nt)
                        try:
                            want = K.segment_order(mode, mt, **kw)
                        except (ValueError, AssertionError) as e:
                            with pytest.raises(type(e)):
                                segment_order(mode, mt, **kw)
                            continue
                        assert segment_order(mode, mt, **kw) == want, (mt, mode, kw)
    with pytest.raises(ValueError):
        segment_order("other", "llava")
    for i in range(0, 600, 7):
        assert dynamic_width(i, 512, 600, 128, True) == K.dynamic_width(i, 512, 600, 128, True)
    assert dynamic_width(5, 512, 600, 128, False) == 512
    assert split_at_suffix(["before_img", "image", "before_suffix", "optim", "after", "target"]) == (
        ["before_img", "image", "before_suffix"], ["optim", "after", "target"])
    assert split_at_suffix(["optim", "x"]) == ([], ["optim", "x"])


def test_tokenizer_helpers_equal_oracle(golden_dir):
    from bimodalattack_amd import synthetic as S
    from bimodalattack_amd.utils import INIT_CHARS, filter_ids, get_nonascii_toks
    z = np.load(os.path.join(golden_dir, "g6_tokens.npz"))
    tok = S.build_tokenizer(S.TINY_WORDS, S.TINY_NONASCII, S.TINY_UNRT)
    assert np.array_equal(get_nonascii_toks(tok).numpy(), z["not_allowed"])
    kept = filter_ids(torch.from_numpy(z["ids"]), tok)
    assert np.array_equal(kept.numpy(), z["kept"])
    assert np.array_equal(kept.numpy(), K.filter_ids(z["ids"], tok))
    all_ok = torch.from_numpy(z["kept"])
    assert filter_ids(all_ok, tok) is all_ok                 # nothing dropped: no copy
    with pytest.raises(RuntimeError, match="No token sequences are the same"):
        filter_ids(torch.full((3, 4), tok.convert_tokens_to_ids("ab0 cd")), tok)
    assert len(INIT_CHARS) == 22 and INIT_CHARS[0] == "." and INIT_CHARS[-1] == "z"


def test_plan_chunk():
    from bimodalattack_amd.utils import is_oom, plan_chunk
    assert plan_chunk(512, 45, 21, 524288, 219136, 200 << 30, user_batch=64) == 64
    assert plan_chunk(10, 45, 21, 524288, 219136, 200 << 30, user_batch=64) == 10
    assert plan_chunk(512, 45, 21, 524288, 219136, 200 << 30) == 512
    c = plan_chunk(512, 45, 599, 524288, 219136, 100 << 30)     # joint layout: prefix copies dominate
    assert 1 <= c < 512 and c * (45 * 219136 + 644 * 524288) <= 50 << 30
    assert plan_chunk(512, 45, 599, 524288, 219136, 1 << 20) == 1
    assert is_oom(RuntimeError("HIP out of memory. Tried to allocate")) and is_oom(RuntimeError("CUDA out of memory."))
    assert not is_oom(RuntimeError("shape mismatch")) and not is_oom(ValueError("out of memory"))


def test_engine_options_from_env(monkeypatch):
    from bimodalattack_amd.config import EngineOptions
    monkeypatch.setenv("BMA_RNG_DEVICE", "cpu")
    monkeypatch.setenv("BMA_PREFIX_REUSE", "0")
    monkeypatch.setenv("BMA_CHUNK", "17")
    o = EngineOptions.from_env(save_images=False)
    assert (o.rng_device, o.prefix_reuse, o.chunk, o.save_images) == ("cpu", False, 17, False)
    with pytest.raises(TypeError):
        EngineOptions.from_env(nonsense=1)
    with pytest.raises(ValueError):
        EngineOptions.from_env(rng_device="tpu")
    # round 5: the options are what a test or bench.py flips -- at most 30 of them; every boolean one has its BMA_<NAME>
    import dataclasses
    names = [f.name for f in dataclasses.fields(EngineOptions)]
    assert len(names) <= 30 and set(EngineOptions._BOOLS) <= set(names)
    for gone in ("skinny_gemm", "mid_gemm", "fuse_qkv", "graph_prefix", "graph_rescore", "emulate_world", "score_log", "group", "tp_graph"):
        assert gone not in names
    monkeypatch.setenv("BMA_OWN_B1_KERNELS", "0")
    monkeypatch.setenv("BMA_GRAPH_SCORING", "false")
    monkeypatch.setenv("BMA_TP_GRADIENT", "graph")
    monkeypatch.setenv("BMA_FILTER_FIRST", "1")
    o = EngineOptions.from_env()
    assert (o.own_b1_kernels, o.graph_scoring, o.tp_gradient, o.filter_first) == (False, False, "graph", True)
    assert EngineOptions.from_env(tp_gradient=True).tp_gradient is True
    with pytest.raises(ValueError):
        EngineOptions.from_env(tp_gradient="eager")
    with pytest.raises(TypeError):
        EngineOptions.from_env(_BOOLS=())


@pytest.mark.parametrize("kind", ["opt", "llava", "gemma3"])
def test_hf_adapter_restructurings_are_identical_maths(kind):
    """Target rows only, last token dropped, shared-prefix keys/values: same logits as
    the reference's full forward, to fp32 rounding."""
    from bimodalattack_amd import synthetic as S
    from bimodalattack_amd.hf_adapter import HFAdapter
    torch.manual_seed(0)
    model, tok, proc, image = S.tiny_case(kind)
    ad = HFAdapter(model, proc, S.Normalize(S.CLIP_MEAN, S.CLIP_STD))
    D = ad.embedding.weight.shape[1]
    B, P, L, T = 5, 7, 9, 4
    prefix, tail = torch.randn(1, P, D), torch.randn(B, L, D)
    full = torch.cat([prefix.expand(B, -1, -1), tail], 1)
    with torch.no_grad():
        ref = model(inputs_embeds=full).logits[:, -T - 1:-1]
        assert torch.equal(ad.target_logits(full, T, rows_only=False), ref)
        np.testing.assert_allclose(ad.target_logits(full[:, :-1], T), ref, rtol=1e-4, atol=1e-4)
        cache = ad.build_prefix(prefix)
        for b in (B, 2):
            got = ad.target_logits(tail[:b, :-1], T, cache=ad.expand_prefix(cache, b))
            np.testing.assert_allclose(got, ref[:b], rtol=1e-4, atol=1e-4)
    if kind != "opt":
        f = ad.image_features(image)
        assert f.dim() == 3 and f.shape[0] == 1 and f.shape[2] == D
    assert (ad.emb_scale != 1.0) == (kind == "gemma3")


def test_keep_index_entries_outlive_other_shapes():
    """A captured hipGraph holds the raw pointer of the index tensor it was captured with: the
    adapter must hand back the SAME tensor for a shape however many other shapes were asked
    for in between (an evicted entry was a use-after-free on graph replay)."""
    from bimodalattack_amd import synthetic as S
    from bimodalattack_amd.hf_adapter import HFAdapter
    model, tok, proc, _ = S.tiny_case("opt")
    hf = HFAdapter(model, proc)
    a = hf._keep_index(40, 5, "cpu")
    ptr = a.data_ptr()
    for L in range(10, 30):
        hf._keep_index(L, 5, "cpu")
    b = hf._keep_index(40, 5, "cpu")
    assert b is a and b.data_ptr() == ptr and b.tolist() == [35, 36, 37, 38, 39]


def test_capturable_normalize_is_bit_identical_or_steps_aside():
    from bimodalattack_amd import synthetic as S
    from bimodalattack_amd.hf_adapter import CapturableNormalize
    x = torch.rand(1, 3, 8, 8)
    ref = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)
    n = CapturableNormalize(ref)
    assert torch.equal(n(x), ref(x)) and n.ok is True
    assert torch.equal(n(x * 0.5), ref(x * 0.5))

    class ListStyle:                       # torchvision keeps Python lists
        mean, std = list(S.CLIP_MEAN), list(S.CLIP_STD)

        def __call__(self, t):
            m = torch.as_tensor(self.mean, dtype=t.dtype).view(-1, 1, 1)
            s = torch.as_tensor(self.std, dtype=t.dtype).view(-1, 1, 1)
            return t.clone().sub_(m).div_(s)
    n = CapturableNormalize(ListStyle())
    assert torch.equal(n(x), ListStyle()(x)) and n.ok is True

    class Liar:                            # exposes mean/std but computes something else
        mean, std = [0.0, 0.0, 0.0], [1.0, 1.0, 1.0]

        def __call__(self, t):
            return t * 2
    n = CapturableNormalize(Liar())
    assert torch.equal(n(x), x * 2) and n.ok is False and torch.equal(n(x), x * 2)
    n = CapturableNormalize(lambda t: t + 1)          # no mean/std: passed through
    assert n.ok is False and torch.equal(n(x), x + 1)


def test_ragged_plan_invariants():
    """layout.ragged_plan: every computed token appears once, slots in front of the first
    replaced position read the parent's rows, the fixed budget is met exactly, and a draw that
    cannot fit is refused."""
    from bimodalattack_amd.layout import first_diff_stats, ragged_plan, ragged_rows
    mean, var = first_diff_stats(20, 1)
    assert abs(mean - 9.5) < 1e-12 and abs(var - (20 ** 2 - 1) / 12) < 1e-9
    assert first_diff_stats(20, 2)[0] < mean and first_diff_stats(5, 9) == (0.0, 0.0)
    rng = np.random.default_rng(0)
    for m, n_opt, L, T, r in [(512, 20, 45, 20, 1), (33, 6, 14, 5, 2), (7, 4, 8, 5, 1), (2, 1, 3, 3, 1)]:
        parent = rng.integers(0, 50, n_opt)
        cand = np.tile(parent, (m, 1))
        for i in range(m):
            for q in rng.choice(n_opt, size=min(r, n_opt), replace=False):
                cand[i, q] = parent[q] + 100 + i
        cand[m // 2] = parent                                  # a candidate equal to its parent
        first = np.where((cand != parent).any(1), (cand != parent).argmax(1), n_opt - 1)
        n_rows = ragged_rows(n_opt + int((L - first).sum()), n_opt + m * L)      # the grid point the engine builds
        assert n_rows <= n_opt + m * L
        plan = ragged_plan(cand, parent, L, T, 7, n_rows, dedup=False)
        N, flat, p = plan["N"], plan["flat"], plan["p"]
        assert len(flat) == N == len(plan["pos"]) and (p <= first).all() and (p >= 0).all()
        assert len(set(flat.tolist())) == N
        assert (flat[:n_opt] == m * L + np.arange(n_opt)).all() and (plan["pos"][:n_opt] == 7 + np.arange(n_opt)).all()
        assert (plan["pos"] == 7 + flat % L).all()
        qs, ks = plan["q_src"].reshape(m + 1, L), plan["kv_src"].reshape(m + 1, L)
        for i in range(m):
            for j in range(L):
                if j >= p[i]:
                    assert flat[qs[i, j]] == i * L + j and ks[i, j] == qs[i, j]
                else:
                    assert ks[i, j] == j and cand[i, j] == parent[j] and flat[qs[i, j]] // L == i
        assert (ks[m, :n_opt] == np.arange(n_opt)).all() and (qs[m] < n_opt).all()
        keep = plan["keep"].reshape(m, T)
        assert (flat[keep] == (np.arange(m)[:, None] * L + (L - T) + np.arange(T)[None, :])).all()
    assert ragged_plan(cand, parent, L, T, 7, n_opt + 1) is None

    # duplicates are computed once: every input candidate still gets its T rows
    from bimodalattack_amd.layout import expected_unique
    m, n_opt, L, T = 40, 5, 12, 4
    parent = np.arange(n_opt)
    base = np.tile(parent, (8, 1))
    base[np.arange(8), np.arange(8) % n_opt] += 50 + np.arange(8)
    cand = base[rng.integers(0, 8, m)]                         # 40 draws of 8 distinct candidates
    plan = ragged_plan(cand, parent, L, T, 3, n_opt + 8 * L)
    assert plan["m"] == len(np.unique(cand, axis=0)) <= 8 and plan["m_out"] == m
    keep, flat = plan["keep"].reshape(m, T), plan["flat"]
    for i in range(m):
        u = int(np.where((plan["cand"] == cand[i]).all(1))[0][0])
        assert (flat[keep[i]] == u * L + (L - T) + np.arange(T)).all()
    mean, var = expected_unique(512, 19, 1, 256)
    assert 480 < mean < 492 and 0 < var < 40
    assert expected_unique(512, 19, 2, 256) == (512.0, 0.0) or expected_unique(512, 19, 2, 256)[0] > 511.9

    # the row count the engine builds: what the draw needs, on the next point of a coarse grid
    assert [ragged_rows(v, 10 ** 6) for v in (17029, 17152, 17153, 4300, 2150, 700)] == [17152, 17152, 17408, 4352, 2176, 704]
    assert ragged_rows(17029, 17000) == 17000
    plan = ragged_plan(cand, parent, L, T, 3)
    assert plan["N"] == ragged_rows(plan["needed"], n_opt + plan["m"] * L) >= plan["needed"]
    with pytest.raises(ValueError):
        ragged_plan(np.zeros((2, 4), int), np.zeros(4, int), 5, 3, 0, 14)     # target rows would precede the suffix end


def test_row_count_grid_covers_what_steps_meet():
    """tools/tune_rows.py walks the grid points a run can meet; simulate draws of the bench configuration
    (512 candidates, 19 suffix positions, top-256) and check that every step's row count -- one GPU, and the
    dealt per-rank share of 2 / 4 / 8 -- is in the tuned set."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    from tune_rows import row_counts
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.layout import ragged_plan
    m, n_opt, L, T, topk = 512, 19, 44, 20, 256
    tuned = set(row_counts(m, n_opt, L, 1, topk))
    rng = np.random.default_rng(5)
    for _ in range(40):
        parent = rng.integers(0, 32000, n_opt)
        cand = np.tile(parent, (m, 1))
        pos = rng.integers(0, n_opt, m)
        cand[np.arange(m), pos] = 40000 + pos * topk + rng.integers(0, topk, m)     # token = f(position, rank)
        assert ragged_plan(cand, parent, L, T, 21)["N"] in tuned
        uniq = np.unique(cand, axis=0)
        diff = uniq != parent[None, :]
        first = np.where(diff.any(1), diff.argmax(1), n_opt - 1)
        by_cost = np.argsort(first, kind="stable")
        for world in (2, 4, 8):
            assert BimodalAttack._dealt_rows((by_cost, None, len(uniq), first), world, L, n_opt) in tuned, world


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` bare (no torch.distributed.run around it, as the driver may call it) starts N
    rank processes itself, as children, before anything touches a GPU; under a launcher it does not nest."""
    import json
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BMA_BENCH_LAUNCH_PROBE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "3", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1, 2] and all(l["n_gpus"] == 3 and l["master"] == "127.0.0.1" for l in lines)
    # already under a launcher (RANK set): no second level of processes
    r = subprocess.run([sys.executable, os.path.join(repo, "bench.py"), "--gpus", "2"],
                       env=dict(env, RANK="1", WORLD_SIZE="2", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="1"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and json.loads(r.stdout.strip())["rank"] == 1


def test_unique_rows_is_exact_and_ordered_by_first_appearance(monkeypatch):
    """layout.unique_rows: np.unique(axis=0)'s groups, in order of first appearance; a hash collision (forced
    here by a constant hash) falls back to the exact path instead of merging different rows."""
    from bimodalattack_amd import layout
    rng = np.random.default_rng(3)
    for _ in range(50):
        a = rng.integers(0, 3, (int(rng.integers(1, 300)), int(rng.integers(1, 22))))
        u, inv = layout.unique_rows(a)
        assert np.array_equal(u[inv], a) and len(u) == len(np.unique(a, axis=0))
        firsts = [int(np.where((a == r).all(1))[0][0]) for r in u]
        assert firsts == sorted(firsts)
    big = rng.integers(0, 2 ** 40, (512, 19))
    big[100] = big[7]
    u, inv = layout.unique_rows(big)
    assert len(u) == 511 and inv[100] == inv[7] == 7
    monkeypatch.setattr(layout, "_HASH_MUL", np.zeros(64, dtype=np.uint64))       # every row hashes to 0
    a = rng.integers(0, 5, (40, 6))
    u, inv = layout.unique_rows(a)
    assert np.array_equal(u[inv], a) and len(u) == len(np.unique(a, axis=0))


def test_padded_heads_attention_is_the_same_attention():
    """Zero columns appended to q, k, v up to a library-friendly head width, dropped from the output: the same
    attention and the same gradients (fp32 on the CPU here; the GPU twin checks bf16 at SigLIP's size)."""
    import torch
    from bimodalattack_amd import prefix_attention as pa
    assert pa.padded_width(72, True) == 128 and pa.padded_width(72, False) == 96
    assert pa.padded_width(64, True) == 64 and pa.padded_width(128, False) == 128 and pa.padded_width(160, True) == 160
    g = torch.Generator().manual_seed(0)
    q, k, v = (torch.randn(2, 3, 40, 24, generator=g, requires_grad=True) for _ in range(3))
    old = pa.PAD_HEADS_MIN_TOKENS
    outs = []
    try:
        for thr in (0, 10 ** 9):
            pa.PAD_HEADS_MIN_TOKENS = thr
            o, w = pa.padded_heads_attention(None, q, k, v, None, scaling=24 ** -0.5)
            assert w is None and o.shape == (2, 40, 3, 24)
            grads = torch.autograd.grad(o.square().sum(), (q, k, v))
            outs.append((o.detach(), grads))
            with torch.no_grad():
                o2, _ = pa.padded_heads_attention(None, q, k, v, None, scaling=24 ** -0.5)
            assert torch.allclose(o2, o, atol=1e-6)
    finally:
        pa.PAD_HEADS_MIN_TOKENS = old
    (o1, g1), (o0, g0) = outs
    assert torch.allclose(o1, o0, atol=1e-6)
    for a, b in zip(g1, g0):
        assert torch.allclose(a, b, atol=1e-5)


# ------------------------------------------------------------------ round 3 host logic
def test_decoder_layer_structure_is_read_from_source():
    """The fused layer forward (residual add + norm in one pass) is installed only on decoder layers whose forward is,
    statement for statement, a structure it restates: llama / mistral / qwen2 share one, gemma3 has its four norms,
    anything else -- another family, or a subclass that changed a line -- keeps HuggingFace's own code."""
    from bimodalattack_amd import fused, synthetic as S
    from transformers.models.gemma3.modeling_gemma3 import Gemma3DecoderLayer
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer
    from transformers.models.mistral.modeling_mistral import MistralDecoderLayer
    from transformers.models.opt.modeling_opt import OPTDecoderLayer
    from transformers.models.qwen2.modeling_qwen2 import Qwen2DecoderLayer
    kinds = {c.__name__: fused._layer_kind(c.__new__(c)) for c in
             (LlamaDecoderLayer, MistralDecoderLayer, Qwen2DecoderLayer, Gemma3DecoderLayer, OPTDecoderLayer)}
    assert kinds == {"LlamaDecoderLayer": "llama", "MistralDecoderLayer": "llama", "Qwen2DecoderLayer": "llama",
                     "Gemma3DecoderLayer": "gemma", "OPTDecoderLayer": None}

    class Scaled(LlamaDecoderLayer):                      # one changed statement: not the structure any more
        def forward(self, hidden_states, **kwargs):
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
            hidden_states, _ = self.self_attn(hidden_states=hidden_states, **kwargs)
            hidden_states = residual + 0.5 * hidden_states
            residual = hidden_states
            hidden_states = self.post_attention_layernorm(hidden_states)
            hidden_states = self.mlp(hidden_states)
            hidden_states = residual + hidden_states
            return hidden_states

    assert fused._layer_kind(Scaled.__new__(Scaled)) is None
    for kind, want in (("llava", "llama"), ("gemma3", "gemma"), ("opt", None)):
        model = S.tiny_case(kind)[0]
        f = fused.FusedInference(model)
        assert [k for _, k, _ in f.layers] == ([want] * 2 if want else [])
        if want:
            # every layer hands over to the NEXT layer's input norm, the last one to the stack's final norm
            (l0, _, n0), (l1, _, n1) = f.layers
            assert n0 is l1.input_layernorm and type(n1).__name__.endswith("RMSNorm") and n1 is not l1.input_layernorm
            assert f.tp_ok(2) and not f.tp_ok(3)
    # the context patches and restores on a CPU model too (the kernels step aside, the structure runs)
    model = S.tiny_case("llava")[0]
    f = fused.FusedInference(model)
    x = torch.randn(2, 7, model.get_input_embeddings().weight.shape[1])
    lm = model.model.language_model
    want = lm(inputs_embeds=x, use_cache=False).last_hidden_state
    with f:
        got = lm(inputs_embeds=x, use_cache=False).last_hidden_state
    assert torch.allclose(got, want, rtol=1e-5, atol=1e-6) and not any("forward" in m.__dict__ for m in model.modules())


def test_image_features_across_transformers_return_types():
    """The reference pins transformers 4.50.2 (/root/reference/requirements.txt:4), where ``get_image_features`` returns
    the (1,N,D) tensor the reference indexes (bimodal_attack.py:528-536, :877-884, :972-979); this image has 5.x, where
    it returns an output object.  ``features_tensor`` takes every form a release has used, and ``HFAdapter`` calls the
    plug-in with the reference's own keyword arguments for each processor family."""
    from types import SimpleNamespace as NS
    from bimodalattack_amd.hf_adapter import HFAdapter, features_tensor
    t = torch.arange(24.0).reshape(1, 4, 6)
    assert features_tensor(t) is t                                                          # 4.50.2
    assert torch.equal(features_tensor([t[0]]), t) and torch.equal(features_tensor((t[0], t[0] + 1)), torch.stack([t[0], t[0] + 1]))
    assert features_tensor(NS(pooler_output=t)) is t                                        # 5.x, Gemma-3
    assert torch.equal(features_tensor(NS(pooler_output=[t[0]])), t)                        # 5.x, LLaVA: a list per image
    hidden = torch.zeros(1, 9, 5)
    assert features_tensor((hidden, t)) is t                                                # 5.x output object as a tuple
    assert torch.equal(features_tensor((hidden, [t[0]])), t)
    with pytest.raises(TypeError):
        features_tensor(NS(last_hidden_state=hidden))
    with pytest.raises(TypeError):
        features_tensor(())

    calls = []

    class Stub450(torch.nn.Module):
        """The plug-in surface of a 4.50.2 model: get_image_features(...) -> Tensor."""

        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Embedding(11, 6)
            self.config = NS(model_type="llava")
            self.dtype, self.device = torch.float32, torch.device("cpu")

        def get_input_embeddings(self):
            return self.emb

        def forward(self, inputs_embeds=None, logits_to_keep=0, past_key_values=None, use_cache=None):
            raise AssertionError("not called here")

        def get_image_features(self, pixel_values=None, **kw):
            calls.append(sorted(kw))
            return pixel_values.mean(dim=(2, 3), keepdim=False).reshape(1, 1, 3).repeat(1, 4, 2)

    img = torch.rand(1, 3, 8, 8)
    for proc_name, want_kw in (("LlavaProcessor", ["vision_feature_layer", "vision_feature_select_strategy"]),
                               ("Gemma3Processor", [])):
        proc = type(proc_name, (), {})()
        hf = HFAdapter(Stub450(), proc, lambda x: x * 2.0)
        got = hf.image_features(img)
        assert got.shape == (1, 4, 6) and torch.allclose(got[0, 0, :3], (img * 2.0).mean(dim=(2, 3))[0])
        assert calls[-1] == want_kw, (proc_name, calls[-1])
        assert hf.emb_scale == 1.0 and hf.has_logits_to_keep and hf.projector_norms() == []


def test_tower_layer_admission_compares_syntax_trees():
    """the vision tower's fused add + LayerNorm forward is admitted by comparing the encoder layer's forward as a
    SYNTAX TREE -- signature (self, hidden_states, attention_mask, **kwargs) and every statement, the attention call's keyword
    list included -- not by filtering lines: the installed CLIP / SigLIP layers pass; the same block with one more keyword in
    the attention call, another signature or a statement more is refused (the restated forward would drop it silently)."""
    import torch
    from transformers import CLIPVisionConfig, SiglipVisionConfig
    from transformers.models.clip.modeling_clip import CLIPEncoderLayer
    from transformers.models.siglip.modeling_siglip import SiglipEncoderLayer
    from bimodalattack_amd.hf_adapter import _clip_layer_ok, _forward_shape, _CLIP_LAYER_SHAPE
    kw = dict(hidden_size=32, intermediate_size=64, num_attention_heads=2, num_hidden_layers=1)
    assert _clip_layer_ok(CLIPEncoderLayer(CLIPVisionConfig(**kw))) and _clip_layer_ok(SiglipEncoderLayer(SiglipVisionConfig(**kw)))
    good = """
def forward(self, hidden_states: "T", attention_mask: "T", **kwargs) -> "T":
    \"\"\"a docstring and annotations do not matter\"\"\"
    residual = hidden_states

    hidden_states = self.layer_norm1(hidden_states)
    hidden_states, _ = self.self_attn(
        hidden_states=hidden_states,
        attention_mask=attention_mask,
        **kwargs,
    )
    hidden_states = residual + hidden_states
    residual = hidden_states
    hidden_states = self.layer_norm2(hidden_states)
    hidden_states = self.mlp(hidden_states)
    hidden_states = residual + hidden_states
    return hidden_states
"""
    assert _forward_shape(good) == _CLIP_LAYER_SHAPE
    for old, new_ in (("attention_mask=attention_mask,", "attention_mask=attention_mask, causal_attention_mask=None,"),   # one more keyword
                      ("**kwargs) ->", "output_attentions=False, **kwargs) ->"),                                            # another signature
                      ("    return hidden_states", "    hidden_states = hidden_states * 1.0\n    return hidden_states"),  # a statement more
                      ("self.layer_norm2(hidden_states)", "self.layer_norm2(residual)")):                                   # another operand
        bad = good.replace(old, new_)
        assert bad != good and _forward_shape(bad) != _CLIP_LAYER_SHAPE, old
    assert _forward_shape("x = (") is None and _forward_shape("a = 1\nb = 2") is None


def test_fused_structure_detection_steps_aside_on_4_50_style_layers():
    """transformers 4.50.2's decoder layers return a TUPLE and unpack ``hidden_states, self_attn_weights = self.self_attn(``;
    the fused layer forward restates 5.x's statements and returns a tensor, so on such a layer it must NOT be installed
    (``_layer_kind`` reads the source) -- while the per-module fusions, which depend on attribute names only
    (``*RMSNorm`` with ``variance_epsilon``, ``gate_proj``/``up_proj``/``down_proj``/``act_fn``, ``q_proj``/``k_proj``/``v_proj``/
    ``o_proj`` with ``layer_idx``), are still admitted.  And without ``transformers.masking_utils`` (added after 4.50) the
    attention-interface routes report themselves unavailable instead of failing."""
    from bimodalattack_amd import fused

    class StubRMSNorm(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.weight = torch.nn.Parameter(torch.ones(d))
            self.variance_epsilon = 1e-6

        def forward(self, x):
            return self.weight * x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.variance_epsilon)

    class StubAttention(torch.nn.Module):
        def __init__(self, d, idx):
            super().__init__()
            self.layer_idx, self.head_dim = idx, d // 2
            self.q_proj, self.k_proj, self.v_proj, self.o_proj = (torch.nn.Linear(d, d, bias=False) for _ in range(4))

        def forward(self, hidden_states, **kw):
            return self.o_proj(self.q_proj(hidden_states) + self.k_proj(hidden_states) + self.v_proj(hidden_states)), None

    class StubMLP(torch.nn.Module):
        def __init__(self, d):
            super().__init__()
            self.gate_proj, self.up_proj, self.down_proj = (torch.nn.Linear(d, 2 * d, bias=False), torch.nn.Linear(d, 2 * d, bias=False),
                                                            torch.nn.Linear(2 * d, d, bias=False))
            self.act_fn = torch.nn.SiLU()

        def forward(self, x):
            return self.down_proj(self.act_fn(self.gate_proj(x)) * self.up_proj(x))

    class DecoderLayer450(torch.nn.Module):
        def __init__(self, d, idx):
            super().__init__()
            self.self_attn, self.mlp = StubAttention(d, idx), StubMLP(d)
            self.input_layernorm, self.post_attention_layernorm = StubRMSNorm(d), StubRMSNorm(d)

        def forward(self, hidden_states, output_attentions=False, **kwargs):          # 4.50.2's statements
            residual = hidden_states
            hidden_states = self.input_layernorm(hidden_states)
            hidden_states, self_attn_weights = self.self_attn(hidden_states=hidden_states, **kwargs)
            hidden_states = residual + hidden_states
            residual = hidden_states
            hidden_states = self.post_attention_layernorm(hidden_states)
            hidden_states = self.mlp(hidden_states)
            hidden_states = residual + hidden_states
            outputs = (hidden_states,)
            if output_attentions:
                outputs += (self_attn_weights,)
            return outputs

    class Stack(torch.nn.Module):
        def __init__(self, d=8, n=2):
            super().__init__()
            self.layers = torch.nn.ModuleList(DecoderLayer450(d, i) for i in range(n))
            self.norm = StubRMSNorm(d)

        def forward(self, x):
            for l in self.layers:
                x = l(x)[0]
            return self.norm(x)

    model = Stack()
    assert fused._layer_kind(model.layers[0]) is None
    f = fused.FusedInference(model)
    assert f.layers == [] and f.admitted["add_norm_layers"] == 0 and f.admitted["layer_kinds"] == []
    assert f.admitted["rmsnorms"] == 5 and f.admitted["gated_mlps"] == 2 and f.admitted["transposed_copy_projections"] == 14
    assert f.admitted["rotary_files"] == [] and f.admitted["qk_norm_in_rotary_blocks"] == 0
    # ... and the engine SAYS what it did not admit and why (logged once at construction)
    assert "statement for statement" in f.refused["add_norm_layers"] and "DecoderLayer450" in f.refused["add_norm_layers"]
    x = torch.randn(1, 5, 8)
    want = model(x)
    with f:                                   # on the CPU the kernels step aside; the 4.50-style tuple flows through untouched
        got = model(x)
    assert torch.equal(got, want) and not any("forward" in m.__dict__ for m in model.modules())

    # transformers without masking_utils / AttentionInterface: register() says no, nothing raises
    import builtins
    import sys
    from bimodalattack_amd import prefix_attention as pa
    real_import = builtins.__import__

    def no_masking_utils(name, *a, **kw):
        if name == "transformers.masking_utils":
            raise ImportError("No module named 'transformers.masking_utils'")
        return real_import(name, *a, **kw)

    saved, was = sys.modules.pop("transformers.masking_utils", None), dict(pa._REGISTERED)
    pa._REGISTERED["done"] = False
    builtins.__import__ = no_masking_utils
    try:
        assert pa.register() is False and pa._REGISTERED["done"] is False
    finally:
        builtins.__import__ = real_import
        if saved is not None:
            sys.modules["transformers.masking_utils"] = saved
        pa._REGISTERED.update(was)


def test_derived_weight_copies_are_versioned():
    from bimodalattack_amd.fused import _CopyCache
    c = _CopyCache()
    w = torch.randn(4, 3)
    wt = c.put(("wt", 1), w.t().contiguous(), (w,))
    assert c.get(("wt", 1), (w,)) is wt and c.get(("wt", 2), (w,)) is None and len(c) == 1
    w.add_(1.0)                                           # an in-place change bumps the version counter
    assert c.get(("wt", 1), (w,)) is None and len(c) == 0
    wt = c.put(("wt", 1), w.t().contiguous(), (w,))
    assert c.get(("wt", 1), (w.clone(),)) is None         # another tensor (another address) is another source


def test_gemm_mid_plan_and_routing_rule():
    """Planning of bma_gemm_mid through the C ABI (no launch): 224-row tiles, the tile width and the K split -- of every
    tile, or of the last columns of tiles only -- chosen so that the grid fits the 256 CUs; the workspace covers the split
    tiles' partials; ops.gemm_mid_ok routes the products the kernel measures faster on at 599-644 rows."""
    import ctypes
    from bimodalattack_amd import ops
    from bimodalattack_amd.native import lib
    plan = (ctypes.c_int * 8)()
    want = {(644, 22016, 4096): (4, 86, 8, 279, 255),      # 85 columns of tiles whole + the 86th split 8 ways in their shadow
            (644, 4096, 22016): (4, 16, 5, 240, 0), (644, 4096, 12288): (4, 16, 5, 240, 0), (599, 4096, 11008): (4, 16, 5, 240, 0),
            (643, 12288, 4096): (3, 64, 1, 192, 192), (644, 11008, 4096): (3, 58, 1, 174, 174), (644, 4096, 4096): (4, 16, 5, 240, 0),
            (450, 5120, 13824): (4, 20, 4, 240, 0), (200, 4096, 4096): (4, 16, 8, 128, 0)}
    for (M, N, K), (nf, n_tiles, S, wgs, unsplit) in want.items():
        assert lib.bma_gemm_mid_plan(M, N, K, plan) == 0
        got = list(plan)
        m_tiles = -(-M // 224)
        assert got == [7, m_tiles, nf, n_tiles, S, 1, wgs, unsplit], ((M, N, K), got)
        assert n_tiles == -(-N // (64 * nf)) and wgs == unsplit + (m_tiles * n_tiles - unsplit) * S
        assert lib.bma_gemm_mid_ws_bytes(M, N, K) == (0 if S == 1 else (m_tiles * n_tiles - unsplit) * S * 224 * 64 * nf * 4)
        assert lib.bma_gemm_mid_ws_bytes(M, N, K) <= ops._GEMM_WS_BYTES
        assert unsplit <= 256 and (wgs <= 256 or unsplit > 0)         # one round, or short pieces behind a round of whole tiles
    # the tuning override pins tile width / splits / split columns for every later call, and lets go again
    lib.bma_gemm_mid_set_plan(3, 2, 0, 0)
    assert lib.bma_gemm_mid_plan(644, 12288, 4096, plan) == 0 and list(plan)[2:8] == [3, 64, 2, 0, 384, 0]
    lib.bma_gemm_mid_set_plan(4, 4, 2, -1)
    assert lib.bma_gemm_mid_plan(644, 22016, 4096, plan) == 0 and list(plan)[2:8] == [4, 86, 4, 1, 252 + 6 * 4, 252]
    lib.bma_gemm_mid_set_plan(5, 0, -1, -1)
    assert lib.bma_gemm_mid_plan(644, 4096, 4096, plan) == -1                              # no such tile
    lib.bma_gemm_mid_set_plan(0, 0, -1, -1)
    assert lib.bma_gemm_mid_plan(644, 12288, 4096, plan) == 0 and list(plan)[2:5] == [3, 64, 1]
    assert lib.bma_gemm_mid_plan(644, 4096, 96, plan) == -1 and lib.bma_gemm_mid_ws_bytes(644, 4096, 96) == 0
    assert lib.bma_gemm_mid(16, 4096, 16, 4096, 16, 4096, 8, 4096, 96, 1, None, 0, None) == -5     # K % 64
    assert lib.bma_gemm_mid(16, 4096, 16, 4096, 16, 4096, 8, 4096, 4096, 0, None, 0, None) == -2   # fp32
    assert lib.bma_gemm_mid(None, 4096, 16, 4096, 16, 4096, 0, 4096, 4096, 1, None, 0, None) == 0  # no rows
    assert lib.bma_gemm_mid(16, 4096, 16, 4096, 16, 4096, 644, 4096, 11008, 1, None, 0, None) == -1   # split plan without a workspace
    assert (ops.GEMM_MID_MIN_ROWS, ops.GEMM_MID_MAX_ROWS, ops.GEMM_MID_MIN_K_OVER_N, ops.GEMM_MID_MIN_N_OVER_K) == (560, 672, 2.5, 4.0)
    assert not ops.gemm_mid_ok(torch.zeros(644, 11008, dtype=torch.bfloat16), torch.zeros(4096, 11008, dtype=torch.bfloat16))   # (CPU tensors never qualify)
    rule = lambda N, K: K >= ops.GEMM_MID_MIN_K_OVER_N * N or N >= ops.GEMM_MID_MIN_N_OVER_K * K      # noqa: E731
    assert [rule(N, K) for N, K in ((4096, 22016), (4096, 12288), (4096, 11008), (22016, 4096), (12288, 4096), (11008, 4096),
                                    (4096, 4096))] == [True, True, True, True, False, False, False]


def test_gemm_nt_plan_and_routing_rule():
    """Planning of bma_gemm_nt through the C ABI (no launch): slab height and split count are chosen together so that
    the workgroups of a product fill the CUs in whole rounds where K is split, the workspace covers every partial tile,
    and ops.gemm_nt_ok routes the products the kernel measures faster on."""
    import ctypes
    from bimodalattack_amd import ops
    from bimodalattack_amd.native import lib
    plan = (ctypes.c_int * 8)()
    for M, N, K in ((65, 4096, 22016), (65, 4096, 12288), (44, 4096, 22016), (65, 22016, 4096), (96, 4096, 4096), (65, 12288, 4096),
                    (65, 11008, 4096), (65, 4096, 11008), (19, 32064, 4096), (65, 20480, 2560)):
        assert lib.bma_gemm_nt_plan(M, N, K, plan) == 0
        mt, m_tiles, ntw, R, slabs, S, xcd, ntl = list(plan)
        assert mt == (4 if M <= 64 else 6) and m_tiles == 1 and ntw in (2, 3) and 16 <= R <= 64 * ntw and R % 4 == 0
        assert slabs == -(-N // R) and 1 <= S <= 16 and ntl == 1
        assert lib.bma_gemm_nt_tiles(M, N, K) == slabs
        assert lib.bma_gemm_nt_ws_bytes(M, N, K) == (0 if S == 1 else slabs * S * 16 * mt * 64 * ntw * 4)
        assert lib.bma_gemm_nt_ws_bytes(M, N, K) <= ops._GEMM_WS_BYTES and slabs <= ops._GEMM_COUNTERS
        assert slabs * S <= 256                                   # one round of the 256 CUs
        if S > 1:
            assert xcd == ((slabs * S) % 8 == 0)                  # a tile's splits on one XCD whenever the grid allows
        if N == 4096 and K >= 11008:
            assert (R, S) == (128, 8)                             # 32 slabs x 8 splits: one full round
        if (N, K) == (22016, 4096):
            assert (R, S) == (128, 1)                             # one pass, 172 CUs: splitting K cost more than it filled
    # the tuning override pins a decomposition for every later call, and lets go again
    lib.bma_gemm_nt_set_plan(3, 172, 2, 0)
    assert lib.bma_gemm_nt_plan(65, 22016, 4096, plan) == 0 and list(plan)[2:8] == [3, 172, 128, 2, 0, 0]
    lib.bma_gemm_nt_set_plan(0, 0, 0, -1)
    assert lib.bma_gemm_nt_plan(65, 22016, 4096, plan) == 0 and list(plan)[2:6] == [2, 128, 172, 1]
    assert lib.bma_gemm_nt_plan(65, 4096, 100, plan) == -1
    assert lib.bma_gemm_nt_ws_bytes(65, 4096, 100) == 0 and lib.bma_gemm_nt_tiles(0, 4096, 4096) == 0
    assert lib.bma_gemm_nt(16, 4096, 16, 4096, 16, 4096, 8, 4096, 100, 1, None, 0, None, 0, None) == -5     # K % 64
    assert lib.bma_gemm_nt(16, 4096, 16, 4096, 16, 4096, 8, 4096, 4096, 0, None, 0, None, 0, None) == -2    # fp32
    assert lib.bma_gemm_nt(None, 4096, 16, 4096, 16, 4096, 0, 4096, 4096, 1, None, 0, None, 0, None) == 0   # no rows
    assert ops.GEMM_NT_MIN_K_OVER_N == 2.5 and ops.GEMM_NT_MIN_N_OVER_K == 2.5 and ops.GEMM_NT_MAX_ROWS == 96
    x = torch.zeros(65, 4096, dtype=torch.bfloat16)
    assert not ops.gemm_nt_ok(x, torch.zeros(4096, 4096, dtype=torch.bfloat16))          # (CPU tensors never qualify)
    # the rule itself, on shapes alone: every product of the pass but the square o_proj
    rule = lambda N, K: K >= ops.GEMM_NT_MIN_K_OVER_N * N or N >= ops.GEMM_NT_MIN_N_OVER_K * K      # noqa: E731
    assert [rule(N, K) for N, K in ((4096, 22016), (4096, 12288), (4096, 11008), (22016, 4096), (12288, 4096), (11008, 4096),
                                    (4096, 4096))] == [True, True, True, True, True, True, False]


def test_bench_gemm_roles_and_extra_workloads():
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    from types import SimpleNamespace as NS
    tc = NS(hidden_size=4096, intermediate_size=11008, num_attention_heads=32, num_key_value_heads=32)
    r = bench.gemm_roles(tc, 32064)
    assert r[(22016, 4096)] == "gate_up_proj (fused)" and r[(4096, 22016)] == "gate_up_proj (fused) dX"
    assert r[(12288, 4096)] == "qkv_proj (fused)" and r[(4096, 11008)] == "down_proj" and r[(11008, 4096)] == "down_proj dX"
    assert r[(4096, 4096)] == "o_proj" and r[(32064, 4096)] == "lm_head / token scores"
    g = NS(hidden_size=2560, intermediate_size=10240, num_attention_heads=8, num_key_value_heads=4, head_dim=256)
    rg = bench.gemm_roles(g, 262208)
    assert rg[(20480, 2560)] == "gate_up_proj (fused)" and rg[(2560, 2048)] == "o_proj" and rg[(1024, 2560)] == "k_proj / v_proj"
    assert set(bench.WORKLOADS) >= {"gcg", "joint", "pgd", "pgd_gcg", "gemma_joint", "opt125m"}
    assert bench.COPY_CEILING_GBS < bench.HBM_PEAK_GBS and bench.MFMA_PEAK_TFLOPS == 2500.0


def test_qk_norm_is_deferred_only_where_the_rotary_function_is_patched(monkeypatch):
    """a head norm may be deferred into the rotary launch only in blocks whose modelling file's
    apply_rotary_pos_emb the fused context patches -- elsewhere nobody picks the un-normalised tensor up.  The
    admission is recorded (`admitted`), and a missed pick-up switches the deferral off instead of failing every call."""
    from bimodalattack_amd import fused, synthetic as S
    model, _, _, _ = S.tiny_case("gemma3")
    f = fused.FusedInference(model)
    assert f.admitted["qk_norm_in_rotary_blocks"] == 2 and f.admitted["qk_norm_blocks_not_admitted"] == 0
    assert f.admitted["rotary_files"] == ["modeling_gemma3"] and f.admitted["layer_kinds"] == ["gemma"]
    with pytest.raises(fused.DeferredNormMissed):
        f._pending[1] = (None, None, 0.0, True)
        f._missed()
    assert not f._rope_norms and not f._pending and f.admitted["qk_norm_in_rotary_blocks"] == 0
    monkeypatch.setattr(fused, "_ROPE_FILES", ("modeling_llama",))
    g = fused.FusedInference(model)
    assert not g._rope_norms and g.admitted["qk_norm_blocks_not_admitted"] == 2 and g.admitted["rotary_files"] == []
    assert "2 blocks" in g.refused["qk_norm_in_rotary_blocks"] and "qk_norm_in_rotary_blocks" not in f.refused


def test_bench_line_fits_the_drivers_window_and_is_strict_json():
    """The driver keeps the last 8000 characters of bench.py's stdout and parses the last line (round 3's 44 KB line
    did not parse).  build_line() turns a full result -- here round 3's own, NaN loss of the Gemma workload included,
    then the same blown up with prose and workloads -- into one line under LINE_LIMIT that strict JSON parsers take,
    carrying the contract's fields, `roofline`, `cpu_baseline` and a `finite` verdict per workload."""
    import json
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, repo)
    import bench
    full = json.load(open(os.path.join(repo, "profiles", "archive", "r3_bench_driver.json")))
    full["finite"] = True
    full["workloads"]["joint"]["finite"] = True
    full["workloads"]["gemma_joint"]["final_loss"] = float("nan")
    full["workloads"]["gemma_joint"]["finite"] = False
    full["rccl"] = dict(backend="nccl", world=8, rccl_version="2.26.6", device_names=["rank %d: AMD Instinct MI355X" % i for i in range(8)],
                        collectives_per_step=2.0, allgather_bytes_per_step=2048, allgather_bytes_per_rank=256,
                        state_broadcast_bytes_per_step=77824, what="x" * 500,
                        # the multi-GPU A/B of the tensor-parallel gradient pass (bench.tp_ab)
                        tp_off_ms=37.91, tp_on_ms=31.25, chosen="on", tp_graph=True, tp_note="n" * 400,
                        tp_fallbacks={"graph_gradient_tp": "RuntimeError: " + "e" * 300})
    # north_star's table for this N, as bench.scaling_table builds it on 8 ranks (round 6)
    full["scaling_table"] = dict(n_gpus=8, attack_steps_per_sec=26.7, candidate_forwards_per_sec=13615.0, ms_per_step=37.46,
                                 own_n1_leg=dict(ms_per_step=180.7, candidate_forwards_per_sec=2826.0, attack_steps_per_sec=5.53, steps=5, finite=True),
                                 own_n1_leg_candidate_forwards_per_sec_per_rank=[2826.0 + i for i in range(8)], efficiency_vs_own_n1=0.6022,
                                 dominant_kernel=dict(kernel="hipBLASLt/rocBLAS GEMM gate_up_proj M=2176 N=22016 K=4096 (decoder gate_up_proj, bf16)",
                                                      bound="mfma", unit="TFLOP/s", peak=2500.0, frac_per_rank=[0.5312 + 0.001 * i for i in range(8)]),
                                 gradient_pass="replicated on every rank")
    for blow in (0, 1):
        if blow:
            full["roofline"]["note"] = "prose " * 2000
            full["roofline"]["kernel"] = "K" * 900
            full["cpu_baseline"]["sample"] = "s" * 5000
            for i in range(40):
                full["workloads"][f"extra{i}"] = dict(full["workloads"]["joint"])
        line = bench.build_line(full, "gpurun_out/bench_detail.json")
        text = json.dumps(line, allow_nan=False)                 # raises on NaN / Infinity
        assert len(text) <= bench.LINE_LIMIT < 8000, len(text)
        back = json.loads(text, parse_constant=lambda c: pytest.fail(f"non-JSON constant {c}"))
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                  "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "finite", "detail_file"):
            assert k in back, k
        assert back["finite"] is True and back["value"] == pytest.approx(full["value"], rel=1e-4)
        assert set(back["roofline"]) >= {"bound", "kernel", "achieved", "peak", "unit", "frac", "traffic"}
        assert back["roofline"]["frac"] == pytest.approx(full["roofline"]["frac"], rel=1e-4)
        assert set(back["cpu_baseline"]) >= {"value", "unit", "cores", "kind", "sample"}
        assert "model" not in back["config"] and "workload" in back["config"]
        if not blow:
            g = back["workloads"]["gemma_joint"]
            assert g["finite"] is False and g["final_loss"] is None and back["workloads"]["joint"]["finite"] is True
            assert back["rccl"]["world"] == 8 and back["rccl"]["devices"] == 8 and "what" not in back["rccl"]
            assert (back["rccl"]["tp_off_ms"], back["rccl"]["tp_on_ms"], back["rccl"]["chosen"]) == (37.91, 31.25, "on")
            assert back["rccl"]["tp_graph"] is True and len(back["rccl"]["tp_note"]) <= 160
            t = back["scaling_table"]
            assert t["n_gpus"] == 8 and len(t["dominant_kernel"]["frac_per_rank"]) == 8 and t["efficiency_vs_own_n1"] == pytest.approx(0.6022, rel=1e-3)
            assert t["own_n1_leg"]["candidate_forwards_per_sec"] == pytest.approx(2826.0)
    assert bench._strict({"a": [float("inf"), 1.0, {"b": float("nan")}]}) == {"a": [None, 1.0, {"b": None}]}


def test_virtual_ids_plan_what_the_real_candidates_need():
    """early_plan (attack.py `_virtual_ids`): the host plans the ragged forward from the random draws alone, on
    stand-ins for ids it has not seen.  For every draw: equal stand-ins mean equal candidates (so the dedup never
    merges two different ones), a stand-in's first difference from the parent is no later than the candidate's (so
    every row a candidate needs is computed), and `unique_rows(return_first)` names a row of the batch that IS the
    distinct candidate."""
    from bimodalattack_amd.attack import BimodalAttack
    from bimodalattack_amd.layout import ragged_plan, unique_rows

    class Done:
        def synchronize(self):
            pass

    rng = np.random.RandomState(7)
    for n_opt, n_replace, topk, width in [(20, 1, 8, 512), (12, 2, 4, 256), (5, 1, 2, 64), (9, 3, 3, 128)]:
        parent = rng.randint(0, 50, size=n_opt)
        table = np.stack([rng.permutation(50)[:topk] for _ in range(n_opt)])         # top-k ids per position: distinct
        table[:, 0] = parent                          # rank 0 proposes the token already there: candidate == parent
        pos = np.stack([rng.permutation(n_opt)[:n_replace] for _ in range(width)])
        rank = rng.randint(0, topk, size=(width, n_replace))
        real = np.repeat(parent[None], width, 0)
        np.put_along_axis(real, pos, table[pos, rank], axis=1)
        early = dict(event=Done(), host=torch.from_numpy(np.stack([pos, rank])))
        fake, par = BimodalAttack._virtual_ids(early, parent.tolist())
        assert (par == parent).all() and fake.shape == real.shape
        uniq, inv, first = unique_rows(fake, return_first=True)
        assert (real[first][inv] == real).all()                      # merged stand-ins <=> identical candidates
        d_fake, d_real = fake != parent, real != parent
        p_fake = np.where(d_fake.any(1), d_fake.argmax(1), n_opt - 1)
        p_real = np.where(d_real.any(1), d_real.argmax(1), n_opt - 1)
        assert (p_fake <= p_real).all() and (p_fake == pos.min(1)).all()
        plan = ragged_plan(uniq, par, L=n_opt + 6, T=3, P=11, dedup=False, padded_maps=False, inverse=inv)
        assert plan is not None and (plan["cand"] == uniq).all() and (plan["p"] <= p_real[first]).all()


def test_kernels_refuse_any_architecture_but_gfx950():
    """The library is gfx950-only by construction (and `bma_gemm_nt`'s unfenced split-K hand-off by validation): the first
    launch on a device looks at its architecture name and refuses anything else."""
    from bimodalattack_amd import ops
    ops.check_arch("gfx950:sramecc+:xnack-")
    ops.check_arch("gfx950")
    for other in ("gfx942:sramecc+:xnack-", "gfx90a", "gfx1100", "unknown", ""):
        with pytest.raises(RuntimeError, match="gfx950"):
            ops.check_arch(other)


def test_bench_line_guard_prints_the_held_line_only_when_its_parent_dies():
    """bench.py on several GPUs: a child of rank 0 holds the first leg's line while the tensor-parallel leg runs.  Parent
    killed (as by a faulting collective, or the launcher's signal): the held line comes out, once.  Parent says DONE (it prints
    its own line): nothing.  Nothing ever held (the first leg itself failed): nothing."""
    import subprocess
    import sys
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, os, signal\n"
        f"sys.path.insert(0, {repo!r})\n"
        "import bench\n"
        "g = bench._LineGuard()\n"
        "mode = sys.argv[1]\n"
        "if mode != 'none':\n"
        "    g.hold('{\"held\": 1}')\n"
        "if mode == 'release':\n"
        "    g.release(); print('{\"own\": 1}', flush=True)\n"
        "elif mode == 'die':\n"
        "    os.kill(os.getpid(), signal.SIGKILL)\n")
    got = {}
    for mode in ("die", "release", "none"):
        r = subprocess.run([sys.executable, "-c", code, mode], capture_output=True, text=True, timeout=120)
        got[mode] = (r.returncode, r.stdout)
    assert got["die"] == (-9, '{"held": 1}\n')
    assert got["release"] == (0, '{"own": 1}\n')
    assert got["none"] == (0, "")


def test_round_cut_splits_only_just_above_whole_tile_rounds():
    """fused.round_cut: a rows x N product is cut into (whole 256 x 256 tile rounds, rest) at the last row-tile boundary on
    which a round ends, when the rest is at most 5/16 of a round's tiles."""
    from bimodalattack_amd.fused import round_cut
    assert round_cut(16896, 4096, 256) == 16384 and round_cut(17152, 4096, 256) == 16384 and round_cut(17664, 4096, 256) == 16384
    assert round_cut(16384, 4096, 256) == 0              # whole rounds already
    assert round_cut(18432, 4096, 256) == 0              # half a round left: one call
    assert round_cut(17920, 4096, 256) == 0              # 6 row tiles x 16 = 96 tiles > 80
    assert round_cut(3000, 4096, 256) == 0               # less than one round
    assert round_cut(8704, 4096, 256) == 8192 and round_cut(4352, 4096, 256) == 4096      # rank 0's rows of 2 / 4 ranks
    assert round_cut(16896, 12288, 256) == 0             # 48 tile columns: 2 row tiles behind the boundary are 96 tiles
    assert round_cut(16640, 12288, 256) == 16384 and round_cut(4352, 12288, 256) == 4096  # one row tile behind it: 48 tiles
    assert round_cut(16896, 22016, 256) == 0             # 86 tile columns: rounds end every 128 row tiles
    assert round_cut(48480, 2560, 256) == 0 and round_cut(48480, 20480, 256) == 0         # Gemma-3's products: too much left
    assert round_cut(16896, 4096, 304) == 0 and round_cut(4200, 8192, 256) == 4096
    assert round_cut(16500, 4096, 256) == 16384          # a row count off the tile grid: the cut is on it


def test_col_cut_takes_the_columns_that_fill_whole_rounds():
    """fused.col_cut: the first call gets the most tile columns that still fill whole rounds when the product has at most six
    of them and at most 5/16 of a round's tiles are left; never at a few hundred rows."""
    from bimodalattack_amd.fused import col_cut, round_cut
    for rows in (2112, 2176, 2240, 2304):                # rank 0's gate/up product of eight GPUs: 9 x 86 tiles = 3.02 rounds
        assert col_cut(rows, 22016, 256) == 85 * 256 and round_cut(rows, 22016, 256) == 0
    assert col_cut(2048, 22016, 256) == 0                # 8 x 86 = 688 tiles: 176 behind 64 columns
    assert col_cut(2432, 22016, 256) == 0                # 10 x 86: 100 tiles behind 76 columns (measured: 340 -> 493 us)
    assert col_cut(16896, 22016, 256) == 0               # 22 whole rounds: the saved one is not worth a 256-column product
    assert col_cut(599, 22016, 256) == 0 and col_cut(644, 22016, 256) == 0      # the batch-1 passes
    assert col_cut(4352, 12288, 256) == 45 * 256         # (the row cut comes first in two_calls: 4096 rows)
    assert col_cut(4096, 4096, 256) == 0                 # whole rounds exactly

