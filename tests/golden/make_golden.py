#!/usr/bin/env python3
"""Capture golden vectors G1-G5 from the REAL reference.

Runs only in the build container, where /root/reference is mounted read-only.
It imports the reference package (never copies it), drives its functions on
small synthetic inputs and stores NUMBERS ONLY -- inputs and expected outputs --
as .npz files next to this script.  The GPU box never sees the reference; tests
there compare the oracle and the HIP path with these files.

    python tests/golden/make_golden.py            # regenerate everything

Adapter notes (what it takes to run the 2025-06 reference on this image):
  * torchvision is not installed and the reference imports it without using it
    -> empty stub modules are registered before the import;
  * transformers 5.x returns an output object from ``get_image_features`` where
    the pinned 4.50.2 returned a tensor -> the MODEL INSTANCE's method is wrapped
    to return the (1, N_img, D) tensor the reference expects.
"""

from __future__ import annotations

import importlib.machinery
import json
import os
import sys
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("BMA_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)
sys.dont_write_bytecode = True

import transformers  # noqa: E402  (must be imported before the stubs)

for _name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
    if _name not in sys.modules:
        _m = types.ModuleType(_name)
        _m.__spec__ = importlib.machinery.ModuleSpec(_name, None)
        _m.__path__ = []
        sys.modules[_name] = _m
sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]

sys.path.insert(0, REF)
import bimodalattack as ref  # noqa: E402
from bimodalattack import bimodal_attack as refmod  # noqa: E402
from bimodalattack import utils as refutils  # noqa: E402

from bimodalattack_amd import synthetic as S  # noqa: E402

refmod.logger.setLevel("ERROR")


def save(name: str, **arrays) -> None:
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {name}: " + ", ".join(f"{k}{tuple(np.shape(v))}" for k, v in arrays.items()))


# --------------------------------------------------------------------------
# G1 -- sample_ids_from_grad (reference :130-163)
# --------------------------------------------------------------------------
def g1_sampling() -> None:
    out = {}
    cases = [
        # name, n_opt, V, sw, topk, n_replace, n_not_allowed
        ("a", 19, 2048, 64, 32, 1, 40),
        ("b", 19, 2048, 48, 256, 2, 0),
        ("c", 7, 515, 33, 16, 3, 17),
        ("d", 19, 2048, 16, 256, 1, 300),
    ]
    for name, n_opt, V, sw, topk, n_rep, n_na in cases:
        rs = np.random.RandomState(100 + ord(name))
        # tie-free fp32 gradient: a permutation of distinct values per row
        grad = np.stack([rs.permutation(V).astype(np.float32) / V - 0.5 + r * 1e-3 for r in range(n_opt)])
        grad = (grad * rs.uniform(0.5, 2.0)).astype(np.float32)
        ids = rs.randint(5, V, size=n_opt).astype(np.int64)
        na = np.sort(rs.choice(V, size=n_na, replace=False)).astype(np.int64) if n_na else None
        seed = 1234 + ord(name)

        # the randoms the reference is about to draw (same generator, same order)
        torch.manual_seed(seed)
        rnd = torch.rand((sw, n_opt))
        rank = torch.randint(0, topk, (sw, n_rep, 1))

        g = torch.from_numpy(grad.copy())
        torch.manual_seed(seed)
        new_ids = refmod.sample_ids_from_grad(
            torch.from_numpy(ids.copy()), g, sw, topk, n_rep,
            not_allowed_ids=None if na is None else torch.from_numpy(na),
        )
        # g was masked in place by the reference (:145); top-k of it is the third-party step
        topk_ids = (-g).topk(topk, dim=1).indices
        pos = torch.argsort(rnd)[..., :n_rep]
        out.update({
            f"{name}_ids": ids, f"{name}_grad": grad,
            f"{name}_not_allowed": np.zeros(0, np.int64) if na is None else na,
            f"{name}_rand": rnd.numpy(), f"{name}_rank": rank.squeeze(2).numpy(),
            f"{name}_pos": pos.numpy(), f"{name}_grad_masked": g.numpy(),
            f"{name}_topk_ids": topk_ids.numpy(), f"{name}_new_ids": new_ids.numpy(),
            f"{name}_meta": np.array([n_opt, V, sw, topk, n_rep], np.int64),
        })
    save("g1_sampling.npz", **out)


# --------------------------------------------------------------------------
# G2 -- candidate CE (reference :1278-1310) and gradient CE (:953-1028)
# --------------------------------------------------------------------------
class _LinearLM:
    """A transparent 'model': logits = inputs_embeds @ W.  Lets the golden pin
    the reference's slicing / CE / autograd plumbing without a transformer."""

    def __init__(self, W: torch.Tensor):
        self.W = W
        self.device = torch.device("cpu")
        self.dtype = W.dtype

    def __call__(self, inputs_embeds=None, **kw):
        return SimpleNamespace(logits=inputs_embeds @ self.W)


def g2_ce() -> None:
    rs = np.random.RandomState(7)
    B, SEQ, D, V, T = 8, 24, 16, 512, 5
    W = torch.from_numpy((rs.standard_normal((D, V)) * 0.7).astype(np.float32))
    embeds = torch.from_numpy(rs.standard_normal((B, SEQ, D)).astype(np.float32))
    target = torch.from_numpy(rs.randint(0, V, size=(1, T)).astype(np.int64))
    # make candidate 5 a perfect argmax match so early_stop fires (:1300-1306)
    logits = embeds @ W
    shift = SEQ - T
    hit = logits[5, shift - 1:-1].argmax(-1)
    target_hit = hit.view(1, T).clone()

    def run(tgt, early):
        stub = SimpleNamespace(model=_LinearLM(W), target_ids=tgt,
                               config=SimpleNamespace(early_stop=early), stop_flag=False)
        loss = refmod.BimodalAttack._compute_candidates_loss_original(stub, 3, embeds)
        return loss, stub.stop_flag

    loss_a, flag_a = run(target, True)
    loss_b, flag_b = run(target_hit, True)
    assert not flag_a and flag_b

    # third-party arithmetic at the call sites (:1010, :1293): per-row CE, mean, dlogits
    sl = logits[:, shift - 1:-1].contiguous().clone().requires_grad_()
    rows = torch.nn.functional.cross_entropy(sl.view(-1, V), target.repeat(B, 1).view(-1), reduction="none")
    mean0 = torch.nn.functional.cross_entropy(sl[0], target.view(-1))
    (dlog0,) = torch.autograd.grad(mean0, sl)

    save("g2_ce.npz", embeds=embeds.numpy(), W=W.numpy(), logits=logits.numpy(), target=target.numpy(),
         target_hit=target_hit.numpy(), loss=loss_a.numpy(), loss_hit=loss_b.numpy(),
         stop_flag=np.array([flag_a, flag_b]), row_loss=rows.detach().view(B, T).numpy(),
         mean0=mean0.detach().numpy(), dlogits0=dlog0[0].numpy())

    # gradient plumbing (:953-1028) on the transparent model, text-only and image layouts
    out = {}
    Vg, Dg, n_opt = 64, 16, 6
    E = torch.nn.Embedding(Vg, Dg)
    with torch.no_grad():
        E.weight.copy_(torch.from_numpy(rs.standard_normal((Vg, Dg)).astype(np.float32)))
    E.weight.requires_grad_(False)
    Wg = torch.from_numpy((rs.standard_normal((Dg, Vg)) * 0.5).astype(np.float32))
    ids = torch.from_numpy(rs.randint(0, Vg, size=(1, n_opt)).astype(np.int64))
    seg = lambda n: torch.from_numpy(rs.standard_normal((1, n, Dg)).astype(np.float32))  # noqa: E731
    tgt = torch.from_numpy(rs.randint(0, Vg, size=(1, 4)).astype(np.int64))
    before, after = seg(3), seg(2)
    stub = SimpleNamespace(
        model=_LinearLM(Wg), embedding_layer=E, target_ids=tgt,
        config=SimpleNamespace(gcg_attack=True, pgd_attack=False),
        before_embeds=before, after_embeds=after, target_embeds=E(tgt),
    )
    g_text, none_img = refmod.BimodalAttack.compute_gradient(stub, ids)
    assert none_img is None
    out.update(E=E.weight.numpy(), W=Wg.numpy(), ids=ids.numpy(), target=tgt.numpy(),
               before=before.numpy(), after=after.numpy(), grad_text=g_text.numpy())

    # image layout: image_features = A * normalize(image) flattened, so d/d image is exact
    P = torch.from_numpy((rs.standard_normal((12, 2 * Dg)) * 0.3).astype(np.float32))

    class _VLM(_LinearLM):
        def get_image_features(self, pixel_values=None, **kw):
            return (pixel_values.reshape(1, -1) @ P).view(1, 2, Dg)

    image = torch.from_numpy(rs.uniform(0, 1, size=(1, 3, 2, 2)).astype(np.float32)).requires_grad_()
    norm = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)
    bi, bs = seg(2), seg(3)
    stub2 = SimpleNamespace(
        model=_VLM(Wg), embedding_layer=E, target_ids=tgt, normalize=norm,
        processor=SimpleNamespace(), config=SimpleNamespace(gcg_attack=True, pgd_attack=True),
        before_img_embeds=bi, before_suffix_embeds=bs, after_embeds=after, target_embeds=E(tgt),
    )
    g_tok, g_img = refmod.BimodalAttack.compute_gradient(stub2, ids, image)
    out.update(P=P.numpy(), image=image.detach().numpy(), before_img=bi.numpy(), before_suffix=bs.numpy(),
               grad_tok_img=g_tok.numpy(), grad_img=g_img.numpy())
    save("g2_grad.npz", **out)


# --------------------------------------------------------------------------
# G3 -- perform_pgd_step (reference :1030-1037)
# --------------------------------------------------------------------------
def g3_pgd() -> None:
    rs = np.random.RandomState(3)
    out = {}
    for name, eps, alpha, shape in [
        ("a", 64 / 255, 4 / 255, (1, 3, 28, 28)),
        ("b", 0.1, 0.01, (1, 3, 17, 19)),       # ragged size: not a multiple of 4
        ("c", 8 / 255, 1.0, (1, 3, 5, 7)),      # step = eps: lands on the ball's surface
    ]:
        x0 = rs.uniform(0, 1, size=shape).astype(np.float32)
        # start somewhere inside the ball, some pixels exactly on its surface / on 0 and 1
        x = np.clip(x0 + rs.uniform(-eps, eps, size=shape).astype(np.float32), 0, 1).astype(np.float32)
        flat = x.reshape(-1)
        flat[::11] = 0.0
        flat[5::13] = 1.0
        g = rs.standard_normal(shape).astype(np.float32)
        g.reshape(-1)[::7] = 0.0
        g.reshape(-1)[3::29] = -0.0
        if name == "b":
            g.reshape(-1)[1::31] = np.nan       # torch.sign(nan) == 0: the pixel only gets re-projected
            g.reshape(-1)[2::37] = np.inf
        y = refmod.BimodalAttack.perform_pgd_step(
            None, torch.from_numpy(x.copy()), eps, alpha, torch.from_numpy(g), torch.from_numpy(x0)
        )
        assert y.requires_grad
        out.update({f"{name}_x": x, f"{name}_g": g, f"{name}_x0": x0, f"{name}_y": y.detach().numpy(),
                    f"{name}_eps_alpha": np.array([eps, alpha], np.float64)})
    # five consecutive steps (what the PGD inner loop does to one image)
    eps, alpha = 64 / 255, 4 / 255
    x0 = rs.uniform(0, 1, size=(1, 3, 8, 8)).astype(np.float32)
    x = torch.from_numpy(x0.copy())
    gs, xs = [], []
    for _ in range(5):
        g = rs.standard_normal(x0.shape).astype(np.float32)
        x = refmod.BimodalAttack.perform_pgd_step(None, x, eps, alpha, torch.from_numpy(g), torch.from_numpy(x0))
        gs.append(g)
        xs.append(x.detach().numpy().copy())
    out.update(chain_x0=x0, chain_g=np.stack(gs), chain_x=np.stack(xs), chain_eps_alpha=np.array([eps, alpha]))
    save("g3_pgd.npz", **out)


# --------------------------------------------------------------------------
# G4 -- _build_input_embeds (reference :1112-1225)
# --------------------------------------------------------------------------
def g4_splice() -> None:
    from transformers.models.gemma3.modeling_gemma3 import Gemma3TextScaledWordEmbedding

    rs = np.random.RandomState(4)
    V, D, n_opt, B = 40, 8, 5, 6
    out = {}
    table = rs.standard_normal((V, D)).astype(np.float32)
    ids = rs.randint(0, V, size=(B, n_opt)).astype(np.int64)
    seg = {k: rs.standard_normal((1, n, D)).astype(np.float32)
           for k, n in [("before", 4), ("before_img", 2), ("before_suffix", 3), ("after", 2), ("target", 3)]}
    image = rs.standard_normal((1, 4, D)).astype(np.float32)
    out.update(table=table, ids=ids, image=image, **{f"seg_{k}": v for k, v in seg.items()})

    for mt in ("llava", "gemma3"):
        if mt == "gemma3":
            emb = Gemma3TextScaledWordEmbedding(V, D, padding_idx=0, embed_scale=D ** 0.5)
        else:
            emb = torch.nn.Embedding(V, D)
        with torch.no_grad():
            emb.weight.copy_(torch.from_numpy(table))
        stub = SimpleNamespace(
            model=SimpleNamespace(config=SimpleNamespace(model_type=mt)), embedding_layer=emb,
            before_embeds=torch.from_numpy(seg["before"]), before_img_embeds=torch.from_numpy(seg["before_img"]),
            before_suffix_embeds=torch.from_numpy(seg["before_suffix"]), after_embeds=torch.from_numpy(seg["after"]),
            target_embeds=torch.from_numpy(seg["target"]),
        )
        combos = [
            ("pgd_single", dict(mode="pgd", single=True), True),
            ("gcg_single", dict(mode="gcg", single=True), False),
            ("gcg_nojoint", dict(mode="gcg", no_joint_eval=True), False),
            ("gcg_notarget", dict(mode="gcg", no_target=True), False),
            ("gcgpgd_single", dict(mode="gcg_pgd", single=True), True),
            ("gcgpgd_notarget", dict(mode="gcg_pgd", no_target=True), True),
            ("gcgpgd_full", dict(mode="gcg_pgd"), True),
        ]
        with torch.no_grad():
            for cname, kw, with_img in combos:
                y = refmod.BimodalAttack._build_input_embeds(
                    stub, torch.from_numpy(ids), image=torch.from_numpy(image) if with_img else None,
                    search_width=B, **kw)
                out[f"{mt}_{cname}"] = y.numpy()
            # the winner re-score call shape (:605-607): one candidate, search_width=None
            y1 = refmod.BimodalAttack._build_input_embeds(
                stub, torch.from_numpy(ids[2:3]), image=torch.from_numpy(image), search_width=None, mode="gcg_pgd")
            out[f"{mt}_gcgpgd_one"] = y1.numpy()
    save("g4_splice.npz", **out)


# --------------------------------------------------------------------------
# G5 -- whole trajectories of run() on tiny random models (reference :251-824)
# --------------------------------------------------------------------------
def _adapt_image_features(model) -> None:
    orig = model.get_image_features

    def tensor_features(*a, **kw):
        o = orig(*a, **kw)
        if torch.is_tensor(o):
            return o
        p = o.pooler_output
        return p if torch.is_tensor(p) else torch.stack(list(p))

    model.get_image_features = tensor_features  # instance attribute only


class _Trace:
    """Wraps the reference's module functions / instance methods to record what
    each step saw.  Nothing about the algorithm is changed."""

    def __init__(self):
        self.steps = []
        self.cur = None

    def install(self):
        t = self
        self._orig = dict(sample=refmod.sample_ids_from_grad, filt=refmod.filter_ids,
                          grad=refmod.BimodalAttack.compute_gradient,
                          loss=refmod.BimodalAttack._compute_candidates_loss_original,
                          pgd=refmod.BimodalAttack.perform_pgd_step,
                          save=refmod.BimodalAttack._save_image, log=refmod.AttackBuffer.log_buffer)

        def sample(ids, grad, *a, **kw):
            t.cur["grad_tok"].append(grad.detach().clone().float().numpy())
            r = t._orig["sample"](ids, grad, *a, **kw)
            t.cur["sampled"] = r.clone().numpy()
            return r

        def filt(ids, tok):
            r = t._orig["filt"](ids, tok)
            t.cur["filtered"] = r.clone().numpy()
            return r

        def grad(self, optim_ids, image=None):
            if t.cur is None or t.cur.get("closed"):
                t.cur = dict(grad_tok=[], grad_img=[], losses=[], n_grad=0, closed=False)
                t.steps.append(t.cur)
            t.cur["n_grad"] += 1
            t.cur["optim_ids_in"] = optim_ids.clone().numpy()
            r = t._orig["grad"](self, optim_ids, image)
            if r[1] is not None:
                t.cur["grad_img"].append(r[1].detach().clone().numpy())
            return r

        def loss(self, bs, embeds):
            r = t._orig["loss"](self, bs, embeds)
            if t.cur is not None:
                t.cur["losses"].append(r.detach().clone().float().numpy())
            else:
                t.init_losses = r.detach().clone().float().numpy()
            return r

        def pgd(self, image, eps, alpha, g, x0):
            r = t._orig["pgd"](self, image, eps, alpha, g, x0)
            t.cur["image_after_pgd"] = r.detach().clone().numpy()
            return r

        def save_img(self, image, path):
            t._orig["save"](self, image, path)

        def log_buffer(self, tokenizer):  # last call of every step (:783); also once at init
            t._orig["log"](self, tokenizer)
            if t.cur is not None:
                t.cur["closed"] = True

        refmod.sample_ids_from_grad = sample
        refmod.filter_ids = filt
        refmod.BimodalAttack.compute_gradient = grad
        refmod.BimodalAttack._compute_candidates_loss_original = loss
        refmod.BimodalAttack.perform_pgd_step = pgd
        refmod.BimodalAttack._save_image = save_img
        refmod.AttackBuffer.log_buffer = log_buffer

    def uninstall(self):
        refmod.sample_ids_from_grad = self._orig["sample"]
        refmod.filter_ids = self._orig["filt"]
        refmod.BimodalAttack.compute_gradient = self._orig["grad"]
        refmod.BimodalAttack._compute_candidates_loss_original = self._orig["loss"]
        refmod.BimodalAttack.perform_pgd_step = self._orig["pgd"]
        refmod.BimodalAttack._save_image = self._orig["save"]
        refmod.AttackBuffer.log_buffer = self._orig["log"]


TRAJ = {
    # name: (model kind, config overrides, steps)
    "opt_gcg": ("opt", dict(num_steps=10, search_width=16, topk=32, pgd_attack=False, gcg_attack=True), None),
    "llava_pgd": ("llava", dict(num_steps=4, pgd_attack=True, gcg_attack=False, eps=64 / 255, alpha=4 / 255), None),
    "llava_gcg": ("llava", dict(num_steps=4, search_width=24, topk=32, pgd_attack=False, gcg_attack=True), None),
    "llava_pgd_gcg": ("llava", dict(num_steps=3, search_width=24, topk=32, pgd_attack=True, gcg_attack=True,
                                    joint_eval=False, eps=64 / 255, alpha=4 / 255), None),
    "llava_joint": ("llava", dict(num_steps=3, search_width=24, topk=32, pgd_attack=True, gcg_attack=True,
                                  joint_eval=True, eps=64 / 255, alpha=4 / 255), None),
    "llava_joint_dyn": ("llava", dict(num_steps=4, search_width=24, topk=16, pgd_attack=True, gcg_attack=True,
                                      joint_eval=True, dynamic_search=True, min_search_width=8, n_replace=2,
                                      buffer_size=3, eps=0.1, alpha=0.05), None),
    "gemma3_joint": ("gemma3", dict(num_steps=3, search_width=16, topk=32, pgd_attack=True, gcg_attack=True,
                                    joint_eval=True, eps=64 / 255, alpha=4 / 255), None),
    "gemma3_pgd_gcg": ("gemma3", dict(num_steps=2, search_width=16, topk=32, pgd_attack=True, gcg_attack=True,
                                      joint_eval=False, eps=64 / 255, alpha=4 / 255), None),
    # BASELINE configs[4] in miniature: Gemma-3 segment order (:1150-1163) with the decaying width (:919-923)
    "gemma3_joint_dyn": ("gemma3", dict(num_steps=5, search_width=24, topk=32, pgd_attack=True, gcg_attack=True,
                                        joint_eval=True, dynamic_search=True, min_search_width=8,
                                        eps=64 / 255, alpha=4 / 255), None),
    # PGD-only on Gemma-3: the step loss comes from the gemma order with the SCALED embedding (:1150-1163, :1142)
    # while the gradient pass uses the llava order and the unscaled table (:968, :981-991)
    "gemma3_pgd": ("gemma3", dict(num_steps=3, pgd_attack=True, gcg_attack=False, eps=64 / 255, alpha=4 / 255), None),
    # early_stop=True runs that DO stop (:1300-1306, :785-787): the target is found by `_early_target`
    "llava_gcg_early": ("llava", dict(num_steps=6, search_width=24, topk=32, pgd_attack=False, gcg_attack=True,
                                      early_stop=True), "early"),
    "llava_pgd_gcg_early": ("llava", dict(num_steps=6, search_width=24, topk=32, pgd_attack=True, gcg_attack=True,
                                          joint_eval=False, early_stop=True, eps=64 / 255, alpha=4 / 255), "early"),
    "llava_joint_early": ("llava", dict(num_steps=6, search_width=24, topk=32, pgd_attack=True, gcg_attack=True,
                                        joint_eval=True, early_stop=True, eps=64 / 255, alpha=4 / 255), "early"),
}


def _run_reference(kind, over, goal, target, trace=True):
    import tempfile

    model, tok, proc, image = S.tiny_case(kind)
    if kind != "opt":
        _adapt_image_features(model)
    tmp = tempfile.mkdtemp(prefix="bma_golden_")
    cfg = ref.BimodalAttackConfig(seed=1, verbosity="ERROR", optim_str_init=S.TINY_OPTIM_INIT,
                                  images_folder=tmp, **over)
    norm = S.Normalize(S.CLIP_MEAN, S.CLIP_STD)
    tr = _Trace() if trace else None
    if tr is not None:
        tr.install()
    try:
        res = ref.run(model, tok, proc, goal, goal, target, image, cfg, normalize=norm)
    finally:
        if tr is not None:
            tr.uninstall()
    return res, tr, model, tmp


def _early_target(kind, over, goal):
    """A target for which the reference's early_stop fires in the MIDDLE of the run: the first
    one-word target, in vocabulary order, whose run ends after step 1..num_steps-2 (so the
    initial suffix does not match, a later candidate does, and steps remain to be skipped)."""
    tok = S.build_tokenizer(S.TINY_WORDS, S.TINY_NONASCII, S.TINY_UNRT)
    words = [w for w, i in sorted(tok.get_vocab().items(), key=lambda kv: kv[1])
             if i >= len(S.SPECIALS) and " " not in w and w.isascii() and not w.startswith("<")]
    for w in words:
        res, _, _, _ = _run_reference(kind, over, goal, w, trace=False)
        if 2 <= len(res.losses) <= over["num_steps"] - 1:
            return w
    raise RuntimeError(f"no one-word target stops {kind} {over} mid-run")


def g5_trajectories(only=None) -> None:
    meta_path = os.path.join(HERE, "g5_meta.json")
    meta = {}
    if only and os.path.exists(meta_path):
        meta = json.load(open(meta_path))["cases"]
    for name, (kind, over, special) in TRAJ.items():
        if only and name not in only:
            continue
        goal, target = "tell me a story about cats", "Sure here is a story"
        if special == "early":
            target = _early_target(kind, over, goal)
            print(f"{name}: early-stop target {target!r}")
        res, tr, model, tmp = _run_reference(kind, over, goal, target)
        if special == "early":
            assert 2 <= len(res.losses) < over["num_steps"], len(res.losses)
        arrays = dict(
            losses=np.array(res.losses, np.float64), best_loss=np.array(res.best_loss),
            init_losses=tr.init_losses, state_checksum=np.array(S.state_checksum(model)),
        )
        for i, st in enumerate(tr.steps):
            arrays[f"s{i}_n_grad"] = np.array(st["n_grad"])
            arrays[f"s{i}_optim_ids_in"] = st["optim_ids_in"]
            for j, g in enumerate(st["grad_tok"]):
                arrays[f"s{i}_grad_tok{j}"] = g
            for j, g in enumerate(st["grad_img"]):
                arrays[f"s{i}_grad_img{j}"] = g
            for j, l in enumerate(st["losses"]):
                arrays[f"s{i}_loss{j}"] = l
            for k in ("sampled", "filtered", "image_after_pgd"):
                if k in st:
                    arrays[f"s{i}_{k}"] = st[k]
        save(f"g5_{name}.npz", **arrays)
        meta[name] = dict(kind=kind, config=over, steps=len(tr.steps), strings=res.strings,
                          best_string=res.best_string, adversarial_suffixes=res.adversarial_suffixes,
                          goal=goal, target=target, optim_str_init=S.TINY_OPTIM_INIT,
                          n_timing=[len(res.gradient_times), len(res.sampling_times), len(res.loss_times),
                                    len(res.pgd_times), len(res.total_times)])
        # the per-step PNGs the reference wrote (:744): keep step 0's pixels as a fixture
        png0 = os.path.join(tmp, "0.png")
        if os.path.exists(png0):
            from PIL import Image
            arrays_png = np.array(Image.open(png0))
            np.savez_compressed(os.path.join(HERE, f"g5_{name}_png0.npz"), png=arrays_png)
    with open(os.path.join(HERE, "g5_meta.json"), "w") as f:
        json.dump(dict(tiny=dict(words=S.TINY_WORDS, nonascii=S.TINY_NONASCII, unroundtrippable=S.TINY_UNRT,
                                 extra_rows=S.TINY_EXTRA_ROWS, std=S.TINY_STD), cases=meta,
                       versions=dict(torch=torch.__version__, transformers=transformers.__version__,
                                     numpy=np.__version__)), f, indent=1)
    print("wrote g5_meta.json")


# --------------------------------------------------------------------------
# G6 -- get_nonascii_toks (reference utils.py:14-33), filter_ids (:166-186)
# --------------------------------------------------------------------------
def g6_tokens() -> None:
    tok = S.build_tokenizer(S.TINY_WORDS, S.TINY_NONASCII, S.TINY_UNRT)
    na = refutils.get_nonascii_toks(tok)
    rs = np.random.RandomState(6)
    ids = rs.randint(0, S.TINY_WORDS, size=(64, 8)).astype(np.int64)
    kept = refmod.filter_ids(torch.from_numpy(ids), tok)
    save("g6_tokens.npz", not_allowed=na.numpy(), ids=ids, kept=kept.numpy())


# --------------------------------------------------------------------------
# G7 -- experiment artefacts (reference experiments.py:54-285, utils/experiments_utils.py:26-71)
# --------------------------------------------------------------------------
def g7_artifacts() -> None:
    """Drive the reference's run_experiment() with canned attack results (its `bimodalattack.run`
    replaced by a stub that returns them) and keep the text of every file it writes.  The
    harness module needs matplotlib / torchvision / requests at import: stubbed; it also reads
    data/advbench relative to the working directory and creates ./experiments: run from a
    scratch directory with a symlink to the reference's data."""
    import tempfile

    work = tempfile.mkdtemp(prefix="bma_g7_")
    os.symlink(os.path.join(REF, "data"), os.path.join(work, "data"))
    for name in ("matplotlib", "matplotlib.pyplot", "torchvision.transforms", "requests"):
        if name not in sys.modules:
            m = types.ModuleType(name)
            m.__spec__ = importlib.machinery.ModuleSpec(name, None)
            m.__path__ = []
            sys.modules[name] = m
    plt = sys.modules["matplotlib.pyplot"]
    for fn in ("figure", "plot", "xlabel", "ylabel", "title", "savefig", "close"):
        setattr(plt, fn, lambda *a, **k: None)
    plt.gca = lambda: SimpleNamespace(text=lambda *a, **k: None, transAxes=None)
    sys.modules["matplotlib"].pyplot = plt
    cwd = os.getcwd()
    os.chdir(work)
    try:
        import experiments as exp   # the reference's harness module
        canned = [
            dict(best_loss=1.25, best_string="x y z", losses=[3.5, 1.25, 2.0], strings=["a", "x y z", "b"],
                 adversarial_suffixes=["a", "x y z", "b"], model_outputs=["out, with comma", "", "line\nbreak"],
                 gradient_times=[0.5, 0.25, 0.125], sampling_times=[0.01, 0.02, 0.03], loss_times=[1.0, 2.0, 3.0],
                 pgd_times=[0.001, 0.002, 0.003], total_times=[1.511, 2.272, 3.158]),
            RuntimeError("boom"),          # a failed prompt becomes a NaN row (:116-137)
            dict(best_loss=0.5, best_string='quote " inside', losses=[0.75, 0.5], strings=["q", 'quote " inside'],
                 adversarial_suffixes=["q", 'quote " inside'], model_outputs=["", ""],
                 gradient_times=[0.5, 0.5, 0.5, 0.5], sampling_times=[0.25, 0.25], loss_times=[1.5, 1.5],
                 pgd_times=[], total_times=[2.25, 2.25]),
        ]
        kwargs = {"num_steps": 3, "search_width": 16, "dynamic_search": False, "min_search_width": 8,
                  "pgd_attack": True, "gcg_attack": True, "alpha": 4 / 255, "eps": 64 / 255, "debug_output": False,
                  "alpha_str": "4/255", "eps_str": "64/255", "joint_eval": True, "model": "llava"}
        pairs = [("goal one", "target one"), ("goal, two", "target two"), ("goal three", 'target "three"')]
        exp.model = exp.tokenizer = exp.processor = exp.image = exp.normalize = None
        experiments = []
        # experiment 1: the middle prompt fails; experiment 2: every prompt succeeds
        for name, plan, prs in (("golden run", canned, pairs), ("clean run", [canned[0], canned[2]], [pairs[0], pairs[2]])):
            it = iter(plan)

            def fake_run(*a, **k):
                c = next(it)
                if isinstance(c, Exception):
                    raise c
                return exp.bimodalattack.BimodalAttackResult(**c)

            exp.bimodalattack.run = fake_run
            exp.run_experiment(name, kwargs, prs)
            folder = os.path.join(work, "experiments", f"exp{len(experiments) + 1}")
            files = {}
            for fn in sorted(os.listdir(folder)):
                path = os.path.join(folder, fn)
                if os.path.isfile(path) and not fn.endswith(".png"):
                    files[fn] = open(path, newline="").read()
            dirs = sorted(d for d in os.listdir(folder) if os.path.isdir(os.path.join(folder, d)))
            experiments.append(dict(name=name, pairs=prs, files=files, dirs=dirs, folder=f"exp{len(experiments) + 1}",
                                    canned=[c if isinstance(c, dict) else {"raise": str(c)} for c in plan]))
    finally:
        os.chdir(cwd)
    with open(os.path.join(HERE, "g7_artifacts.json"), "w") as f:
        json.dump(dict(config_kwargs=kwargs, seed=1, experiments=experiments), f, indent=1)
    print("wrote g7_artifacts.json:", [(e["folder"], sorted(e["files"]), e["dirs"]) for e in experiments])


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7"]
    torch.set_num_threads(1)  # one reduction order
    if "g1" in which:
        g1_sampling()
    if "g2" in which:
        g2_ce()
    if "g3" in which:
        g3_pgd()
    if "g4" in which:
        g4_splice()
    if "g6" in which:
        g6_tokens()
    if "g5" in which:
        g5_trajectories([w for w in which if w in TRAJ] or None)
    if "g7" in which:
        g7_artifacts()
