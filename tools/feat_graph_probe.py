#!/usr/bin/env python3
"""Which op of the captured image-features graph reads memory that is no longer its own?  Capture hf.image_features with
every module output kept alive, replay, poison freed memory with NaN, replay again and report the first module
whose output turned non-finite."""
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")
import torch

import bench
from bimodalattack_amd.attack import _ReplayGraph
from bimodalattack_amd.hf_adapter import HFAdapter

dev = torch.device("cuda", 0)
model, tok, proc, messages, goal, target, image, norm = bench.build_plugins("gemma_joint", dev, torch.bfloat16, 2, share=True)
hf = HFAdapter(model, proc, norm)
for name, val in (a.split("=") for a in sys.argv[1:]):
    setattr(hf, name, val == "1")
print("options:", hf.pad_vision_heads, hf.fuse_quick_gelu, hf.fuse_tower_qkv, flush=True)

rec = []
on = {"v": False}


def hook(name):
    def h(mod, args, out):
        if on["v"]:
            t = out[0] if isinstance(out, tuple) else out
            if torch.is_tensor(t):
                rec.append((name, t))
    return h


for name, m in model.named_modules():
    if ("vision_tower" in name or "multi_modal_projector" in name) and len(list(m.children())) == 0:
        m.register_forward_hook(hook(name))

with torch.no_grad():
    want = hf.image_features(image)
print("eager bad:", int((~torch.isfinite(want.float())).sum()), flush=True)


def fn(img):
    return hf.image_features(img)


class G(_ReplayGraph):
    pass


# capture with the recorder on only inside the capture: patch _ReplayGraph's capture by wrapping fn
state = {"n": 0}


def fn_rec(img):
    state["n"] += 1
    on["v"] = state["n"] == 2          # call 1 = warm-up, call 2 = the capture
    try:
        return hf.image_features(img)
    finally:
        on["v"] = False


g = _ReplayGraph(dev, fn_rec, image)
print("captured; recorded", len(rec), "module outputs", flush=True)


def report(tag):
    torch.cuda.synchronize()
    out = g.out
    print(f"{tag}: out bad {int((~torch.isfinite(out.float())).sum())} of {out.numel()}", flush=True)
    shown = 0
    for name, t in rec:
        nb = int((~torch.isfinite(t.float())).sum())
        if nb:
            bad = ~torch.isfinite(t.float())
            rows = bad.reshape(-1, bad.shape[-1]).any(1).nonzero().flatten()
            print(f"   first bad: {name} shape {tuple(t.shape)} bad {nb} rows {int(rows.min())}..{int(rows.max())} x{rows.numel()}", flush=True)
            shown += 1
            if shown >= 3:
                break


g(image)
report("replay 1 (same image)")
img2 = torch.rand_like(image)
g(img2)
report("replay 2 (new image)")
with torch.no_grad():
    e2 = hf.image_features(img2)
torch.cuda.synchronize()
print("   eager vs graph max diff:", float((e2.float() - g.out.float()).abs().max()), flush=True)
# poison whatever is free
junk = []
for mb in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    for _ in range(4):
        try:
            junk.append(torch.full((mb << 18,), float("nan"), device=dev))
        except Exception:
            break
small = [torch.full((n,), float("nan"), device=dev) for n in (64, 128, 512, 2048, 9728, 19 * 384, 19 * 128, 65536) for _ in range(8)]
del junk, small
g(img2)
report("replay 3 (after NaN-poisoning freed memory)")
torch.cuda.empty_cache()
junk = [torch.full((1 << 26,), float("nan"), device=dev) for _ in range(8)]
del junk
g(img2)
report("replay 4 (after empty_cache + poisoning)")

# ---------------------------------------------------------------------------------------------------------------
# the same inside the engine's own order of captures: prefix graph (init_buffer), gradient graph, feature graph
print("=== engine order", flush=True)
import tempfile
from bimodalattack_amd import BimodalAttackConfig
from bimodalattack_amd.attack import BimodalAttack, logger
from bimodalattack_amd.config import EngineOptions
logger.setLevel("ERROR")
cfg = BimodalAttackConfig(num_steps=3, search_width=64, topk=256, seed=1, verbosity="ERROR", pgd_attack=True, gcg_attack=True,
                          joint_eval=True, eps=64 / 255, alpha=4 / 255, images_folder=tempfile.mkdtemp())
opts = {}
if os.environ.get("PROBE_NO_GRAD_GRAPH"):
    opts["graph_gradient"] = False
attack = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, **opts))
rec.clear()
hf2 = attack.hf
orig_feats = hf2.image_features
flag = {"armed": False}


def feats_rec(img):
    on["v"] = bool(flag["armed"] and torch.cuda.is_current_stream_capturing())
    try:
        return orig_feats(img)
    finally:
        on["v"] = False


hf2.image_features = feats_rec
img = image.detach().clone()
attack._prepare_prompt(messages, target)
buf = attack.init_buffer(img)
ids = buf.get_best_ids()
img.requires_grad_(True)
with torch.enable_grad():
    g_tok, g_img, _ = attack.compute_gradient(ids, img)
torch.cuda.synchronize()
print("gradient: g_img bad", int((~torch.isfinite(g_img)).sum()), "graphs", attack.graphs_captured, flush=True)
flag["armed"] = True
with torch.no_grad():
    f = attack.scoring_features(img)
torch.cuda.synchronize()
flag["armed"] = False
g = attack._feat_graph
print("feature graph captured:", g not in (None, False), "recorded", len(rec), "graphs", attack.graphs_captured, flush=True)
report("engine replay 0")
for k in range(3):
    with torch.enable_grad():
        attack.compute_gradient(ids, img)
    img2 = torch.rand_like(image)
    junk = [torch.full((n,), float("nan"), device=dev) for n in (19 * 384, 19 * 128, 19 * 460, 65536, 1 << 20, 1 << 24) for _ in range(4)]
    del junk
    with torch.no_grad():
        attack.scoring_features(img2)
    report(f"engine replay {k + 1} (gradient pass in between, poisoned)")
    with torch.no_grad():
        e2 = orig_feats(img2)
    torch.cuda.synchronize()
    print("   eager bad", int((~torch.isfinite(e2.float())).sum()), "max diff vs graph", float((e2.float() - g.out.float()).nan_to_num(1e9).abs().max()), flush=True)
