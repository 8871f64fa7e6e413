#!/usr/bin/env python3
"""Where does a non-finite value first appear in a full-size attack run?  (the Gemma-3-4b joint
workload ended with a NaN loss while the oracle on the same model stays finite.)

    python tools/nan_bisect.py [--workload gemma_joint] [--steps 3] [--width 64] [--variants default,nopad,...]

Runs the workload's engine on bench.py's plugins with the step trace on, once per option variant, and prints for
every step whether the token gradient, the pixel gradient, the image after PGD, the candidate losses and the step
loss are finite (and their extremes).  One JSON document goes to --out."""

from __future__ import annotations

import argparse
import json
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")

# engine options of a variant; keys "module:NAME" set a module constant of bimodalattack_amd for the variant's run
# (the A/B switches that are not engine options: ops.OWN_KERNELS entries as "ops:OWN_KERNELS.skinny_gemm")
VARIANTS = {
    "default": {},
    "nopad": {"hf_adapter:PAD_VISION_HEADS": False},
    "nographs": dict(graph_gradient=False, graph_scoring=False),
    "notowerqkv": {"hf_adapter:FUSE_TOWER_QKV": False},
    "noscoringgraphs": dict(graph_scoring=False),
    "nogradgraph": dict(graph_gradient=False),
    "cpurng": dict(rng_device="cpu"),
    "noaddnorm": dict(fuse_add_norm=False),
    "noqkrope": {"fused:FUSE_QK_ROPE": False},
    "nofused": dict(fused_elementwise=False),
    "noskinny": {"ops:OWN_KERNELS.skinny_gemm": False},
    "noownkernels": dict(own_b1_kernels=False),
    "nocopies": dict(derived_weight_copies=False),
    "nomaskless": {"attack:MASKLESS_B1_ATTENTION": False},
    "plain": {"hf_adapter:PAD_VISION_HEADS": False, "hf_adapter:FUSE_TOWER_QKV": False, "fused:FUSE_QK_ROPE": False,
              "attack:MASKLESS_B1_ATTENTION": False, "graph_gradient": False, "graph_scoring": False, "fuse_add_norm": False,
              "fused_elementwise": False, "own_b1_kernels": False, "derived_weight_copies": False, "shared_prefix_attention": False,
              "ragged_suffix": False, "gradient_ahead": False, "early_plan": False, "gemm_tuning": "off"},
}


def split_variant(v: dict):
    """(engine keyword arguments, a function that sets the variant's module constants and returns the undo function)."""
    import importlib
    kw = {k: val for k, val in v.items() if ":" not in k}
    consts = {k: val for k, val in v.items() if ":" in k}

    def apply():
        undo = []
        for key, val in consts.items():
            mod_name, attr = key.split(":")
            mod = importlib.import_module("bimodalattack_amd." + mod_name)
            if "." in attr:
                table, item = attr.split(".")
                d = getattr(mod, table)
                undo.append((d.__setitem__, item, d[item]))
                d[item] = val
            else:
                undo.append((lambda name, old, m=mod: setattr(m, name, old), attr, getattr(mod, attr)))
                setattr(mod, attr, val)
        return lambda: [f(a, b) for f, a, b in undo]
    return kw, apply


def stats(a) -> dict:
    import numpy as np
    a = np.asarray(a, dtype=np.float64)
    fin = np.isfinite(a)
    return dict(finite=bool(fin.all()), n_bad=int((~fin).sum()), n=int(a.size),
                absmax=float(np.abs(a[fin]).max()) if fin.any() else None)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="gemma_joint")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--width", type=int, default=64)
    ap.add_argument("--layers", type=int, default=32)
    ap.add_argument("--variants", default="default")
    ap.add_argument("--no-trace", action="store_true", help="no step trace (its device-to-host copies synchronise every phase)")
    ap.add_argument("--bench-schedule", action="store_true", help="bench.py's widths: 2 steps at the full width, then the "
                    "600-step schedule sampled evenly over steps-4 steps, then 2 at its middle")
    ap.add_argument("--probe-feats", action="store_true", help="check the image features right after they are made; when "
                    "non-finite: the graph's input, a second replay, the eager function on the same image")
    ap.add_argument("--out", default=os.path.join(REPO, "gpurun_out", "nan_bisect.json"))
    args = ap.parse_args()

    import torch
    import bench
    from bimodalattack_amd import BimodalAttackConfig
    from bimodalattack_amd.attack import BimodalAttack, logger
    from bimodalattack_amd.config import EngineOptions

    logger.setLevel("ERROR")
    dev = torch.device("cuda", 0)
    wl = bench.WORKLOADS[args.workload]
    model, tok, proc, messages, goal, target, image0, norm = bench.build_plugins(args.workload, dev, torch.bfloat16, args.layers, share=True)
    report = {}
    for name in args.variants.split(","):
        kw, apply_constants = split_variant(VARIANTS[name])
        undo_constants = apply_constants()
        trace = None if args.no_trace else []
        score_log = []
        if args.bench_schedule:
            from bimodalattack_amd.layout import dynamic_width
            K = max(1, args.steps - 4)

            def width_of(i, K=K):
                v = 0 if i < 2 else (int(round((i - 2 + 0.5) * 600 / K)) if i < 2 + K else 300)
                return dynamic_width(min(v, 599), args.width, 600, min(128, args.width), True)
            kw["width_override"] = width_of
        cfg = BimodalAttackConfig(num_steps=args.steps, search_width=args.width, topk=256, seed=1, verbosity="ERROR",
                                  pgd_attack=wl["pgd_attack"], gcg_attack=wl["gcg_attack"], joint_eval=wl["joint_eval"],
                                  eps=64 / 255, alpha=4 / 255, images_folder=tempfile.mkdtemp(prefix="bma_nan_"),
                                  dynamic_search=bool(wl.get("gemma")), min_search_width=min(128, args.width))
        image = None if image0 is None else image0.detach().clone()
        attack = BimodalAttack(model, tok, proc, cfg, norm, EngineOptions.from_env(save_images=False, trace=trace, **kw))
        attack.score_log = score_log
        if args.probe_feats:
            orig = attack.scoring_features
            rec, arm = [], {"v": False}
            if not getattr(model, "_bma_probe_hooks", False):
                model._bma_probe_hooks = True
                model._bma_probe = (rec, arm)

                def hook(nm):
                    def h(mod, a, out):
                        r, on = model._bma_probe
                        if on["v"] and torch.cuda.is_current_stream_capturing():
                            t = out[0] if isinstance(out, tuple) else out
                            if torch.is_tensor(t):
                                r.append((nm, t))
                    return h
                for nm, mod in model.named_modules():
                    if ("vision_tower" in nm or "multi_modal_projector" in nm) and len(list(mod.children())) == 0:
                        mod.register_forward_hook(hook(nm))
            model._bma_probe = (rec, arm)

            def probed(img, orig=orig, attack=attack, rec=rec, arm=arm):
                arm["v"] = True
                f = orig(img)
                arm["v"] = False
                bad = ~torch.isfinite(f.float())
                nb = int(bad.sum())
                g = attack._feat_graph
                msg = f"   features: bad={nb} of {f.numel()} shape={tuple(f.shape)} graph={'yes' if g not in (None, False) else 'no'} img_bad={int((~torch.isfinite(img)).sum())}"
                if nb:
                    rows = bad[0].any(1).nonzero().flatten()
                    cols = bad[0].any(0).nonzero().flatten()
                    msg += f" rows[{int(rows.min())}..{int(rows.max())}]x{rows.numel()} cols[{int(cols.min())}..{int(cols.max())}]x{cols.numel()}"
                    if g not in (None, False):
                        msg += f" graph_input_bad={int((~torch.isfinite(g.inputs[0])).sum())} input_eq_img={bool(torch.equal(g.inputs[0], img.detach()))}"
                        g.graph.replay()
                        torch.cuda.synchronize()
                        msg += f" replay_again_bad={int((~torch.isfinite(g.out.float())).sum())}"
                    shown = 0
                    for nm, t in rec:
                        b = ~torch.isfinite(t.float())
                        if bool(b.any()):
                            r2 = b.reshape(-1, b.shape[-1]).any(1).nonzero().flatten()
                            msg += f"\n      bad module output: {nm} {tuple(t.shape)} bad={int(b.sum())} rows {int(r2.min())}..{int(r2.max())} x{r2.numel()}"
                            shown += 1
                            if shown >= 4:
                                break
                    msg += f"\n      recorded {len(rec)} outputs"
                    with torch.no_grad():
                        e = attack.hf.image_features(img.detach())
                    msg += f" eager_bad={int((~torch.isfinite(e.float())).sum())}"
                print(msg, flush=True)
                return f
            attack.scoring_features = probed
        try:
            res = attack.run(messages, goal, target, image)
            err = None
        except Exception as e:        # report and carry on with the next variant
            res, err = None, f"{type(e).__name__}: {e}"
            torch.cuda.synchronize()
        rows = []
        for i, st in enumerate(trace or []):
            row = dict(step=i, n_grad=st.get("n_grad"))
            for k in ("grad_tok", "grad_img", "losses"):
                row[k] = [stats(a) for a in st.get(k, [])]
            if "image_after_pgd" in st:
                row["image_after_pgd"] = stats(st["image_after_pgd"])
            row["current_loss"] = st.get("current_loss")
            rows.append(row)
        calls = [{k: (int(v.item()) if hasattr(v, "item") else v) for k, v in c.items()} for c in score_log]
        undo_constants()
        kw.pop("width_override", None)
        kw.update({k: v for k, v in VARIANTS[name].items() if ":" in k})
        for c in calls:
            print("   " + " ".join(f"{k}={v}" for k, v in c.items()), flush=True)
        rep = dict(options=kw, error=err, score_calls=calls, init_losses=[float(v) for v in attack.init_losses.tolist()] if hasattr(attack, "init_losses") else None,
                   losses=None if res is None else res.losses, steps=rows, engine=attack.engine_state())
        report[name] = rep
        print(f"== {name}: init {rep['init_losses']} losses {rep['losses']} error {err}", flush=True)
        for r in rows:
            def f(lst):
                return "/".join(("ok" if s["finite"] else f"BAD{s['n_bad']}") + (f"({s['absmax']:.3g})" if s["absmax"] is not None else "") for s in lst) or "-"
            print(f"   step {r['step']}: g_tok {f(r['grad_tok'])}  g_img {f(r['grad_img'])}  image "
                  f"{f([r['image_after_pgd']]) if 'image_after_pgd' in r else '-'}  cand {f(r['losses'])}  loss {r['current_loss']}", flush=True)
        del attack
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as fh:
        json.dump(report, fh, indent=1, default=str)


if __name__ == "__main__":
    main()
