"""ORACLE -- test infrastructure, not product code.

CPU restatement of the reference's attack loop (``BimodalAttack.run`` and its
helpers, /root/reference/bimodalattack/bimodal_attack.py:193-1338) in the
build's own words: HuggingFace model calls through the same plugin points,
torch-CPU autograd for the gradient pass, and ``oracle.kernels`` (numpy) for
sampling, splice, cross-entropy and the PGD step.  It does what the reference
does -- full (B,S,V) logits, no prefix sharing, one tokenizer call per
candidate -- because it is the checker and the CPU baseline, not the product.

Parity status: PINNED by ``tests/golden/g5_*.npz`` (whole trajectories of the
real reference on tiny random models: sampled ids, survivors of the filter,
per-candidate losses, images after every PGD step).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this.
"""

from __future__ import annotations

import copy
import os
import time
from types import SimpleNamespace
from typing import List, Optional

import numpy as np
import torch

from . import kernels as K

INIT_CHARS = [".", ",", "!", "?", ";", ":", "(", ")", "[", "]", "{", "}",
              "@", "#", "$", "%", "&", "*", "w", "x", "y", "z"]  # reference utils.py:8-12

TEMPLATE_PGD = "USER: <image>\n{{ messages[0]['content'][0]['text'] }} \nASSISTANT: "   # :238
TEMPLATE_GCG = "{% for message in messages %}{{ message['content'] }}{% endfor %}"        # :245-247


def _features_tensor(out) -> torch.Tensor:
    """get_image_features returned a (1,N,D) tensor in transformers 4.50 and an
    output object (pooler_output: tensor or list of (N,D)) in 5.x."""
    if torch.is_tensor(out):
        return out
    p = out.pooler_output
    return p if torch.is_tensor(p) else torch.stack(list(p))


class _Buffer:
    """:91-124 -- keeps (loss, ids) sorted by loss; size 0 keeps only the latest."""

    def __init__(self, size: int):
        self.size, self.items = size, []

    def add(self, loss, ids) -> None:
        if self.size == 0:
            self.items = [(loss, ids)]
            return
        if len(self.items) < self.size:
            self.items.append((loss, ids))
        else:
            self.items[-1] = (loss, ids)
        self.items.sort(key=lambda it: float(it[0]))

    best_ids = property(lambda self: self.items[0][1])
    lowest = property(lambda self: self.items[0][0])
    highest = property(lambda self: self.items[-1][0])


class OracleAttack:
    def __init__(self, model, tokenizer, processor, config, normalize=None):
        self.model, self.tok, self.proc, self.cfg, self.normalize = model, tokenizer, processor, config, normalize
        self.emb = model.get_input_embeddings()
        self.not_allowed = None if config.allow_non_ascii else K.nonascii_tokens(tokenizer)
        self.gemma_proc = processor.__class__.__name__ == "Gemma3Processor"
        self.stop = False
        self.trace: List[dict] = []
        if not getattr(tokenizer, "chat_template", None):                     # :233-249
            tpl = TEMPLATE_PGD if config.pgd_attack else TEMPLATE_GCG
            tokenizer.chat_template = tpl
            processor.chat_template = tpl

    # ------------------------------------------------------------------ setup
    def _ids(self, text, specials: bool) -> torch.Tensor:
        kw = {} if specials else {"add_special_tokens": False}
        return self.tok(text, padding=False, return_tensors="pt", **kw)["input_ids"].to(self.model.device, torch.int64)

    def _prepare(self, messages, target):
        cfg, tok = self.cfg, self.tok
        msgs = [{"role": "user", "content": messages}] if isinstance(messages, str) else copy.deepcopy(messages)
        last = msgs[-1]
        if isinstance(last["content"], str) and "{optim_str}" not in last["content"]:
            last["content"] = last["content"] + " {optim_str}"                  # :283-287
        if cfg.pgd_attack:                                                     # :290-301
            if isinstance(last["content"], str):
                last["content"] = [{"type": "text", "text": last["content"]}, {"type": "image"}]
            elif isinstance(last["content"], list) and not any(it.get("type") == "image" for it in last["content"]):
                last["content"].append({"type": "image"})
        prompt = self.proc.apply_chat_template(msgs, add_generation_prompt=True)
        if tok.bos_token and prompt.startswith(tok.bos_token):
            prompt = prompt.replace(tok.bos_token, "")                          # :309-310
        seg = {}
        if cfg.pgd_attack:
            if self.gemma_proc:                                                # :314-331
                head, tail = prompt.split("{optim_str}", 1)
                if "<start_of_image>" not in tail:
                    raise ValueError("Expected <start_of_image> token in Gemma PGD prompt.")
                mid, sep, rest = tail.partition("<start_of_image>")
                s_before_img, s_before_suffix, s_after = head.strip(), (mid + sep).strip(), rest.strip()
            else:                                                              # :332-339
                for marker in ("<start_of_image>", "<image>"):
                    if marker in prompt:
                        s_before_img, rest = prompt.split(marker, 1)
                        break
                else:
                    raise ValueError("No image token found in prompt for PGD attack")
                s_before_suffix, s_after = rest.split("{optim_str}", 1)
            seg_ids = dict(before_img=self._ids(s_before_img, True), before_suffix=self._ids(s_before_suffix, True),
                           after=self._ids(s_after, False))
        else:
            s_before, s_after = prompt.split("{optim_str}")                     # :359
            seg_ids = dict(before=self._ids(s_before, True), after=self._ids(s_after, False))
        self.target_ids = self._ids(target, False)
        seg_ids["target"] = self.target_ids
        with torch.no_grad():
            self.seg = {k: self.emb(v) for k, v in seg_ids.items()}             # :373-393
        self.seg_ids = seg_ids

    # ------------------------------------------------------------- primitives
    def _image_features(self, image):
        px = self.normalize(image)
        if self.gemma_proc:
            return _features_tensor(self.model.get_image_features(pixel_values=px))
        return _features_tensor(self.model.get_image_features(
            pixel_values=px, vision_feature_layer=-2, vision_feature_select_strategy="default"))

    def _splice(self, ids: torch.Tensor, image_features, search_width, mode, **flags) -> torch.Tensor:
        order = K.segment_order(mode, self.model.config.model_type, **flags)
        parts = []
        for name in order:
            if name == "optim":
                t = self.emb(ids)
            elif name == "image":
                t = image_features
            else:
                t = self.seg[name]
            if search_width is not None and t.shape[0] == 1:
                t = t.repeat(search_width, 1, 1)
            parts.append(t)
        return torch.cat(parts, dim=1)

    def _score(self, chunk: int, embeds: torch.Tensor) -> torch.Tensor:
        """:1278-1310 -- full logits, CE over the T rows in front of each target token."""
        T = self.target_ids.shape[1]
        labels = self.target_ids[0].cpu().numpy()
        out = []
        for s in range(0, embeds.shape[0], chunk):
            with torch.no_grad():
                logits = self.model(inputs_embeds=embeds[s:s + chunk]).logits
            sl = logits[:, embeds.shape[1] - T - 1:-1, :].float().cpu().numpy()
            loss, match = K.ce_target(sl, labels)
            if self.cfg.early_stop and bool(match.any()):
                self.stop = True
            out.append(torch.from_numpy(loss).to(logits.dtype))
        return torch.cat(out).to(embeds.device)

    def _gradient(self, optim_ids: torch.Tensor, image=None):
        """:953-1028 -- one forward/backward; token gradient w.r.t. the one-hot, image
        gradient w.r.t. the pixels."""
        cfg, model = self.cfg, self.model
        V = self.emb.num_embeddings
        onehot = torch.nn.functional.one_hot(optim_ids, num_classes=V).to(model.device, model.dtype)
        if cfg.gcg_attack:
            onehot.requires_grad_()
        optim = onehot @ self.emb.weight                                       # unscaled, also for Gemma (:968)
        if cfg.pgd_attack:
            parts = [self.seg["before_img"], self._image_features(image), self.seg["before_suffix"], optim,
                     self.seg["after"], self.seg["target"]]                    # llava order for every model (:981-991)
        else:
            parts = [self.seg["before"], optim, self.seg["after"], self.seg["target"]]
        x = torch.cat(parts, dim=1)
        logits = model(inputs_embeds=x).logits
        T = self.target_ids.shape[1]
        sl = logits[0, x.shape[1] - T - 1:-1, :]
        loss = torch.nn.functional.cross_entropy(sl, self.target_ids[0])
        wanted = ([onehot] if cfg.gcg_attack else []) + ([image] if cfg.pgd_attack else [])
        grads = list(torch.autograd.grad(loss, wanted))
        g_tok = grads.pop(0) if cfg.gcg_attack else None
        g_img = grads.pop(0) if cfg.pgd_attack else None
        return g_tok, g_img

    def _init_buffer(self, image) -> _Buffer:                                   # :826-906
        cfg = self.cfg
        buf = _Buffer(cfg.buffer_size)
        if isinstance(cfg.optim_str_init, str):
            first = self.tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(self.model.device)
            if cfg.buffer_size > 1:
                pool = self.tok(INIT_CHARS, add_special_tokens=False, return_tensors="pt")["input_ids"].squeeze().to(self.model.device)
                pick = torch.randint(0, pool.shape[0], (cfg.buffer_size - 1, first.shape[1]))
                ids = torch.cat([first, pool[pick]], dim=0)
            else:
                ids = first
        else:
            ids = self.tok(cfg.optim_str_init, add_special_tokens=False, return_tensors="pt")["input_ids"].to(self.model.device)
        n = max(1, cfg.buffer_size)
        with torch.no_grad():
            if cfg.pgd_attack:
                x = self._splice(ids, self._image_features(image), n, "gcg_pgd", single=True)
            else:
                x = self._splice(ids, None, n, "gcg", no_joint_eval=True)
        losses = self._score(n, x)
        self.init_losses = losses.clone()
        for i in range(n):
            buf.add(losses[i], ids[[i]])
        return buf

    # ------------------------------------------------------------------- run
    def run(self, messages, goal, target, image=None):
        from transformers import set_seed

        cfg, tok, model = self.cfg, self.tok, self.model
        os.makedirs(cfg.images_folder, exist_ok=True)
        if cfg.seed is not None:
            set_seed(cfg.seed)
            torch.use_deterministic_algorithms(True, warn_only=True)
        self._prepare(messages, target)
        buf = self._init_buffer(image)
        optim_ids = buf.best_ids

        R = SimpleNamespace(losses=[], strings=[], suffixes=[], outputs=[], t_grad=[], t_samp=[], t_loss=[],
                            t_pgd=[], t_total=[])
        if cfg.pgd_attack:
            image.requires_grad = True                                         # caller's tensor (:425)
            image0 = image.clone()
        if cfg.pgd_after_gcg:
            raise TypeError("unsupported format string passed to NoneType.__format__")  # reference bug at :661

        for i in range(cfg.num_steps):
            st = dict(optim_ids_in=optim_ids.clone().numpy(), grad_tok=[], grad_img=[], losses=[], n_grad=0)
            self.trace.append(st)

            def grad_pass():
                t0 = time.perf_counter()
                g = self._gradient(optim_ids, image if cfg.pgd_attack else None)
                R.t_grad.append(time.perf_counter() - t0)
                st["n_grad"] += 1
                if g[0] is not None:
                    st["grad_tok"].append(g[0][0].detach().float().numpy().copy())
                if g[1] is not None:
                    st["grad_img"].append(g[1].detach().numpy().copy())
                return g

            g_tok, g_img = grad_pass()                                         # phase A
            t_pgd = 0.0
            if cfg.pgd_attack:                                                 # phase B
                t0 = time.perf_counter()
                y = K.linf_step(image.detach().numpy(), g_img.numpy(), image0.detach().numpy(), cfg.eps, cfg.alpha)
                image = torch.from_numpy(y).requires_grad_()
                t_pgd = time.perf_counter() - t0
                R.t_pgd.append(t_pgd)
                st["image_after_pgd"] = y.copy()
                if cfg.gcg_attack and not cfg.joint_eval:                      # phase C
                    g_tok, g_img = grad_pass()

            # phase D: sampling
            width = K.dynamic_width(i, cfg.search_width, cfg.num_steps, cfg.min_search_width, cfg.dynamic_search)
            t_samp = 0.0
            if cfg.gcg_attack:
                t0 = time.perf_counter()
                n_opt = optim_ids.shape[1]
                rnd = torch.rand((width, n_opt))                               # same generator, same order (:151, :159)
                rank = torch.randint(0, cfg.topk, (width, cfg.n_replace, 1)).squeeze(2)
                sampled = K.sample_ids_from_grad(optim_ids[0].numpy(), g_tok[0].detach().float().numpy(), cfg.topk,
                                                 cfg.n_replace, self.not_allowed, rnd.numpy(), rank.numpy())
                st["sampled"] = sampled.copy()
                if cfg.filter_ids:
                    sampled = K.filter_ids(sampled, tok)
                    st["filtered"] = sampled.copy()
                sampled = torch.from_numpy(sampled)
                t_samp = time.perf_counter() - t0
                R.t_samp.append(t_samp)
            else:
                sampled = optim_ids
            n = sampled.shape[0]

            # phase D: scoring
            t0 = time.perf_counter()
            chunk = n if cfg.batch_size is None else cfg.batch_size
            with torch.no_grad():
                if cfg.pgd_attack:
                    feats = self._image_features(image)
                    if cfg.joint_eval:
                        loss = self._score(chunk, self._splice(sampled, feats, n, "pgd", single=True))
                        best = int(loss.argmin())
                        st["losses"].append(loss.float().numpy().copy())
                    elif cfg.gcg_attack:
                        loss = self._score(chunk, self._splice(sampled, None, n, "gcg", single=True))
                        best = int(loss.argmin())
                        st["losses"].append(loss.float().numpy().copy())
                    else:
                        best = 0
                    full = self._score(1, self._splice(sampled[best:best + 1], feats, None, "gcg_pgd"))
                    st["losses"].append(full.float().numpy().copy())
                    current = full.item()
                else:
                    loss = self._score(chunk, self._splice(sampled, None, n, "gcg", no_joint_eval=True))
                    st["losses"].append(loss.float().numpy().copy())
                    current = loss.min().item()
                    best = int(loss.argmin())
                optim_ids = sampled[best:best + 1]
                st["best_idx"], st["current_loss"] = best, current
                R.losses.append(current)
                R.strings.append(tok.batch_decode(optim_ids)[0])
                if buf.size == 0 or current < float(buf.highest):
                    buf.add(current, optim_ids)
            t_loss = time.perf_counter() - t0
            R.t_loss.append(t_loss)

            if cfg.pgd_attack:                                                 # :744, :1312-1317
                from PIL import Image
                arr = (image.squeeze(0).detach().cpu().numpy().transpose(1, 2, 0) * 255).astype(np.uint8)
                Image.fromarray(arr).save(os.path.join(cfg.images_folder, f"{i}.png"))
            R.outputs.append("")
            R.suffixes.append(tok.batch_decode(optim_ids)[0])
            if self.stop:
                break
            R.t_total.append(R.t_grad[-1] + t_samp + t_pgd + t_loss)

        k = R.losses.index(min(R.losses))
        self.final_image = image
        return dict(best_loss=R.losses[k], best_string=R.strings[k], losses=R.losses, strings=R.strings,
                    adversarial_suffixes=R.suffixes, model_outputs=R.outputs, gradient_times=R.t_grad,
                    sampling_times=R.t_samp, loss_times=R.t_loss, pgd_times=R.t_pgd, total_times=R.t_total)


def run_oracle(model, tokenizer, processor, messages, goal, target, image=None, config=None, normalize=None):
    """Same call shape as the reference's ``run`` (:1323-1338); returns (result dict, trace, attack)."""
    atk = OracleAttack(model, tokenizer, processor, config, normalize)
    res = atk.run(messages, goal, target, image)
    return res, atk.trace, atk
