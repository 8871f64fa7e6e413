"""Synthetic plugins for tests, smoke and bench: tokenizer, processor, models.

There is no network, so no checkpoint or tokenizer file exists anywhere.  This
module builds, in process, the three plugin objects the engine's boundary takes
 -- a HuggingFace fast tokenizer, a processor shim whose
``apply_chat_template`` returns a string, and random-weight HuggingFace models
of the public config shapes (top) -- plus a ``Normalize`` callable
standing in for ``torchvision.transforms.Normalize`` (reference
experiments.py:383-397).

Weights never come from the framework's RNG: every parameter is overwritten in
sorted-name order from a counter-based hash so the same bytes appear in this
container, on the GPU box, on CPU and on the device.
"""

from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import numpy as np
import torch

SPECIALS = ["<pad>", "<s>", "</s>", "<unk>", "<image>"]
PAD_ID, BOS_ID, EOS_ID, UNK_ID, IMAGE_ID = range(5)

# Words the default prompts/targets of tests and bench are made of.
_BASE_WORDS = [
    "x", ".", ",", "!", "?", ";", ":", "(", ")", "[", "]", "{", "}",
    "@", "#", "$", "%", "&", "*", "w", "y", "z",
    "USER:", "ASSISTANT:", "Sure", "here", "is", "a", "the", "plan", "to", "do",
    "write", "tell", "me", "how", "story", "about", "cats", "and", "dogs", "please",
    "<start_of_turn>user", "<start_of_turn>model", "<end_of_turn>", "<start_of_image>",
]


def build_vocab(n_words: int, n_nonascii: int = 0, n_unroundtrippable: int = 0) -> Dict[str, int]:
    """Word-level vocabulary: specials, base words, filler words, then (optionally)
    tokens that must be forbidden (non-ASCII / non-printable) and tokens that are
    printable ASCII but cannot survive decode->encode (they contain a space)."""
    vocab: Dict[str, int] = {}
    for tok in SPECIALS + _BASE_WORDS:
        vocab[tok] = len(vocab)
    i = 0
    n_fill = n_words - len(vocab) - n_nonascii - n_unroundtrippable
    if n_fill < 0:
        raise ValueError("n_words too small")
    width = max(3, len(str(max(n_fill - 1, 1))))
    while i < n_fill:
        vocab[f"t{i:0{width}d}"] = len(vocab)
        i += 1
    nonascii_pool = ["é", "ü", "日本", "λ", "naïve", "ctl", "ß", "中"]
    for j in range(n_nonascii):
        vocab[nonascii_pool[j % len(nonascii_pool)] + (str(j) if j >= len(nonascii_pool) else "")] = len(vocab)
    for j in range(n_unroundtrippable):
        vocab[f"ab{j} cd"] = len(vocab)
    assert len(vocab) == n_words, (len(vocab), n_words)
    return vocab


def build_tokenizer(n_words: int = 256, n_nonascii: int = 6, n_unroundtrippable: int = 6):
    """In-process ``PreTrainedTokenizerFast``: WordLevel + WhitespaceSplit, a
    ``<s> $A`` post-processor (so ``add_special_tokens=True`` prepends BOS like
    the Llama tokenizer does -- the reference relies on that, :346-351)."""
    from tokenizers import Tokenizer, models, pre_tokenizers, processors
    from transformers import PreTrainedTokenizerFast

    vocab = build_vocab(n_words, n_nonascii, n_unroundtrippable)
    tok = Tokenizer(models.WordLevel(vocab=vocab, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.WhitespaceSplit()
    tok.post_processor = processors.TemplateProcessing(
        single="<s> $A", pair="<s> $A $B", special_tokens=[("<s>", BOS_ID)]
    )
    fast = PreTrainedTokenizerFast(
        tokenizer_object=tok,
        bos_token="<s>",
        eos_token="</s>",
        unk_token="<unk>",
        pad_token="<pad>",
        clean_up_tokenization_spaces=False,
    )
    return fast


class SyntheticProcessor:
    """Six-line processor shim: only what the engine touches --
    ``chat_template`` (read/write) and ``apply_chat_template`` -> ``str``."""

    def __init__(self, tokenizer, chat_template: Optional[str] = None):
        self.tokenizer = tokenizer
        self.chat_template = chat_template

    def apply_chat_template(self, messages, add_generation_prompt: bool = True, **kw) -> str:
        return self.tokenizer.apply_chat_template(
            messages,
            chat_template=self.chat_template,
            add_generation_prompt=add_generation_prompt,
            tokenize=False,
        )


class Gemma3Processor(SyntheticProcessor):
    """Same shim under the class name the engine dispatches on
    (``processor.__class__.__name__ == "Gemma3Processor"``, reference :314)."""


class Normalize:
    """Differentiable per-channel normalisation on (1,3,H,W) in [0,1]."""

    def __init__(self, mean: Sequence[float], std: Sequence[float]):
        self.mean = torch.tensor(list(mean), dtype=torch.float32).view(1, -1, 1, 1)
        self.std = torch.tensor(list(std), dtype=torch.float32).view(1, -1, 1, 1)

    def __call__(self, x: torch.Tensor) -> torch.Tensor:
        return (x - self.mean.to(x.device, x.dtype)) / self.std.to(x.device, x.dtype)


CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


# --------------------------------------------------------------------------
# deterministic weights
# --------------------------------------------------------------------------
def _hash_normal(n: int, seed: int, device: torch.device) -> torch.Tensor:
    """n pseudo-normal fp32 values from an integer hash of (seed, index): the
    same values on every device and library version (no framework RNG)."""
    idx = torch.arange(n, device=device, dtype=torch.int64)

    def mix(v: torch.Tensor) -> torch.Tensor:
        v = (v ^ (v >> 16)) * 0x45D9F3B & 0xFFFFFFFF
        v = (v ^ (v >> 16)) * 0x45D9F3B & 0xFFFFFFFF
        return (v ^ (v >> 16)) & 0xFFFFFFFF

    a = mix(idx * 2 + 1 + seed * 0x9E3779B1 & 0xFFFFFFFF)
    b = mix(idx * 2 + 2 + seed * 0x85EBCA6B & 0xFFFFFFFF)
    u1 = (a.to(torch.float64) + 1.0) / 4294967297.0
    u2 = b.to(torch.float64) / 4294967296.0
    z = torch.sqrt(-2.0 * torch.log(u1)) * torch.cos(2.0 * math.pi * u2)
    return z.to(torch.float32)


@torch.no_grad()
def fill_deterministic(model: torch.nn.Module, seed: int = 0, std: float = 0.02, chunk: int = 1 << 26) -> None:
    """Overwrite every parameter (sorted by name) with hash-normal values:
    N(0, std^2) for matrices/biases, 1 + N(0, std^2) for norm scales."""
    named = sorted(model.named_parameters(), key=lambda kv: kv[0])
    seen = set()
    for k, (name, p) in enumerate(named):
        if id(p) in seen:  # tied weights
            continue
        seen.add(id(p))
        flat = p.data.view(-1)
        n = flat.numel()
        is_norm_scale = p.dim() == 1 and ("norm" in name.lower() or "ln" in name.lower()) and name.endswith("weight")
        for s in range(0, n, chunk):
            e = min(n, s + chunk)
            z = _hash_normal(e - s, seed * 1000003 + k * 7919 + s // chunk, flat.device) * std
            if is_norm_scale:
                z = z + 1.0
            flat[s:e] = z.to(flat.dtype)
    model.eval()
    for p in model.parameters():
        p.requires_grad_(False)


def state_checksum(model: torch.nn.Module) -> float:
    tot = 0.0
    for name, p in sorted(model.named_parameters(), key=lambda kv: kv[0]):
        tot += float(p.detach().double().abs().sum().cpu())
    return tot


# --------------------------------------------------------------------------
# model builders
# --------------------------------------------------------------------------
def _build(cls, config, dtype, device, seed, std=0.02):
    try:  # skip HF's own (slow, RNG-dependent) init; every byte is overwritten below
        from transformers.initialization import no_init_weights
    except ImportError:  # transformers 4.x
        try:
            from transformers.modeling_utils import no_init_weights
        except ImportError:
            from contextlib import nullcontext as no_init_weights

    prev = torch.get_default_dtype()
    torch.set_default_dtype(dtype)
    try:
        with no_init_weights():
            with torch.device(device):
                model = cls(config)
    finally:
        torch.set_default_dtype(prev)
    model = model.to(dtype=dtype)
    fill_deterministic(model, seed=seed, std=std)
    # non-persistent buffers (rotary inv_freq, position ids) are built by the
    # constructor on `device` already; nothing to load.
    return model


def tiny_opt(vocab_rows: int, dtype=torch.float32, device="cpu", seed: int = 0, std: float = 0.02):
    """2-layer OPT: the text-only surrogate of BASELINE config 1, shrunk."""
    from transformers import OPTConfig, OPTForCausalLM

    cfg = OPTConfig(
        vocab_size=vocab_rows, hidden_size=32, num_hidden_layers=2, ffn_dim=64,
        num_attention_heads=4, max_position_embeddings=160, word_embed_proj_dim=32,
        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, layerdrop=0.0,
        pad_token_id=PAD_ID, bos_token_id=BOS_ID, eos_token_id=EOS_ID,
    )
    cfg._attn_implementation = "eager" if str(device) == "cpu" else "sdpa"
    return _build(OPTForCausalLM, cfg, dtype, device, seed, std)


def opt_125m_shaped(vocab_rows: int, dtype=torch.float32, device="cpu", seed: int = 0):
    """OPT-125M shape (768 / 12 layers / 12 heads / 3072), vocab = tokenizer size."""
    from transformers import OPTConfig, OPTForCausalLM

    cfg = OPTConfig(
        vocab_size=vocab_rows, hidden_size=768, num_hidden_layers=12, ffn_dim=3072,
        num_attention_heads=12, max_position_embeddings=2048, word_embed_proj_dim=768,
        dropout=0.0, attention_dropout=0.0, activation_dropout=0.0, layerdrop=0.0,
        pad_token_id=PAD_ID, bos_token_id=BOS_ID, eos_token_id=EOS_ID,
    )
    cfg._attn_implementation = "eager" if str(device) == "cpu" else "sdpa"
    return _build(OPTForCausalLM, cfg, dtype, device, seed)


def _llava(vocab_rows, text_hidden, text_inter, text_layers, text_heads, vis_hidden, vis_inter,
           vis_layers, vis_heads, image_size, patch, max_pos, dtype, device, seed, attn, std=0.02):
    from transformers import CLIPVisionConfig, LlamaConfig, LlavaConfig, LlavaForConditionalGeneration

    vcfg = CLIPVisionConfig(
        hidden_size=vis_hidden, intermediate_size=vis_inter, num_hidden_layers=vis_layers,
        num_attention_heads=vis_heads, image_size=image_size, patch_size=patch,
        projection_dim=vis_hidden, attention_dropout=0.0,
    )
    tcfg = LlamaConfig(
        vocab_size=vocab_rows, hidden_size=text_hidden, intermediate_size=text_inter,
        num_hidden_layers=text_layers, num_attention_heads=text_heads,
        num_key_value_heads=text_heads, max_position_embeddings=max_pos,
        rms_norm_eps=1e-5, attention_dropout=0.0,
        pad_token_id=PAD_ID, bos_token_id=BOS_ID, eos_token_id=EOS_ID, tie_word_embeddings=False,
    )
    cfg = LlavaConfig(
        vision_config=vcfg, text_config=tcfg, image_token_index=IMAGE_ID,
        vision_feature_layer=-2, vision_feature_select_strategy="default",
        projector_hidden_act="gelu",
    )
    cfg._attn_implementation = attn
    model = _build(LlavaForConditionalGeneration, cfg, dtype, device, seed, std)
    return model


def tiny_llava(vocab_rows: int, dtype=torch.float32, device="cpu", seed: int = 0, std: float = 0.02):
    """2-layer Llama + 3-layer CLIP tower, 28 px / patch 14 -> N_img = 4."""
    attn = "eager" if str(device) == "cpu" else "sdpa"
    return _llava(vocab_rows, 32, 64, 2, 4, 32, 64, 3, 4, 28, 14, 256, dtype, device, seed, attn, std)


def llava_15_7b_shaped(dtype=torch.bfloat16, device="cuda", seed: int = 0, vocab_rows: int = 32064,
                       text_layers: int = 32):
    """LLaVA-1.5-7B shape: Llama-2-7B (4096 / 32 layers / 32 heads / 11008) with
    V = 32064 embedding rows, CLIP-L/14-336 tower (1024 / 24 layers / 16 heads),
    N_img = 576 (top)."""
    return _llava(vocab_rows, 4096, 11008, text_layers, 32, 1024, 4096, 24, 16, 336, 14, 4096,
                  dtype, device, seed, "sdpa")


def _gemma3(vocab_rows, text_hidden, text_inter, text_layers, heads, kv_heads, head_dim, vis_hidden,
            vis_inter, vis_layers, vis_heads, image_size, patch, mm_tokens, sliding, dtype, device, seed, attn,
            std=0.02):
    from transformers import Gemma3Config, Gemma3ForConditionalGeneration, Gemma3TextConfig, SiglipVisionConfig

    vcfg = SiglipVisionConfig(
        hidden_size=vis_hidden, intermediate_size=vis_inter, num_hidden_layers=vis_layers,
        num_attention_heads=vis_heads, image_size=image_size, patch_size=patch,
        attention_dropout=0.0, vision_use_head=False,
    )
    tcfg = Gemma3TextConfig(
        vocab_size=vocab_rows, hidden_size=text_hidden, intermediate_size=text_inter,
        num_hidden_layers=text_layers, num_attention_heads=heads, num_key_value_heads=kv_heads,
        head_dim=head_dim, max_position_embeddings=8192, sliding_window=sliding,
        query_pre_attn_scalar=head_dim, attention_dropout=0.0,
        pad_token_id=PAD_ID, bos_token_id=BOS_ID, eos_token_id=EOS_ID,
    )
    cfg = Gemma3Config(
        text_config=tcfg, vision_config=vcfg, mm_tokens_per_image=mm_tokens,
        image_token_index=IMAGE_ID, boi_token_index=IMAGE_ID, eoi_token_index=IMAGE_ID,
    )
    cfg._attn_implementation = attn
    return _build(Gemma3ForConditionalGeneration, cfg, dtype, device, seed, std)


def tiny_gemma3(vocab_rows: int, dtype=torch.float32, device="cpu", seed: int = 0, std: float = 0.02):
    """2-layer Gemma-3 text + 2-layer SigLIP, 56 px / patch 14 -> 16 patches -> 4 image tokens."""
    attn = "eager" if str(device) == "cpu" else "sdpa"
    return _gemma3(vocab_rows, 32, 64, 2, 4, 2, 8, 32, 64, 2, 4, 56, 14, 4, 64, dtype, device, seed, attn, std)


def gemma3_4b_shaped(dtype=torch.bfloat16, device="cuda", seed: int = 0, vocab_rows: int = 262208,
                     text_layers: int = 34):
    """Gemma-3-4b-it shape: 2560 / 34 layers / 8 heads (4 kv) x 256, FFN 10240,
    SigLIP-So400m 896 px tower (1152 / 27 layers), N_img = 256."""
    return _gemma3(vocab_rows, 2560, 10240, text_layers, 8, 4, 256, 1152, 4304, 27, 16, 896, 14, 256, 1024,
                   dtype, device, seed, "sdpa")


def synthetic_image(h: int, w: int, seed: int = 0, device="cpu") -> torch.Tensor:
    """U[0,1) fp32 image (1,3,H,W) from a seeded CPU generator."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    return torch.rand((1, 3, h, w), generator=g, dtype=torch.float32).to(device)


def synthetic_prompt(tokenizer, n_goal_tokens: int, n_target_tokens: int, seed: int = 0):
    """(goal, target) strings of exactly the requested token counts made of
    filler words, so segment lengths are fixed (top)."""
    vocab = tokenizer.get_vocab()
    fillers = sorted(t for t in vocab if t.startswith("t") and t[1:].isdigit())
    rs = np.random.RandomState(seed)
    goal = " ".join(fillers[int(i)] for i in rs.randint(0, len(fillers), size=n_goal_tokens))
    target = " ".join(fillers[int(i)] for i in rs.randint(0, len(fillers), size=n_target_tokens))
    return goal, target


# --------------------------------------------------------------------------
# the tiny cases the golden trajectories were captured on (tests/golden/g5_*)
# --------------------------------------------------------------------------
TINY_WORDS, TINY_NONASCII, TINY_UNRT, TINY_EXTRA_ROWS = 256, 6, 6, 8
TINY_STD = 0.35          # wide enough that candidate losses differ by >> 1e-4
TINY_OPTIM_INIT = "x x x x x x x x"
GEMMA_TEMPLATE = (
    "{{ bos_token }}<start_of_turn>user\n"
    "{% for item in messages[0]['content'] %}"
    "{% if item['type'] == 'text' %}{{ item['text'] }}{% elif item['type'] == 'image' %}<start_of_image>{% endif %}"
    "{% endfor %}<end_of_turn>\n<start_of_turn>model\n"
)


def tiny_case(kind: str, dtype=torch.float32, device="cpu"):
    """(model, tokenizer, processor, image) for kind in {opt, llava, gemma3}."""
    tok = build_tokenizer(TINY_WORDS, TINY_NONASCII, TINY_UNRT)
    rows = TINY_WORDS + TINY_EXTRA_ROWS   # embedding rows > tokenizer.vocab_size, like 32064 vs 32000
    if kind == "opt":
        model, proc, image = tiny_opt(TINY_WORDS, dtype, device, std=TINY_STD), SyntheticProcessor(tok), None
    elif kind == "llava":
        model, proc = tiny_llava(rows, dtype, device, std=TINY_STD), SyntheticProcessor(tok)
        image = synthetic_image(28, 28, seed=0, device=device)
    elif kind == "gemma3":
        tok.chat_template = GEMMA_TEMPLATE
        model, proc = tiny_gemma3(rows, dtype, device, std=TINY_STD), Gemma3Processor(tok, GEMMA_TEMPLATE)
        image = synthetic_image(56, 56, seed=0, device=device)
    else:
        raise ValueError(kind)
    return model, tok, proc, image
