#!/usr/bin/env python3
"""Which aten ops (with shapes) one image-features pass + backward of the Gemma-3 SigLIP tower issues through the engine's
context (2 tower layers): finds the copies / adds that are pure launch or bandwidth overhead around the own attention kernels."""
import collections
import os
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
os.environ.setdefault("MIOPEN_FIND_MODE", "FAST")      # no exhaustive search for the patch-embedding convolution
import torch  # noqa: E402
from torch.utils._python_dispatch import TorchDispatchMode  # noqa: E402
from bimodalattack_amd import synthetic as S  # noqa: E402
from bimodalattack_amd.hf_adapter import HFAdapter  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "gemma"
dev = torch.device("cuda", 0)
if kind == "gemma":
    model = S._gemma3(1024, 256, 512, 1, 4, 2, 64, 1152, 4304, 2, 16, 896, 14, 256, 1024, torch.bfloat16, dev, 0, "sdpa")
    tok = S.build_tokenizer(256, 0, 0)
    hf = HFAdapter(model, S.Gemma3Processor(tok, S.GEMMA_TEMPLATE), S.Normalize((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)))
    image = S.synthetic_image(896, 896, seed=0, device=dev)
else:
    model = S._llava(512, 256, 512, 1, 4, 1024, 4096, 3, 16, 336, 14, 4096, torch.bfloat16, dev, 0, "sdpa")
    tok = S.build_tokenizer(256, 0, 0)
    hf = HFAdapter(model, S.SyntheticProcessor(tok), S.Normalize(S.CLIP_MEAN, S.CLIP_STD))
    image = S.synthetic_image(336, 336, seed=0, device=dev)
seen = collections.Counter()


class Rec(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        shapes = tuple(tuple(a.shape) for a in args if torch.is_tensor(a))[:2]
        out = func(*args, **(kwargs or {}))
        if not name.startswith(("view", "_unsafe_view", "t.", "transpose", "unsqueeze", "squeeze", "expand", "detach", "alias", "slice", "select", "split", "permute", "reshape", "as_strided", "unbind")):
            seen[(name, shapes)] += 1
        return out


img = image.clone().requires_grad_()
hf.image_features(img).float().sum().backward()       # lazy initialisations out of the count
img = image.clone().requires_grad_()
with Rec():
    f = hf.image_features(img)
    f.float().sum().backward()
for (name, shapes), n in sorted(seen.items(), key=lambda kv: (-kv[1], kv[0][0])):
    print(f"{n:4d}  {name:40s} {shapes}")
