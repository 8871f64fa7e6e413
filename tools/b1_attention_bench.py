"""bma_b1_attention forward / backward (65 tokens, 32 heads) and the library's causal attention forward, each as 200
back-to-back launches of one hipGraph.

    python tools/b1_attention_bench.py
"""
import torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bimodalattack_amd import ops
DEV = "cuda:0"
S, H = 65, 32
g = torch.Generator(device=DEV).manual_seed(0)
qkv = torch.randn((S, 3 * H * 128), generator=g, device=DEV).to(torch.bfloat16)
ang = torch.rand((S, 64), generator=g, device=DEV)
cos = torch.cat([ang.cos(), ang.cos()], -1).to(torch.bfloat16)
sin = torch.cat([ang.sin(), ang.sin()], -1).to(torch.bfloat16)
dout = torch.randn((S, H * 128), generator=g, device=DEV).to(torch.bfloat16)
out, lse = ops.b1_attention(qkv, cos, sin, H, 0.088)


def timeit(fn, n=200):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(n):
            fn()
    gr.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    gr.replay()
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


print("fwd us", timeit(lambda: ops.b1_attention(qkv, cos, sin, H, 0.088)))
print("bwd us", timeit(lambda: ops.b1_attention_bwd(qkv, cos, sin, out, lse, dout, H, 0.088)))
q = torch.randn((1, H, S, 128), device=DEV, dtype=torch.bfloat16, requires_grad=True)
k = torch.randn((1, H, S, 128), device=DEV, dtype=torch.bfloat16, requires_grad=True)
v = torch.randn((1, H, S, 128), device=DEV, dtype=torch.bfloat16, requires_grad=True)
print("sdpa fwd us", timeit(lambda: torch.nn.functional.scaled_dot_product_attention(q, k, v, is_causal=True)))
