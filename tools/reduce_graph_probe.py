#!/usr/bin/env python3
"""Does ATen's global reduce (semaphores reset by a captured hipMemsetAsync) survive hipGraph replays when the graph
reuses the semaphore block for something else afterwards?"""
import torch
dev = torch.device("cuda", 0)
torch.manual_seed(0)


def fn(x):
    xt = x.transpose(1, 2)                       # (1,256,1152), reduced dim strided by 256
    f = xt.float()
    m = f.pow(2).mean(-1, keepdim=True)          # ATen reduce over the strided dim: global reduce (buffer + semaphores)
    junk = torch.full((64,), 12345, dtype=torch.int32, device=dev)      # small allocations that may land on the freed semaphores
    junk2 = torch.full((64,), -1, dtype=torch.int32, device=dev) + junk
    return m * 1.0, junk2


x = torch.randn(1, 1152, 256, device=dev, dtype=torch.bfloat16)
static = x.clone()
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side), torch.no_grad():
    fn(static)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.no_grad(), torch.cuda.graph(g, capture_error_mode="thread_local"):
    out = fn(static)
for i in range(6):
    xi = torch.randn_like(x) * (i + 1)
    static.copy_(xi)
    g.replay()
    torch.cuda.synchronize()
    want = fn(xi)[0]
    got = out[0]
    bad = ~torch.isfinite(got)
    print(f"replay {i}: bad {int(bad.sum())} max diff {float((got - want).nan_to_num(1e9).abs().max()):.3g}", flush=True)
