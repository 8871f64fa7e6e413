#!/usr/bin/env python3
"""Does an RCCL collective survive hipGraph capture and replay on this stack?  One rank on one GPU (world_size 1: the
all-reduce moves nothing, but it goes through ProcessGroupNCCL's capture path and RCCL's enqueue under stream capture --
what the tensor-parallel gradient pass's one-graph mode, EngineOptions.tp_gradient="graph", relies on and what no run of
this repository has met on more than one GPU):

    python3 tools/rccl_graph_probe.py

Prints what happened at each stage; exits 0 only if the replayed graph gave the eager result."""
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29517")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    print("rccl", ".".join(str(v) for v in torch.cuda.nccl.version()), "torch", torch.__version__, flush=True)
    x = torch.arange(1 << 16, device=dev, dtype=torch.float32)
    w = torch.randn((256, 256), device=dev)
    dist.all_reduce(x)                                   # communicator warm-up outside any capture
    torch.cuda.synchronize(dev)

    def body(t):
        y = t @ w[:, :1].expand(256, 256)[:1].t().contiguous().expand(1, 1) if False else t * 2.0
        dist.all_reduce(y)
        z = y + 1.0
        dist.all_reduce(z)
        return z

    want = body(x.clone())
    torch.cuda.synchronize(dev)
    print("eager ok", float(want.sum()), flush=True)
    for mode in ("thread_local", "global", "relaxed"):
        try:
            static_in = x.clone()
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream(dev)
            s.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(s):
                body(static_in)                          # warm-up on the side stream
            torch.cuda.current_stream(dev).wait_stream(s)
            torch.cuda.synchronize(dev)
            with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
                static_out = body(static_in)
            for _ in range(3):
                static_in.copy_(x)
                g.replay()
            torch.cuda.synchronize(dev)
            ok = bool(torch.equal(static_out, want))
            print(f"capture_error_mode={mode}: captured and replayed x3, equal to eager: {ok}", flush=True)
            if not ok:
                sys.exit(2)
        except Exception as e:
            print(f"capture_error_mode={mode}: {type(e).__name__}: {str(e)[:300]}", flush=True)
            torch.cuda.synchronize(dev)
    dist.destroy_process_group()
    print("done", flush=True)


if __name__ == "__main__":
    main()
