"""bma_prefix_attention: both kernels (bma_prefix_attention_set_plan 1 / 4 / 8) against float64 attention on six shapes, then
medians of 10 rounds x 32 launches back to back at BASELINE configs[3]'s size (profiles/r6_prefix_attn32.txt).

    PYTHONPATH=. python tools/prefix_attn_probe.py            # PA_TIMING_ONLY=1: skip the parity part (ablation builds)
"""
import sys, time, torch, numpy as np
from bimodalattack_amd import ops, native
lib = native.lib
DEV = "cuda:0"
def ref(q, pk, pv, scale, rows):
    H, Hk = q.shape[1], pk.shape[1]
    rep = H // Hk
    s_ = (q[0][:, rows].double() @ pk[0].double().repeat_interleave(rep, 0).transpose(-1, -2)) * scale
    want = (torch.softmax(s_, -1) @ pv[0].double().repeat_interleave(rep, 0)).transpose(0, 1)
    return want, torch.logsumexp(s_, -1)
def row_scale_err(o, want):
    return float(((o.double() - want).abs().amax(-1) / want.abs().amax(-1)).max())
shapes = [(300, 599, 8, 8), (129, 70, 8, 4), (128, 65, 3, 3), (1000, 640, 16, 16), (257, 129, 32, 32), (17152, 599, 32, 32)]
import os
ok = True
for dt in (() if os.environ.get('PA_TIMING_ONLY') else (torch.bfloat16, torch.float16)):
  for (N, P, H, Hk) in shapes:
    g = torch.Generator(device=DEV).manual_seed(N + P)
    q = torch.randn((1, N, H, 128), generator=g, device=DEV).to(dt).transpose(1, 2)
    pk, pv = (torch.randn((1, P, Hk, 128), generator=g, device=DEV).to(dt).transpose(1, 2) for _ in range(2))
    # asymmetric data: distinct scale per key so a permuted k order shows
    pv = (pv.float() * torch.linspace(0.5, 2.0, P, device=DEV)[None, None, :, None]).to(dt)
    scale = 128 ** -0.5
    rows = torch.randperm(N, generator=torch.Generator().manual_seed(3))[:256].to(DEV)
    want, wl = ref(q, pk, pv, scale, rows)
    for plan in (1, 4, 8):
        lib.bma_prefix_attention_set_plan(plan)
        o, lse = ops.prefix_attention(q, pk, pv, scale)
        torch.cuda.synchronize()
        e = row_scale_err(o[rows], want)
        le = float((lse[:, rows].double() - wl).abs().max())
        fin = bool(torch.isfinite(o.float()).all())
        good = e <= 3 * 2.0 ** -8 * (1 if dt == torch.bfloat16 else 0.125) and le < 2e-3 and fin
        ok &= good
        print(f"{str(dt)[6:]:9s} N{N} P{P} H{H}/{Hk} plan {plan}: row-scale err {e:.2e} lse err {le:.2e} finite {fin} {'ok' if good else 'BAD'}", flush=True)
lib.bma_prefix_attention_set_plan(0)
if not ok:
    print("PARITY FAILED"); sys.exit(1)
# timing at the joint shape
N, P, H = 17152, 599, 32
g = torch.Generator(device=DEV).manual_seed(11)
q = torch.randn((1, N, H, 128), generator=g, device=DEV).to(torch.bfloat16).transpose(1, 2)
pk, pv = (torch.randn((1, P, H, 128), generator=g, device=DEV).to(torch.bfloat16).transpose(1, 2) for _ in range(2))
import statistics
plans = (4,) if os.environ.get('PA_TIMING_ONLY') else (1, 4, 8)
res = {p_: [] for p_ in plans}
for rnd in range(12):
  for plan in plans:
    lib.bma_prefix_attention_set_plan(plan)
    for _ in range(3): ops.prefix_attention(q, pk, pv, 128 ** -0.5)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(32): ops.prefix_attention(q, pk, pv, 128 ** -0.5)
    b.record(); torch.cuda.synchronize()
    res[plan].append(a.elapsed_time(b) / 32 * 1e3)
for plan in plans:
    r = res[plan][2:]
    us = statistics.median(r)
    print(f"plan {plan}: median {us:.1f} us (min {min(r):.1f}, max {max(r):.1f}) {4*N*P*128*H/us/1e6:.0f} TFLOP/s  {4*N*P*128*H/us/1e6/2500:.3f} of peak", flush=True)
lib.bma_prefix_attention_set_plan(0)
