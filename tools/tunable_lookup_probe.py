#!/usr/bin/env python3
"""Is the shipped GEMM selection table HONOURED for the scoring forward's products?  (Round 5: TunableOp's own timing loop
measures the table's pick for down_proj 17152 x 4096 x 11008 at ~900 us, a plain loop under the table at ~1060 = the library
default.)  Times each product three ways in one process: library default (TunableOp off), the shipped table (lookup only), and
TunableOp tuning the shape right here (its pick used by the very next calls)."""
import os
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import torch  # noqa: E402
import torch.cuda.tunable as tun  # noqa: E402

dev = torch.device("cuda", 0)
SHAPES = [("down_proj", 17152, 4096, 11008), ("o_proj", 17152, 4096, 4096), ("qkv", 17152, 12288, 4096), ("gate_up", 17152, 22016, 4096)]


def bench(x, w, n=100):
    for _ in range(5):
        torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        torch.nn.functional.linear(x, w)
    e1.record()
    torch.cuda.synchronize()
    return 1e3 * e0.elapsed_time(e1) / n


ops = {}
for name, M, N, K in SHAPES:
    ops[name] = (torch.randn((1, M, K), device=dev).to(torch.bfloat16), (torch.randn((N, K), device=dev) * 0.02).to(torch.bfloat16))
res = {n: {} for n in ops}
for n, (x, w) in ops.items():
    res[n]["default"] = bench(x, w)
MODE = sys.argv[1] if len(sys.argv) > 1 else "table"
if MODE == "table":
    from bimodalattack_amd import gemm_tuning  # noqa: E402
    ok = gemm_tuning.enable("auto", dev)
    print("table accepted:", ok, "entries:", len(tun.get_results()), "tuning enabled:", tun.tuning_is_enabled(), "enabled:", tun.is_enabled(), flush=True)
    for n, (x, w) in ops.items():
        res[n]["table"] = bench(x, w)
    hits = [r for r in tun.get_results() if "17152" in r[1]]
    print("table rows for 17152:", [(r[1], r[2], round(float(r[3]), 3)) for r in hits][:8], flush=True)
    for n, (_, M, N, K) in zip(ops, SHAPES):
        r = res[n]
        fl = 2.0 * M * N * K
        print(f"{n:9s} default {r['default']:7.1f} us ({fl / r['default'] / 2.5e9:.3f})   table {r['table']:7.1f} us ({fl / r['table'] / 2.5e9:.3f})", flush=True)
    sys.exit(0)
# MODE "tune": no table -- TunableOp tunes these shapes in THIS process and its picks serve the very next calls
tun.enable(True)
for n in res:
    res[n]["table"] = float("nan")
tun.tuning_enable(True)
tun.set_max_tuning_duration(60)
tun.set_max_tuning_iterations(100)
try:
    tun.set_rotating_buffer_size(512)
except Exception as e:
    print("rotating buffer:", e)
for n, (x, w) in ops.items():
    t0 = time.perf_counter()
    torch.nn.functional.linear(x, w)          # tunes on first sight... only if the shape is NOT in the table already
    torch.cuda.synchronize()
    res[n]["after_tuning_call_s"] = time.perf_counter() - t0
tun.tuning_enable(False)
for n, (x, w) in ops.items():
    res[n]["retuned"] = bench(x, w)
for n, (_, M, N, K) in zip(ops, SHAPES):
    r = res[n]
    fl = 2.0 * M * N * K
    print(f"{n:9s} default {r['default']:7.1f} us ({fl / r['default'] / 2.5e9:.3f})   table {r['table']:7.1f} us ({fl / r['table'] / 2.5e9:.3f})   "
          f"after in-process tuning {r['retuned']:7.1f} us ({fl / r['retuned'] / 2.5e9:.3f})  [tuning call took {r['after_tuning_call_s']:.1f} s]", flush=True)
