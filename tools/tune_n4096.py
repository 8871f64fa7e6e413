#!/usr/bin/env python3
"""Every rocBLAS / hipBLASLt solution PyTorch's TunableOp can offer for the two N = 4096 products of the scoring forward
(down_proj 17152 x 4096 x 11008, o_proj 17152 x 4096 x 4096), timed one by one (is a 256 x 128 / stream-K
solution left on the table at 1072 tiles = 4.19 rounds of 256 CUs?).

    python tools/tune_n4096.py > profiles/r5_gemm_n4096.txt

TunableOp in tuning mode times EVERY solution the two libraries list for the shape (it is where tuning/gfx950.csv's winners come
from); with PYTORCH_TUNABLEOP_VERBOSE=3 it prints each candidate's time.  This script runs that for the two shapes from an
EMPTY results file, parses the per-candidate lines, and prints them sorted -- then launches the winner and the runners-up
under the in-process kernel timer to name the kernels (macro-tile in the symbol) behind the indices.
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [("down_proj", 17152, 4096, 11008), ("o_proj", 17152, 4096, 4096)]

CHILD = r"""
import torch, sys
M, N, K = (int(v) for v in sys.argv[1:4])
x = torch.randn((1, M, K), device="cuda").to(torch.bfloat16)
w = (torch.randn((N, K), device="cuda") * 0.02).to(torch.bfloat16)
torch.nn.functional.linear(x, w)
torch.cuda.synchronize()
"""


def main():
    for name, M, N, K in SHAPES:
        tmp = tempfile.mkdtemp(prefix="bma_tune_")
        env = dict(os.environ, PYTORCH_TUNABLEOP_ENABLED="1", PYTORCH_TUNABLEOP_TUNING="1", PYTORCH_TUNABLEOP_VERBOSE="3",
                   PYTORCH_TUNABLEOP_FILENAME=os.path.join(tmp, "t.csv"), PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS="60",
                   PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS="10", PYTORCH_TUNABLEOP_ROTATING_BUFFER_SIZE="512")
        r = subprocess.run([sys.executable, "-c", CHILD, str(M), str(N), str(K)], env=env, capture_output=True, text=True, timeout=900)
        text = r.stdout + r.stderr
        cands, skipped = [], 0
        for line in text.splitlines():
            m = re.search(r"found (?:better|slower) instance id=(\d+)\. ([0-9.eE+-]+)ms\. (\S+)", line)
            if m:
                cands.append((float(m.group(2)), m.group(3), line.strip()[:160]))
            elif "skip slow instance" in line or "unsupported" in line:
                skipped += 1
        raw = [l for l in text.splitlines() if "instance id=" in l or "Gemm_" in l]
        with open(os.path.join(REPO, "gpurun_out", f"r5_gemm_n4096_raw_{name}.txt"), "w") as fh:
            fh.write("\n".join(raw[:400]) + "\n")
        flops = 2.0 * M * N * K
        print(f"== {name}: {M} x {N} x {K} bf16 ({flops / 1e12:.3f} TFLOP); {len(cands)} candidates timed in full, {skipped} skipped early "
              f"(unsupported, or slower than the best so far after a few iterations); rc={r.returncode}")
        seen = {}
        for t, nm, line in cands:
            if t > 0:
                seen[nm] = min(t, seen.get(nm, 1e9))
        ranked = sorted(seen.items(), key=lambda kv: kv[1])
        for nm, t in ranked[:30]:
            print(f"   {nm:34s} {1e3 * t:9.1f} us   {flops / (t * 1e-3) / 1e12:7.0f} TFLOP/s = {flops / (t * 1e-3) / 2.5e15:5.3f} of 2.5 PF")
        if len(ranked) > 30:
            print(f"   ... {len(ranked) - 30} slower ones; slowest {1e3 * ranked[-1][1]:.0f} us")
        if not ranked:
            print("   (no candidate lines recognised; tail of the tuner's output follows)")
            print("\n".join("   | " + l for l in text.splitlines()[-40:]))
        try:
            print("   results file:", open(os.path.join(tmp, "t0.csv")).read().strip().splitlines()[-1])
        except OSError:
            pass


if __name__ == "__main__":
    main()
