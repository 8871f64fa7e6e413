#!/usr/bin/env python3
"""Turn one round's raw measurement directory (gpurun_out/<dir>) into the committed
summaries under profiles/:  python tools/make_profiles.py gpurun_out/r1b r1"""
import csv
import json
import os
import re
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, tag = sys.argv[1], sys.argv[2]
out = os.path.join(REPO, "profiles")
os.makedirs(out, exist_ok=True)

shutil.copyfile(os.path.join(src, "kernel_bench.json"), os.path.join(out, f"{tag}_kernel_bench.json"))
for w in ("gcg", "joint", "pgd", "gemma_joint"):
    for a, b in ((f"kt_{w}_kernel_stats.csv", f"{tag}_bench_{w}_kernel_stats.csv"),
                 (f"kt_{w}_by_grid.txt", f"{tag}_bench_{w}_kernel_by_grid.txt"),
                 (f"bench_{w}_under_rocprof.json", f"{tag}_bench_{w}_under_rocprof.json")):
        if os.path.exists(os.path.join(src, a)):
            shutil.copyfile(os.path.join(src, a), os.path.join(out, b))
for w in ("gcg", "joint", "pgd", "gemma_joint"):  # the batch-1 gradient pass alone (tools/grad_pass_profile.py)
    for a, b in ((f"gp_{w}_by_grid.txt", f"{tag}_gradient_pass_{w}_by_grid.txt"), (f"gp_{w}.txt", None)):
        pa = os.path.join(src, a)
        if os.path.exists(pa) and b:
            shutil.copyfile(pa, os.path.join(out, b))
            note = os.path.join(src, f"gp_{w}.txt")
            if os.path.exists(note):
                last = [l for l in open(note).read().splitlines() if "gradient pass" in l][-1:]
                with open(os.path.join(out, b), "a") as f:
                    f.write("# " + (last[0] if last else "") + "\n")
for extra in ("bench_driver_detail.json", "bench_gcg_under_rocprof_detail.json", "bench_joint_under_rocprof_detail.json",
              "bench_pgd_under_rocprof_detail.json", "bench_gemma_joint_under_rocprof_detail.json", "bench_em8_detail.json",
              "bench_joint_em8_detail.json", "bench_opt125m_detail.json", "bench_em2.json", "bench_em4.json", "bench_em8.json", "bench_joint_em2.json", "bench_joint_em4.json", "bench_joint_em8.json", "bench_opt125m.json", "bench_pgd.json", "bench_pgd_gcg.json", "bench_joint.json",
              "bench_gemma_joint.json", "bench_default.json", "bench_driver.json", "gemm_bench.json", "gemm_bench.txt", "gemm_mid_bench.json", "gemm_mid_bench.txt", "gemm_chain.json", "gemm_chain.txt", "kernel_bench.txt"):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copyfile(os.path.join(src, extra), os.path.join(out, f"{tag}_{extra}"))
for c in ("fetch", "write"):
    rows = list(csv.DictReader(open(os.path.join(src, f"{c}_counter_collection.csv"))))
    keep = [r for r in rows if ("anonymous namespace" in r["Kernel_Name"] and "at::native" not in r["Kernel_Name"])
            or "Cijk_" in r["Kernel_Name"]]
    if len(keep) > 1500:                       # keep the committed raw rows small: every case, first 8 dispatches
        seen, small = {}, []
        for r in keep:
            k = (r["Kernel_Name"], r.get("Grid_Size", ""))
            seen[k] = seen.get(k, 0) + 1
            if seen[k] <= 8:
                small.append(r)
        keep = small
    name = f"{tag}_kernel_bench_pmc_{'FETCH_SIZE' if c == 'fetch' else 'WRITE_SIZE'}.csv"
    with open(os.path.join(out, name), "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=rows[0].keys())
        w.writeheader()
        w.writerows(keep)
folded_path = os.path.join(src, "pmc_folded.json")
subprocess.run([sys.executable, os.path.join(REPO, "tools", "pmc_traffic.py"), os.path.join(src, "fetch_counter_collection.csv"),
                os.path.join(src, "write_counter_collection.csv"), folded_path], check=True, stdout=subprocess.DEVNULL)
folded_all = json.load(open(folded_path))
folded, calibration = folded_all["kernels"], folded_all.get("calibration")
kb = json.load(open(os.path.join(src, "kernel_bench.json")))
case_index = {name: i for i, name in enumerate(kb)}          # kernel_bench.py's case order = the order of its markers in the PMC passes

# kernel_bench case -> (bench kernel name, PMC key = "<symbol><template>/threads<total threads>")
CASES = [
    ("rmsnorm/c3r_17152x4096", "rmsnorm", "rmsnorm_kernel<1, 2, false>/threads4390912", "17152 x 4096 bf16 (C3 ragged candidate forward)"),
    ("swiglu/c3r_17152x11008", "swiglu", "swiglu_kernel<1, 0, false>/threads5900288", "17152 x 11008 bf16 (C3 ragged candidate forward)"),
    ("rope/c3r_N17152_H32_Dh128", "rope", "rope_kernel<1>/threads4390912", "17152 rows, H=32 Dh=128 bf16 (C3 ragged)"),
    ("ragged_attn/c3r_sw512_P21_L44_H32_Dh128", "ragged_attn", "ragged_attn_kernel<1, 3, 128>/threads92160", "17152 rows, 480 candidates x 32 heads, 21 prefix keys (C3 ragged)"),
    ("ragged_attn/gemma_B164_L303_P20_H8_Hk4_Dh256", "ragged_attn", "ragged_attn_long_kernel<1, 256, 2, 1, 4>/threads131072", "Gemma-3 joint blocks: 164 x 303 tokens, 8 heads on 4, 256 wide, 20 prefix keys (persistent long-block kernel)"),
    ("ragged_attn/c4_B512_L45_P0_H32_Dh128", "ragged_attn", "ragged_attn_kernel<1, 3, 128>/threads98304", "C4 padded blocks 512 x 45, no prefix"),
    ("prefix_attn/c4_N17152_P599_H32_Dh128", "prefix_attn", "prefix_attn32_kernel<1, 4>/threads1097728", "joint scoring: 17152 rows x 599 shared prefix keys x 32 heads x 128 (168 GFLOP; the 32x32x16 kernel)"),
    ("attn_merge/c3r_N17152_B481_L44", "attn_merge", "attn_merge_kernel<1>/threads4390912", "17152 rows vs padded 481 x 44, H=32 Dh=128 bf16 (library-attention route)"),
    ("gather_rows/c3r_21164_of_17152x4096", "gather_rows", "gather_rows_kernel/threads5417984", "21164 padded slots from 17152 rows of 8 KiB (library-attention route)"),
    ("ce_rows/llava_B512_T20_V32064", "ce_rows", "ce_rows_kernel<1, true, false>/threads2621440", "B=512 T=20 V=32064 bf16 (C3/C4 scoring, one chunk)"),
    ("ce_rows/gemma_B64_T20_V262208", "ce_rows", "ce_rows_kernel<1, true, false>/threads327680", "B=64 T=20 V=262208 bf16 (Gemma-3 scoring)"),
    ("ce_dlogits/llava_T20_V32064", "ce_dlogits", "ce_dlogits_kernel<1, true>/threads5120", "B=1 T=20 V=32064 bf16 (gradient pass)"),
    ("splice/c3_tail_B512_S44_D4096", "splice", "splice_kernel<1>/threads720896", "C3 tail: B=512, 19 gathered + 25 shared rows, D=4096 bf16"),
    ("splice/c3_full_B512_S65_D4096", "splice", "splice_kernel<1>/threads532480", "C3 full: B=512 S=65 D=4096 bf16"),
    ("splice/c4_full_B512_S643_D4096", "splice", "splice_kernel<1>/threads5267456", "C4 full: B=512 S=643 D=4096 bf16"),
    ("linf/gemma_3x896x896", "linf", "linf_step_vec4/threads524288", "Gemma image 3x896x896 fp32"),
    ("linf/llava_3x336x336", "linf", "linf_step_vec4/threads84736", "LLaVA image 3x336x336 fp32"),
    ("mask_topk/llava_19x32064_bf16", "mask_topk", "topk_slice_kernel<1, true>/threads38912", "19 x 32064 bf16, stage 1 (slice select)"),
    ("mask_topk/gemma_19x262208_bf16", "mask_topk", "topk_slice_kernel<1, true>/threads316160", "19 x 262208 bf16, stage 1 (slice select)"),
    ("rmsnorm/c3_22528x4096", "rmsnorm", "rmsnorm_kernel<1, 2, false>/threads5767168", "22528 x 4096 bf16 (C3 candidate forward)"),
    ("swiglu/c3_22528x11008", "swiglu", "swiglu_kernel<1, 0, false>/threads7749632", "22528 x 11008 bf16 (C3 candidate forward)"),
    ("rope/c3_B512_L44_H32_Dh128", "rope", "rope_kernel<1>/threads5767168", "B=512 L=44 H=32 Dh=128 bf16"),
    ("attn_merge/c4_B512_L45_H32_Dh128", "attn_merge", "attn_merge_kernel<1>/threads5898240", "B=512 L=45 H=32 Dh=128 bf16 (C4)"),
    ("rope/c3r_qk_N17152_H32_Dh128", "rope", "rope2_kernel<1>/threads4390912", "q and k of 17152 rows in one launch, H=32+32 Dh=128 bf16, strided views of the fused q/k/v output"),
    ("rope/gemma_qknorm_B160_L303_H8_Hk4_Dh256", "rope", "qknorm_rope2_kernel<1>/threads12410880", "Gemma-3 chunk of 160 candidates x 303 tokens: per-head q/k norm + rotary of 8 + 4 heads x 256 in one launch"),
    ("add_rmsnorm/c3r_17152x4096", "add_rmsnorm", "add_rmsnorm_kernel<1, 2, false, false>/threads4390912", "17152 x 4096 bf16: residual add + RMSNorm in one pass (C3 ragged candidate forward)"),
    ("splice/c3r_rows_17152_D4096", "splice", "splice_rows_kernel<1>/threads4390912", "C3 ragged row list: 17152 rows of 8 KiB straight from the segments and the table"),
    ("gemm_nt/gate_up_dX_65x4096x22016", "gemm_nt", "gemm_nt_kernel<1, 6, 2, 4, true>/threads65536", "bma_gemm_nt 65 x 4096 x 22016 bf16 (input gradient of the fused gate/up product, 8-way split-K)"),
    ("gemm_nt/qkv_dX_65x4096x12288", "gemm_nt", "gemm_nt_kernel<1, 6, 2, 4, true>/threads65536", "bma_gemm_nt 65 x 4096 x 12288 bf16 (input gradient of the fused q/k/v product, 8-way split-K)"),
    ("gemm_nt/gate_up_65x22016x4096", "gemm_nt", "gemm_nt_kernel", "bma_gemm_nt 65 x 22016 x 4096 bf16 (fused gate/up product: 172 slabs, unsplit)"),
    ("gemm_nt/down_65x4096x11008", "gemm_nt", "gemm_nt_kernel", "bma_gemm_nt 65 x 4096 x 11008 bf16 (down_proj, 8-way split-K)"),
    ("gemm_mid/down_644x4096x11008", "gemm_mid", "gemm_mid_kernel", "bma_gemm_mid 644 x 4096 x 11008 bf16 (down_proj at 644 rows; the partials' second launch not included)"),
    ("gemm_mid/gate_up_dX_644x4096x22016", "gemm_mid", "gemm_mid_kernel<1, 7, 4, 3>/threads122880", "bma_gemm_mid 644 x 4096 x 22016 bf16 (input gradient of the fused gate/up product at 644 rows: 48 tiles of 224 x 256, K split 5 ways; the partials' second launch not included)"),
    ("gemm_mid/gate_up_644x22016x4096", "gemm_mid", "gemm_mid_kernel<1, 7, 4, 3>/threads142848", "bma_gemm_mid 644 x 22016 x 4096 bf16 (fused gate/up product at 644 rows: 255 whole tiles + the last column split 8 ways)"),
    ("causal_attn/fwd_L643_H32", "causal_attn", "causal_fwd_kernel<1, 128>/threads180224", "causal attention forward, 643 tokens x 32 heads x 128 at batch 1 (3.4 GFLOP: latency-bound)"),
    # library GEMMs (keys are matched by symbol prefix: the kernel name depends on the selection table)
    ("gemm/gate_up_17152x22016x4096", "gemm_gate_up_proj", "Cijk", "fused gate/up product of the C3 ragged candidate forward, 17152 x 22016 x 4096 bf16"),
    ("gemm/down_17152x4096x11008", "gemm_down_proj", "Cijk", "down_proj, 17152 x 4096 x 11008 bf16"),
    ("gemm/qkv_17152x12288x4096", "gemm_qkv_proj", "Cijk", "fused q/k/v product, 17152 x 12288 x 4096 bf16"),
]
entries = []
for case, kernel, key, shape in CASES:
    if case not in kb:
        continue
    mine = f"case{case_index[case]}/"                        # only rows between THIS case's marker and the next one
    if key == "Cijk":
        # library GEMMs share one symbol family: the launch of THIS product is the one whose grid is the product's tile
        # count (macro-tile MTaxb from the kernel name, 256 threads per workgroup) -- round 2 took the launch whose
        # traffic was nearest to the operand bytes, which gave all three products the smallest one's row
        Mg, Ng = (int(v) for v in case.rsplit("_", 1)[1].split("x")[:2])
        cands = []
        for k in folded:
            if not k.startswith(mine):
                continue
            sym = k.split("/")[1]
            mt = re.search(r"_MT(\d+)x(\d+)x", sym)
            th = re.search(r"/threads(\d+)$", k)
            if "Cijk_" in sym and mt and th:
                tiles = -(-Mg // int(mt.group(1))) * -(-Ng // int(mt.group(2)))
                if int(th.group(1)) in (tiles * 256, tiles * 512, tiles * 128):
                    cands.append(k)
    else:
        cands = [k for k in folded if k.startswith(mine) and k.split("/", 1)[1] == key] or \
            [k for k in folded if k.startswith(mine) and k.split("/")[1].split("<")[0] == key.split("/")[0].split("<")[0]]
    # several shapes can share a symbol: take the launch whose traffic is closest to the algorithmic bytes
    algo = kb[case]["algorithmic_MB"] * 1e6
    if not cands:
        continue
    best = min(cands, key=lambda k: abs(folded[k]["hbm_bytes_per_launch"] - algo))
    v = folded[best]
    ratio = v["hbm_bytes_per_launch"] / algo
    # a kernel cannot move fewer bytes than its operands hold: such a row is a measurement fault (mixed cases, an operand
    # resident in the Infinity Cache between launches, an access width the x2 does not hold for), not evidence
    valid = ratio >= 0.97
    entries.append(dict(kernel=kernel, shape=shape, pmc_key=best, valid=valid, algorithmic_bytes=algo, fetch_bytes_uncorrected=v["fetch_bytes_per_launch_corrected"] / 2 if v["fetch_bytes_per_launch_corrected"] else None,
                        hbm_bytes_per_launch=v["hbm_bytes_per_launch"], fetch_bytes_corrected=v["fetch_bytes_per_launch_corrected"],
                        write_bytes=v["write_bytes_per_launch"], ratio_to_algorithmic=v["hbm_bytes_per_launch"] / algo,
                        avg_us=kb[case]["avg_us"], achieved_GBps=kb[case]["achieved_GBps"],
                        achieved_TFLOPs=kb[case].get("achieved_TFLOPs")))
try:
    commit = subprocess.run(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
except Exception:
    commit = None
json.dump(dict(source="rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes with --kernel-trace only, of "
                      "tools/kernel_bench.py --iters 5 on MI355X; raw rows in *_kernel_bench_pmc_*.csv",
               generated_at_commit=commit,
               corrections="counters are KiB; FETCH_SIZE x2 (gfx950 counts 128-B requests at 64 B, MI355X_MICROARCH.md HBM); WRITE_SIZE x1.  "
                           "The counters sit on the L2's memory side: Infinity-Cache hits are counted, so for a library GEMM "
                           "(tiles re-read operand panels; 16-byte-per-lane buffer loads assumed for the x2) the figure is L2-miss "
                           "traffic, an upper bound on HBM bytes; without the x2 it would be fetch/2 + write",
               calibration=calibration,
               entries=entries), open(os.path.join(out, f"{tag}_pmc_traffic.json"), "w"), indent=1)
print("calibration:", calibration)
for e in entries:
    print(f"{e['kernel']:11s} {e['shape'][:52]:52s} {e['avg_us']:8.1f} us {e['achieved_GBps']:7.0f} GB/s  traffic/algorithmic = {e['ratio_to_algorithmic']:.3f}"
          + ("" if e["valid"] else "   INVALID (< 1: not evidence)"))
bad = [e for e in entries if not e["valid"]]
if bad:
    print(f"WARNING: {len(bad)} entries report fewer bytes than their operands hold", file=sys.stderr)
