#!/bin/bash
# One round's measurements on an MI355X box, into gpurun_out/<dir> (scratch); turn them into the
# committed summaries with  python tools/make_profiles.py gpurun_out/<dir> <tag>.
#   gpurun --timeout 1150 -- 'bash tools/measure_round.sh r3'
# PMC passes are their own runs (kernel-trace only), over the standalone kernel driver.
# A round's measurements no longer fit one 20-minute call: `measure_round.sh r5 a` (driver line, gradient-pass traces,
# kernel driver + PMC passes) and `measure_round.sh r5 b` (rocprofv3 traces of the four workloads, emulated ranks, GEMM benches).
set -u
D=gpurun_out/${1:-round}
PART=${2:-all}
mkdir -p "$D"
export TMPDIR=/tmp
if [ "$PART" != "b" ]; then
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke
# the driver's own command line first: every single-GPU BASELINE config in one JSON line
timeout -k 10 400 python bench.py --gpus 1 --steps 20 --warmup 5 --detail "$D/bench_driver_detail.json" > "$D/bench_driver.json" 2> "$D/bench_driver.err"; echo "driver-rc=$?"
for w in gcg joint pgd gemma_joint; do
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$D" -o gp_$w -- python3 tools/grad_pass_profile.py --workload $w > "$D/gp_$w.txt" 2>&1; tail -1 "$D/gp_$w.txt"
  python3 tools/trace_by_grid.py "$D/gp_${w}_kernel_trace.csv" "$D/gp_${w}_by_grid.txt" 40 --between-markers; rm -f "$D/gp_${w}_kernel_trace.csv"
done
python3 tools/kernel_bench.py --json "$D/kernel_bench.json" > "$D/kernel_bench.txt" 2>&1; echo "kb-rc=$?"
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d "$D" -o fetch -- python3 tools/kernel_bench.py --iters 5 > /dev/null 2>&1; echo "fetch-rc=$?"
timeout -k 10 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d "$D" -o write -- python3 tools/kernel_bench.py --iters 5 > /dev/null 2>&1; echo "write-rc=$?"
fi
if [ "$PART" = "a" ]; then exit 0; fi
for w in gcg joint pgd gemma_joint; do
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$D" -o kt_$w -- python3 bench.py --workload $w --no-cpu-baseline --extra-workloads none --detail "$D/bench_${w}_under_rocprof_detail.json" > "$D/bench_${w}_under_rocprof.json" 2> /dev/null; echo "kt-$w-rc=$?"
  python3 tools/trace_by_grid.py "$D/kt_${w}_kernel_trace.csv" "$D/kt_${w}_by_grid.txt" 70 --between-markers
done
rm -f "$D"/*_kernel_trace.csv
timeout -k 10 300 python bench.py --workload pgd_gcg --no-cpu-baseline --extra-workloads none --detail "$D/bench_pgd_gcg_detail.json" > "$D/bench_pgd_gcg.json" 2> /dev/null; echo "pgd_gcg-rc=$?"
for n in 2 4 8; do
  BMA_EMULATE_WORLD=$n timeout -k 10 300 python bench.py --no-cpu-baseline --detail "$D/bench_em${n}_detail.json" > "$D/bench_em$n.json" 2> /dev/null; echo "em$n-rc=$?"
done
for n in 2 4 8; do
  BMA_EMULATE_WORLD=$n timeout -k 10 300 python bench.py --workload joint --no-cpu-baseline --detail "$D/bench_joint_em${n}_detail.json" > "$D/bench_joint_em$n.json" 2> /dev/null; echo "joint-em$n-rc=$?"
done
timeout -k 10 300 python bench.py --workload opt125m --detail "$D/bench_opt125m_detail.json" > "$D/bench_opt125m.json" 2> "$D/bench_opt125m.err"; echo "opt125m-rc=$?"
timeout -k 10 200 python tools/gemm_bench.py --rows 65,44 --json "$D/gemm_bench.json" 2>&1 | grep -v amdgpu.ids > "$D/gemm_bench.txt"; echo "gemm-bench-rc=$?"
timeout -k 10 200 python tools/gemm_bench.py --chain --rows 65,44 --json "$D/gemm_chain.json" 2>&1 | grep -v amdgpu.ids > "$D/gemm_chain.txt"; echo "gemm-chain-rc=$?"
timeout -k 10 200 python tools/gemm_bench.py --mid --rows 644,599 --json "$D/gemm_mid_bench.json" 2>&1 | grep -v amdgpu.ids > "$D/gemm_mid_bench.txt"; echo "gemm-mid-bench-rc=$?"
