#!/bin/bash
# Everything profiles/ holds for one round, in one gpurun call (run from the repository root on the GPU box):
#   DCA_COMMIT=<git rev-parse --short HEAD> bash tools/profile_round.sh <tag>      -> gpurun_out/<tag>/...
# (.git does not travel to the GPU box: the commit the traffic figures belong to comes in through DCA_COMMIT)
# bench.py first (fresh process, cold box), then the rocprofv3 passes: kernel trace + stats, HBM traffic
# (FETCH_SIZE / WRITE_SIZE in separate passes), SQ counters in their own passes (no tracing alongside --pmc).
set -u
tag=${1:-final}
root=$(pwd)
out=$root/gpurun_out/$tag
mkdir -p "$out"
python bench.py > "$out/bench_D.json" 2> "$out/bench_D.err"
python bench.py --workload C > "$out/bench_C.json" 2> "$out/bench_C.err"
python bench.py --workload E > "$out/bench_E.json" 2> "$out/bench_E.err"
cd /tmp && export TMPDIR=/tmp
B="python $root/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-e2e --no-rna --no-modes"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- $B > "$out/stats_run.log" 2>&1
cp "$(ls -t /tmp/p_stats/*/*kernel_stats.csv | head -1)" "$out/D_kernel_stats.csv"
rocprofv3 --kernel-trace --output-format csv -d /tmp/p_inv -- python "$root/tools/time_mf.py" --reps 2 > /dev/null 2>&1
python "$root/tools/experiments/inverse_trace.py" "$(ls -t /tmp/p_inv/*/*kernel_trace.csv | head -1)" > "$out/inverse_trace.txt" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- $B > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- $B > /dev/null 2>&1
python "$root/tools/pmc_summary.py" "$(ls -t /tmp/p_fetch/*/*counter_collection.csv | head -1)" "$(ls -t /tmp/p_write/*/*counter_collection.csv | head -1)" \
    "$out/D_hbm_traffic.csv" "$out/traffic.json" D
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_ANY" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d /tmp/p_sq_e$i -- python "$root/tools/time_eval.py" --reps 3 > /dev/null 2>&1
    rocprofv3 --pmc $set --output-format csv -d /tmp/p_sq_m$i -- python "$root/tools/time_mf.py" --reps 1 > /dev/null 2>&1
done
python "$root/tools/sq_summary.py" "$out/D_sq_counters.csv" $(for d in /tmp/p_sq_*; do ls -t $d/*/*counter_collection.csv | head -1; done)
ls -la "$out"
