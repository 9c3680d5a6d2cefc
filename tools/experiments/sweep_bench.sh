#!/bin/bash
cd "$(dirname "$0")"
echo "== full kernel"; ./sweep_bench_a0 10048 512; ./sweep_bench_a0 4032 256
echo "== no tile load"; ./sweep_bench_a1 10048 512 | grep REST
cd ../..
python tools/experiments/sweep_diag.py 1088
DCA_SWEEP_PANEL=128 python tools/experiments/sweep_diag.py 1088 | head -3
for n in 1024 1472 4032; do python tools/time_inv.py --n $n --reps 2 --check --tag sweep; done
python tools/time_inv.py --n 10048 --reps 3 --check --tag sweep
DCA_SWEEP_PER_CU=1 DCA_SWEEP_CAP=248 DCA_SWEEP_STAGES=4 python tools/time_inv.py --n 10048 --reps 3 --tag sweep-1percu
DCA_CHOLINV_TRACE=1 python tools/time_inv.py --n 10048 --reps 2 --tag trace 2> gpurun_out/sweep_trace_D.txt
DCA_CHOLINV_TRACE=1 python tools/time_inv.py --n 4032 --reps 2 --tag trace 2> gpurun_out/sweep_trace_B.txt
