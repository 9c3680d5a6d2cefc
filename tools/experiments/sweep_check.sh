#!/bin/bash
# round 6: the block sweep against LAPACK at ragged sizes, then timings against the three-phase form
cd "$(dirname "$0")/../.."
for n in 1024 1088 1472 2112 4032; do python tools/time_inv.py --n $n --reps 2 --check --tag sweep; done
python tools/time_inv.py --n 10048 --reps 3 --check --tag sweep
DCA_SWEEP=0 python tools/time_inv.py --n 10048 --reps 3 --tag three-phase
DCA_SWEEP=0 python tools/time_inv.py --n 4032 --reps 3 --tag three-phase
python tools/time_inv.py --n 4032 --reps 4 --tag sweep
DCA_CHOLINV_TRACE=1 python tools/time_inv.py --n 10048 --reps 2 --tag trace 2> gpurun_out/sweep_trace_D.txt
DCA_CHOLINV_TRACE=1 python tools/time_inv.py --n 4032 --reps 2 --tag trace 2> gpurun_out/sweep_trace_B.txt
