#!/bin/bash
cd "$(dirname "$0")/../.."
run() { tag=$1; n=$2; shift 2; env "$@" python tools/time_inv.py --n $n --reps 4 --tag "$tag" | tail -1; }
for pw in 128 256 384 512; do for cap in 128 256 384; do run "panel$pw-cap$cap" 4032 DCA_SWEEP_PANEL=$pw DCA_SWEEP_CAP=$cap; done; done
for pc in 16 32 128; do run "panel256-prio$pc" 4032 DCA_SWEEP_PRIO_CAP=$pc; done
run "fused-walk" 4032 DCA_SWEEP=0
DCA_CHOLINV_TRACE=1 python tools/time_inv.py --n 4032 --reps 2 --tag trace 2> gpurun_out/sweep_trace_B.txt | tail -1
python tools/experiments/sweep_trace_summary.py gpurun_out/sweep_trace_B.txt | head -10
