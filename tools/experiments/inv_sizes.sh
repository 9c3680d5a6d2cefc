#!/bin/bash
# blocked (look-ahead) inverse against the fused walk over matrix sizes
cd "$(dirname "$0")/../.."
out=gpurun_out/inv_sizes.txt
: > $out
for n in 2560 3072 4032 5056 6016 8000 10048; do
  for v in DCA_CHOLINV_BLOCKED=0 DCA_CHOLINV_BLOCKED_MIN=0 "DCA_CHOLINV_BLOCKED_MIN=0 DCA_CHOLINV_PANEL=256"; do
    env $v python tools/time_inv.py --n $n --reps 4 --tag "$v" 2>&1 | grep BEST >> $out
  done
done
cat $out
