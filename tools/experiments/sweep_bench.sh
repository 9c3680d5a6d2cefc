#!/bin/bash
cd "$(dirname "$0")/../.."
run() { tag=$1; n=$2; shift 2; env "$@" python tools/time_inv.py --n $n --reps 3 --tag "$tag" | tail -1; }
for n in 1024 1536 2048 3072 4032 5056 6016 8000 10048 12032; do
  run "sweep256" $n DCA_SWEEP_PANEL=256; run "sweep512" $n DCA_SWEEP_PANEL=512; run "three-phase" $n DCA_SWEEP=0
done
