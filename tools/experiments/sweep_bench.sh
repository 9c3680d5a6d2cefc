#!/bin/bash
cd "$(dirname "$0")/../.."
python tools/experiments/mf_inv_time.py default
for k in 1 2; do DCA_SWEEP_WKERNEL=$k python tools/experiments/mf_inv_time.py wkernel$k; done
for f in 128 256; do DCA_SWEEP_FIRST=$f python tools/experiments/mf_inv_time.py first$f; done
python tools/experiments/mf_inv_time.py default-again
