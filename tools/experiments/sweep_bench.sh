#!/bin/bash
cd "$(dirname "$0")/../.."
for n in 2112 4032; do python tools/time_inv.py --n $n --reps 2 --check --tag sweep | tail -2; done
python tools/time_inv.py --n 10048 --reps 3 --check --tag sweep | tail -2
DCA_CHOLINV_TRACE=1 python tools/experiments/mf_twice.py 2> gpurun_out/sweep_trace_mf.txt | tail -2
python tools/experiments/sweep_trace_summary.py gpurun_out/sweep_trace_mf.txt
for w in D C; do python bench.py --workload $w --no-cpu-baseline --no-e2e --no-rna --no-modes 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$w', d['value'], d['mfdca']['stages_ms'])"; done
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "spd_inverse or mf_" 2>&1 | tail -3
