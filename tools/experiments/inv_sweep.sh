#!/bin/bash
# round 5: the blocked (look-ahead) inverse against the fused walk; one process per variant
cd "$(dirname "$0")/../.."
out=gpurun_out/inv_sweep.txt
: > $out
run() { echo "== $*" >> $out; env "$@" python tools/time_inv.py --n ${N:-10048} --reps 3 --tag "$*" 2>&1 | grep BEST >> $out; }
DCA_CHOLINV_BLOCKED_MIN=0 DCA_CHOLINV_PANEL=128 DCA_CHOLINV_SPLITK_MIN=128 python -m pytest tests/test_gpu_parity.py -q -x -k "spd_inverse" 2>&1 | tail -3 >> $out
echo "== default with check" >> $out
python tools/time_inv.py --n 10048 --reps 3 --check --tag default 2>&1 >> $out
for v in "$@"; do run $v; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt; rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python /root/repo/tools/time_inv.py --reps 2 > /dev/null 2>&1
python /root/repo/tools/experiments/inv_timeline2.py "$(ls -t /tmp/kt/*/*kernel_trace.csv | head -1)" ${MINUS:-200} > /root/repo/gpurun_out/inv_timeline.txt 2>&1
cd /root/repo; cat $out
