cd /root/repo
DCA_CHOLINV_TRACE=1 python tools/time_inv.py --n 4032 --reps 3 2> gpurun_out/c_trace.raw > /dev/null
python tools/experiments/sweep_trace_summary.py gpurun_out/c_trace.raw > gpurun_out/r06_inverse_events_n4032.txt
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/kt -- python /root/repo/tools/time_inv.py --n 4032 --reps 3 > /dev/null 2>&1
f=$(ls /tmp/kt/*/*kernel_trace.csv | head -1); python /root/repo/tools/experiments/chain_gaps.py $f > /root/repo/gpurun_out/r06_inverse_chain_n4032.txt
tail -30 /root/repo/gpurun_out/r06_inverse_chain_n4032.txt; head -8 /root/repo/gpurun_out/r06_inverse_events_n4032.txt
