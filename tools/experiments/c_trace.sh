cd /root/repo
for P in 128 256 512; do
  DCA_SWEEP_PANEL=$P python tools/time_inv.py --n 4032 --reps 5 --tag panel$P 2>/dev/null | tail -1
done
for P in 128 256; do
  DCA_SWEEP_PANEL=$P DCA_CHOLINV_TRACE=1 python tools/time_inv.py --n 4032 --reps 3 2> gpurun_out/c_trace_$P.raw > /dev/null
  python tools/experiments/sweep_trace_summary.py gpurun_out/c_trace_$P.raw > gpurun_out/c_trace_$P.txt
done
head -24 gpurun_out/c_trace_256.txt
