cd /root/repo
for R in 0 4 8; do
  for N in 4032 5056 6016; do
    DCA_SWEEP_RESERVE=$R DCA_SWEEP_RESERVE_MAX_N=100000 python tools/time_inv.py --n $N --reps 5 --tag reserve$R 2>/dev/null | tail -1
  done
done
DCA_SWEEP_RESERVE=4 DCA_SWEEP_RESERVE_MAX_N=100000 python tools/time_inv.py --n 4032 --reps 2 --check --tag check 2>&1 | tail -2
for R in 0 4; do
  DCA_SWEEP_RESERVE=$R DCA_SWEEP_RESERVE_MAX_N=100000 python tools/experiments/mf_inv_time.py reserve$R
done
DCA_SWEEP_RESERVE=4 DCA_SWEEP_RESERVE_MAX_N=100000 DCA_SWEEP_PRIO_CAP=128 python tools/experiments/mf_inv_time.py reserve4-prio128
