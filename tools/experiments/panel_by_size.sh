cd /root/repo
for N in 2560 3072 4032 5056 6016 7040; do
  for P in 256 384 512; do
    DCA_SWEEP_PANEL=$P python tools/time_inv.py --n $N --reps 6 --tag panel$P 2>/dev/null | tail -1
  done
done
