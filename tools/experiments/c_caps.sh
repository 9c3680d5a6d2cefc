cd /root/repo
for N in 3072 4032; do
for P in 256 384 512; do
  DCA_SWEEP_PANEL=$P python tools/time_inv.py --n $N --reps 6 --tag panel$P 2>/dev/null | tail -1
  DCA_SWEEP_PANEL=$P DCA_SWEEP_PER_CU=1 DCA_SWEEP_STAGES=3 DCA_SWEEP_CAP=192 python tools/time_inv.py --n $N --reps 6 --tag panel$P-percu1-192 2>/dev/null | tail -1
done; done
