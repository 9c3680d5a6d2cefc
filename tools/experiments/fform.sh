cd /root/repo
for r in 1 2; do
for C in 0 1; do
  DCA_SWEEP_COPY=$C python tools/experiments/mf_inv_time.py copy$C
done
done
