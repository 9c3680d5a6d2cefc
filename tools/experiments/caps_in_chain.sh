cd /root/repo
python tools/experiments/mf_inv_time.py default
for C in 352 416 448; do DCA_SWEEP_CAP=$C python tools/experiments/mf_inv_time.py cap$C; done
for P in 32 96 128; do DCA_SWEEP_PRIO_CAP=$P python tools/experiments/mf_inv_time.py prio$P; done
DCA_SWEEP_CAP=416 DCA_SWEEP_PRIO_CAP=96 python tools/experiments/mf_inv_time.py cap416-prio96
DCA_SWEEP_CAP=352 DCA_SWEEP_PRIO_CAP=128 python tools/experiments/mf_inv_time.py cap352-prio128
python tools/experiments/mf_inv_time.py default-again
