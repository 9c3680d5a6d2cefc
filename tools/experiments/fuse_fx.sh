cd /root/repo
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -k "lbfgs or P3 or trajectory or cap" 2>&1 | tail -3
for W in C E; do for F in 0 1 0 1; do
  DCA_PLM_FUSE_FX=$F timeout 300 python bench.py --workload $W --steps 40 --warmup 5 2>/dev/null | tail -1 > /tmp/b.json
  python -c "
import json;j=json.load(open('/tmp/b.json'));print('$W fuse=$F', round(j['value'],1), round(j['ms_per_step'],4), j['fx'])"
done; done
