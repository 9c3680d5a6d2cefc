cd /root/repo
for M in 0 1; do
  echo "== merge $M"
  DCA_SCATTER_MERGE=$M python tools/time_eval.py --L 200 --N 10000 --q 21 --seed 12345 --reps 20
  DCA_SCATTER_MERGE=$M python tools/time_eval.py --L 500 --N 50000 --q 21 --seed 12346 --reps 6
  DCA_SCATTER_MERGE=$M python tools/time_eval.py --L 150 --N 200000 --q 5 --seed 12347 --reps 10
  DCA_SCATTER_MERGE=$M python - <<'PY'
import sys, hashlib
sys.path.insert(0, '/root/repo')
import numpy as np
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
for (L, N, q, seed) in ((200, 10000, 21, 12345), (333, 20000, 21, 5), (500, 50000, 21, 12346)):
    X = dedup(generate(L, N, q, seed))
    c = _lib.Context(0, _lib.DCA_F32); c.set_msa(X, q); c.compute_weights(0.8, _lib.DCA_F32); c.plm_configure(1.0, 50.0); c.plm_init_x()
    c.plm_lbfgs_begin(100); st = c.plm_lbfgs_iterate(3)
    fx = c.plm_gradient(); g = c.plm_get_g()
    print(L, N, repr(fx), hashlib.sha256(g.tobytes()).hexdigest()[:16], st.evaluations)
    c.close()
PY
done
