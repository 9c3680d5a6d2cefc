cd /root/repo
for M in 0 1; do
  echo "== fold merge $M"
  DCA_FOLD_MERGE=$M python tools/time_eval.py --L 200 --N 10000 --q 21 --seed 12345 --reps 20
  DCA_FOLD_MERGE=$M python tools/time_eval.py --L 150 --N 200000 --q 5 --seed 12347 --reps 10
  DCA_FOLD_MERGE=$M python - <<'PY'
import sys, hashlib
sys.path.insert(0, '/root/repo')
import numpy as np
from pydca_amd import _lib
from tools.gen_msa import dedup, generate
for (L, N, q, seed, prec) in ((200, 10000, 21, 12345, 32), (150, 30000, 5, 7, 32), (60, 3000, 21, 5, 64), (150, 20000, 5, 7, 64)):
    X = dedup(generate(L, N, q, seed))
    P = _lib.DCA_F64 if prec == 64 else _lib.DCA_F32
    c = _lib.Context(0, P); c.set_msa(X, q); c.compute_weights(0.8, P); c.plm_configure(1.0, 50.0); c.plm_init_x()
    c.plm_lbfgs_begin(100); st = c.plm_lbfgs_iterate(3)
    fx = c.plm_gradient(); g = c.plm_get_g(np.float64 if prec == 64 else np.float32)
    print(L, N, q, prec, repr(fx), repr(st.fx), hashlib.sha256(g.tobytes()).hexdigest()[:16], st.evaluations)
    c.close()
PY
done
