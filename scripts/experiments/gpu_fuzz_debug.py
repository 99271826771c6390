import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_qp_gpu as T
from oracle import osqp_ref as o
from sco_py_amd import _lib
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4
rng = np.random.default_rng(500 + seed)
n, m = int(rng.integers(4, 40)), int(rng.integers(0, 40))
base = T._random_qp(rng, n, m, float(rng.uniform(0.05, 0.4)))
P, q, A, lo, hi = base
print("n", n, "m", A.shape[0], "empty rows", int((np.abs(A).sum(axis=1) == 0).sum()), "empty cols of P", int((np.abs(P).sum(axis=0) == 0).sum()))
ref = o.solve(*base)
print("oracle", ref.info.status_val, ref.info.iter, ref.info.pri_res, ref.info.dua_res)
for env in [{}, {"SCO_QP_NO_RL": "1"}, {"SCO_QP_NO_RL": "1", "SCO_QP_NO_REG": "1"}, {"SCO_QP_NO_RL": "1", "SCO_QP_NO_REG": "1", "SCO_QP_NO_FAST": "1"},
            {"SCO_QP_FORCE_BIG": "1"}, {"SCO_QP_FORCE_BIG": "1", "SCO_QP_NO_BT": "1"}, {"SCO_QP_NO_ELIM": "1"}]:
    for k in ("SCO_QP_NO_RL", "SCO_QP_NO_REG", "SCO_QP_NO_FAST", "SCO_QP_FORCE_BIG", "SCO_QP_NO_BT", "SCO_QP_NO_ELIM"):
        os.environ.pop(k, None)
    os.environ.update(env)
    nn, mm, Pp, Pi, Ap, Ai, Pval, qq, Aval, l, u = T._stack([base])
    qp = _lib.BatchedQP(1, nn, mm, Pp, Pi, Ap, Ai)
    qp.load(Pval, qq, Aval, l, u)
    x, y, st, it, res = qp.solve(); info = qp.info(); qp.close()
    print(env, info, st[0], it[0], res[0], "dx %.2e" % np.abs(x[0] - ref.x).max(), "dy %.2e" % np.abs(y[0] - ref.y).max())
