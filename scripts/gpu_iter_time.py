"""Per-iteration time of the row-local ADMM kernel against the number of busy CUs: the same 7x20-shaped QP
replicated B times (identical iteration counts), max_iter fixed."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from sco_py_amd import _build
if os.environ.get("SCO_LIB_OVERRIDE"):
    _build.LIB = os.environ["SCO_LIB_OVERRIDE"]; print("library", _build.LIB)
from sco_py_amd import _lib
from test_qp_plan import penalty_qp
from test_qp_gpu import _stack
rng = np.random.default_rng(9)
pr = penalty_qp(rng, 20, 7, 10)
import os
CHK = int(os.environ.get("CHECK", "25"))
st = _lib.default_qp_settings(max_iter=20000, eps_abs=1e-30, eps_rel=1e-30, check_termination=CHK)     # never converges: 20000 iterations each
print("check_termination", CHK)
for B in (256,):
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack([pr] * B)
    qp = _lib.BatchedQP(B, n, m, Pp, Pi, Ap, Ai)
    qp.load(Pval, q, Aval, l, u)
    qp.solve(st)
    x, y, s, it, res = qp.solve(st)
    tm = qp.last_timing()
    waves = -(-B // 256)
    print("B=%4d iterations %d admm %.2f ms -> %.3f us per iteration per CU pass (%d passes), setup %.2f ms" % (
        B, it[0], tm["admm_ms"], 1e3 * tm["admm_ms"] / it[0] / waves, waves, tm["setup_ms"]), flush=True)
    qp.close()
