"""The two QPs of r03's 2 / 512 parity misses on the device, tier by tier, beside the four CPU routes of the fixture
(tests/golden/adjudicate_r03.npz, tests/golden/make_adjudicate.py).  VERDICT r03 item 7.

    python scripts/gpu_adjudicate.py > gpurun_out/r04_adjudicate.txt
"""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sco_py_amd import _lib as L

g = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "adjudicate_r03.npz"))
TIERS = (("default", {}), ("no row-local", dict(SCO_QP_NO_RL="1")), ("no row-local, no register", dict(SCO_QP_NO_RL="1", SCO_QP_NO_REG="1")),
         ("sliced-ELL off too (generic)", dict(SCO_QP_NO_RL="1", SCO_QP_NO_REG="1", SCO_QP_NO_FAST="1")),
         ("global memory", dict(SCO_QP_FORCE_BIG="1")), ("Cholesky factor", dict(SCO_QP_FACTOR_CHOLESKY="1")),
         ("no elimination", dict(SCO_QP_NO_ELIM="1")))
import ctypes as C
lib = L.load()
lib.sco_debug_qp_tiers.restype = C.c_int; lib.sco_debug_qp_tiers.argtypes = [C.c_void_p]

for i in (43, 57):
    t = "p%d_" % i
    P, q, A, l, u, w = (g[t + k] for k in ("P", "q", "A", "l", "u", "w"))
    it_dev, it_orc = (int(v) for v in g[t + "counts"])
    n, m = len(q), len(l)
    Pu = sp.triu(sp.csc_matrix(P), format="csc"); Pu.sort_indices()
    Ac = sp.csc_matrix((A != 0).astype(float)); Ac.sort_indices()
    pr, pc = Pu.indices, np.repeat(np.arange(n), np.diff(Pu.indptr))
    ar, ac = Ac.indices, np.repeat(np.arange(n), np.diff(Ac.indptr))
    Pval = P[pr, pc][None]; Aval = A[ar, ac][None]
    print("problem %d: n %d m %d |q|inf %.1e  r03 device %d, oracle %d iterations" % (i, n, m, np.abs(q).max(), it_dev, it_orc))
    for name in ("f64_kkt", "f64_reduced", "x87_kkt", "x87_reduced"):
        ck = g[t + name + "_checks"]
        print("   CPU %-12s ends at %6d; dua_res at the check of iteration %d: %.9e, at %d: %.9e" % (name, g[t + name + "_final"][1], ck[1, 0], ck[1, 3], ck[3, 0], ck[3, 3]))
    xr, yr = g[t + "x87_kkt_x"], g[t + "x87_kkt_y"]
    print("   |x - x_x87| / |y - y_x87|inf rel at iteration %d:  f64_kkt %.2e / %.2e   f64_reduced %.2e / %.2e" % (
        it_dev, np.abs(g[t + "f64_kkt_x"] - xr).max(), np.abs(g[t + "f64_kkt_y"] - yr).max() / np.abs(yr).max(),
        np.abs(g[t + "f64_reduced_x"] - xr).max(), np.abs(g[t + "f64_reduced_y"] - yr).max() / np.abs(yr).max()))
    for tname, env in TIERS:
        for k in ("SCO_QP_NO_RL", "SCO_QP_NO_REG", "SCO_QP_NO_FAST", "SCO_QP_FORCE_BIG", "SCO_QP_FACTOR_CHOLESKY", "SCO_QP_NO_ELIM"):
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            qp = L.BatchedQP(1, n, m, Pu.indptr, Pu.indices, Ac.indptr, Ac.indices)
        except Exception as e:
            print("   device %-30s: %s" % (tname, e)); continue
        tiers = lib.sco_debug_qp_tiers(qp._h)
        qp.load(Pval, q[None], Aval, l[None], u[None], w[None].astype(np.int32))
        x, y, st, it, rs = qp.solve(L.default_qp_settings())
        x2, y2, st2, it2, rs2 = qp.solve(L.default_qp_settings(max_iter=it_dev))
        qp.close()
        print("   device %-30s (tiers %3d): ends at %6d status %d; capped at %d: status %d, resid %s, |x - x_x87| %.2e, |y - y_x87| rel %.2e" % (
            tname, tiers, it[0], st[0], it_dev, st2[0], np.array2string(rs2[0], precision=9), np.abs(x2[0] - xr).max(), np.abs(y2[0] - yr).max() / np.abs(yr).max()))
