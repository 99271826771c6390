"""ADMM-free reference solutions of the golden 7-DOF x 20 QPs (tests/golden/trajopt_7x20.npz: the projection QP,
the first penalty QP, and the second penalty QP that quirk Q1 makes too stiff for rho = 0.1).

    python tests/golden/make_qp_exact.py        ->  tests/golden/qp_exact_7x20.npz

Solver: oracle/qp_exact.py (interior point + active-set polish in extended precision).  The fixture holds x*, y*
and the optimal objective of every QP; tests/test_qp_exact.py re-checks the optimality conditions of the stored pair
with residual arithmetic alone, so the fixture does not have to be trusted."""
import os
import sys

import numpy as np
import scipy.sparse as sp

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import qp_exact as qe            # noqa: E402


def golden_qp(g, pre):
    def mat(name):
        return sp.coo_matrix((g[name + "_val"], (g[name + "_row"], g[name + "_col"])), shape=tuple(g[name + "_shape"])).toarray()
    P = mat(pre + "_P")
    P = np.triu(P) + np.triu(P, 1).T
    return P, g[pre + "_q"], mat(pre + "_A"), g[pre + "_l"], g[pre + "_u"]


if __name__ == "__main__":
    g = np.load(os.path.join(HERE, "trajopt_7x20.npz"))
    out = {}
    for k in range(int(g["p0_n_qp"])):
        pre = "p0_qp%d" % k
        x, y, rep = qe.solve_exact(*golden_qp(g, pre))
        print(pre, rep, "|x_admm - x*| %.3e" % np.abs(g[pre + "_x"] - x).max())
        out[pre + "_x"] = x; out[pre + "_y"] = y; out[pre + "_obj"] = np.float64(rep["objective"])
    np.savez_compressed(os.path.join(HERE, "qp_exact_7x20.npz"), **out)
    print("qp_exact_7x20.npz", os.path.getsize(os.path.join(HERE, "qp_exact_7x20.npz")), "bytes")
