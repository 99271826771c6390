"""The ADMM (oracle/osqp_ref.c, and through it every iterate-level claim) pinned to something that is not an ADMM.

tests/golden/qp_exact_7x20.npz holds solutions of the three golden QPs of the 7-DOF x 20 problem computed by an
interior-point + active-set method in extended precision (oracle/qp_exact.py).  Here:
  * the stored (x*, y*) certify themselves: optimality conditions by residual arithmetic, no solver;
  * the oracle ADMM's answers to the converging QPs sit within OSQP's accuracy of x*;
  * the second penalty QP -- stiffened by the reference's compounded penalty (Q1, prob.py:414-426) -- really is
    unconverged after max_iter = 100 000 iterations: OSQP's own residuals, recomputed here from the returned (x, y)
    alone, exceed its tolerances, and the iterate is far from x*.  That QP is ~95 % of the timed work of bench.py.
"""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
from make_qp_exact import golden_qp          # noqa: E402
from oracle import osqp_ref, qp_exact as qe  # noqa: E402

GOLD = os.path.join(HERE, "golden")
REF_SETTINGS = dict(rho=0.1, sigma=5e-10, eps_abs=1e-6, eps_rel=1e-9, max_iter=100000)    # osqp_utils.py:10-15


@pytest.fixture(scope="module")
def data():
    return np.load(os.path.join(GOLD, "trajopt_7x20.npz")), np.load(os.path.join(GOLD, "qp_exact_7x20.npz"))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_stored_solutions_satisfy_the_optimality_conditions(data, k):
    g, ex = data
    pre = "p0_qp%d" % k
    rep = qe.check_kkt(*golden_qp(g, pre), ex[pre + "_x"], ex[pre + "_y"])
    assert rep["stationarity"] < 1e-9 and rep["primal"] < 1e-12 and rep["complementarity"] < 1e-9 and rep["dual_sign"] == 0.0
    assert abs(rep["objective"] - float(ex[pre + "_obj"])) < 1e-10 * (1 + abs(rep["objective"]))


@pytest.mark.parametrize("k", [0, 1, 2])
def test_interior_point_reproduces_the_fixture(data, k):
    g, ex = data
    pre = "p0_qp%d" % k
    x, y, rep = qe.solve_exact(*golden_qp(g, pre))
    assert np.abs(x - ex[pre + "_x"]).max() < 1e-10
    assert max(rep["stationarity"], rep["primal"], rep["complementarity"]) < 1e-12


@pytest.mark.parametrize("k,tol", [(0, 1e-7), (1, 2e-4)])
def test_converging_qps_admm_answer_is_the_optimum(data, k, tol):
    """tol: OSQP stops at eps_abs = 1e-6 on residuals, not on x; on the scaled penalty QP that is 4e-5 in x."""
    g, ex = data
    pre = "p0_qp%d" % k
    P, q, A, l, u = golden_qp(g, pre)
    kw = REF_SETTINGS if k else {}                  # the projection QP runs with OSQP's defaults (Q7)
    res = osqp_ref.solve(P, q, A, l, u, **kw)
    assert res.info.status_val == 1 and res.info.iter == int(g[pre + "_iters"])
    assert np.abs(res.x - ex[pre + "_x"]).max() < tol
    assert np.abs(g[pre + "_x"] - ex[pre + "_x"]).max() < tol            # the golden run's recorded answer, too
    obj = 0.5 * res.x @ P @ res.x + q @ res.x
    assert abs(obj - float(ex[pre + "_obj"])) < 1e-6 * (1 + abs(obj))


def test_compounded_penalty_qp_is_unconverged_at_max_iter(data):
    g, ex = data
    pre = "p0_qp2"
    P, q, A, l, u = golden_qp(g, pre)
    res = osqp_ref.solve(P, q, A, l, u, **REF_SETTINGS)
    assert res.info.status_val == -2 and res.info.iter == 100000 == int(g[pre + "_iters"])
    r = qe.osqp_residuals(P, q, A, l, u, res.x, res.y)
    eps_p = REF_SETTINGS["eps_abs"] + REF_SETTINGS["eps_rel"] * r["prim_scale"]
    eps_d = REF_SETTINGS["eps_abs"] + REF_SETTINGS["eps_rel"] * r["dual_scale"]
    # OSQP stops when BOTH residuals are under their tolerance; recomputed from (x, y) alone at least one is far above
    assert r["primal_lower_bound"] > 100 * eps_p or r["dual"] > 100 * eps_d, (r, eps_p, eps_d)
    # and the iterate is nowhere near the optimum, in x or in objective value
    assert np.abs(res.x - ex[pre + "_x"]).max() > 0.1
    obj = 0.5 * res.x @ P @ res.x + q @ res.x
    assert obj > float(ex[pre + "_obj"]) + 1.0
    # the slack cost that does it: quirk Q1 multiplies it by the penalty coefficient on every update_obj
    assert np.abs(q).max() == pytest.approx(1e6)
