"""-m gpu half of tests/test_adjudicate.py: the two QPs of r03's parity misses on the device."""
import os
import numpy as np
import pytest
import scipy.sparse as sp

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "adjudicate_r03.npz")
TIER_ENVS = [{}, dict(SCO_QP_NO_RL="1"), dict(SCO_QP_NO_RL="1", SCO_QP_NO_REG="1"),
             dict(SCO_QP_NO_RL="1", SCO_QP_NO_REG="1", SCO_QP_NO_FAST="1"), dict(SCO_QP_FORCE_BIG="1")]


@pytest.fixture(scope="module")
def gpu():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return 1


@pytest.mark.parametrize("i", [43, 57])
@pytest.mark.parametrize("env", TIER_ENVS, ids=["default", "register", "sliced-ELL", "generic", "global-memory"])
def test_every_device_tier_counts_like_the_four_cpu_routes(gpu, monkeypatch, i, env):
    """On the QP exactly as the oracle's loop built it the device ends at the oracle's iteration count on every ADMM tier, and
    at the check where the r03 device LOOP stopped its iterate stands as close to the x87 trajectory as the oracle's own
    float64 routes do (|x - x_x87| ~ 1e-9, dual residual equal to ~1e-4 of the tolerance)."""
    from sco_py_amd import _lib as L
    fx = np.load(GOLD)
    t = "p%d_" % i
    P, q, A, l, u, w = (fx[t + k] for k in ("P", "q", "A", "l", "u", "w"))
    it_dev, it_orc = (int(v) for v in fx[t + "counts"])
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    n, m = len(q), len(l)
    Pu = sp.triu(sp.csc_matrix(P), format="csc"); Pu.sort_indices()
    Ac = sp.csc_matrix((A != 0).astype(float)); Ac.sort_indices()
    pr, pc = Pu.indices, np.repeat(np.arange(n), np.diff(Pu.indptr))
    ar, ac = Ac.indices, np.repeat(np.arange(n), np.diff(Ac.indptr))
    qp = L.BatchedQP(1, n, m, Pu.indptr, Pu.indices, Ac.indptr, Ac.indices)
    try:
        qp.load(P[pr, pc][None], q[None], A[ar, ac][None], l[None], u[None], w[None].astype(np.int32))
        x, y, st, it, rs = qp.solve(L.default_qp_settings())
        x2, y2, st2, it2, rs2 = qp.solve(L.default_qp_settings(max_iter=it_dev))
    finally:
        qp.close()
    assert (st[0], it[0]) == (1, it_orc)
    assert (st2[0], it2[0]) == (2, it_dev)                      # cut at the r03 loop's count: not solved there
    eps = 1e-6 + 1e-9 * np.abs(q).max()
    ref = fx[t + "x87_kkt_checks"][1, 3]
    assert abs(rs2[0][1] - ref) / eps < 6e-4 and rs2[0][1] > eps
    assert np.abs(x2[0] - fx[t + "x87_kkt_x"]).max() < 2e-8
    yr = fx[t + "x87_kkt_y"]
    assert np.abs(y2[0] - yr).max() / np.abs(yr).max() < 5e-12


@pytest.mark.parametrize("i,qp_index", [(43, 12), (57, 13)])
def test_the_two_problems_through_the_device_loop(gpu, i, qp_index):
    """The whole SQP solve of the two problems.  With forward-mode Jacobians on both sides the loops' states agree to 1e-8
    when the adjudicated QP is built (an earlier QP of the problem ended at max_iter, status 2), which is enough to move that
    QP's termination by a few checks: every decision, every status and every OTHER iteration count agree, success agrees,
    x to 5e-6.  If the counts of that QP agree as well (other rounding), so much the better."""
    from oracle import arm_family as af, sco_ref as sr
    from sco_py_amd import batch as sb
    KW = dict(program=True, variant="dynamics", d=3, T=10, K=1)
    ref = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(i, **KW), analytic_jac=True), None, emulate_memo=True)
    arrays, _ = af.make_batch(1, first=i, **KW)
    res = sb.solve_batch(arrays, analytic_jac=True)
    g, tr = res.trace[0], ref.trace[:64]
    assert g.shape == tr.shape
    assert np.array_equal(g[:, 0], tr[:, 0]) and np.array_equal(g[:, 6], tr[:, 6])
    differ = np.flatnonzero(g[:, 7] != tr[:, 7])
    assert set(differ.tolist()) <= {qp_index}, differ
    assert tr[:qp_index, 6].max() == 2                           # an unconverged QP came first
    assert bool(res.success[0]) == ref.success
    assert np.abs(res.x[0] - ref.x).max() < 5e-6


def test_the_device_loop_is_the_oracles_reduced_route_on_the_two_sweep_problems(gpu):
    """r04 sweeps (profiles/r04_parity_sweep.txt): problem 6072 (dynamics) and 6233 (jerk), forward-mode Jacobians, and 6059
    (7 x 20 with general affine rows) -- the three "mismatches" against the oracle's KKT route.  Against the oracle's float64 REDUCED route (the device's algebra) the device
    agrees decision for decision, status for status, count for count on both, and in x to the usual 1e-6 on the first."""
    from oracle import arm_family as af
    from sco_py_amd import batch as sb
    from test_adjudicate import WIDE, WIDE_AJ, oracle_route
    for i in (6072, 6233, 6059):
        ref = oracle_route(i, linsys=1)
        arrays, _ = af.make_batch(1, first=i, **WIDE[i])
        res = sb.solve_batch(arrays, analytic_jac=WIDE_AJ[i])
        g, tr = res.trace[0], ref.trace[:64]
        assert g.shape == tr.shape and np.array_equal(g[:, 0], tr[:, 0]) and np.array_equal(g[:, 6:8], tr[:, 6:8]), i
        assert bool(res.success[0]) == ref.success
        assert np.abs(res.x[0] - ref.x).max() < (1e-4 if i == 6233 else 1e-6), (i, np.abs(res.x[0] - ref.x).max())
