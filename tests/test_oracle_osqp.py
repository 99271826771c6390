"""The oracle's ADMM (oracle/osqp_ref.c) against known answers.

OSQP itself is absent here; what pins the restatement is
(a) the analytic known-answer QPs the reference's own tests use at the OSQP
boundary, (b) an ADMM-independent KKT check, (c) internal consistency between
its two linear-system back-ends and its two duplicate-row representations.
"""
import numpy as np
import pytest

from oracle import osqp_ref as o

INF = np.inf


def test_scalar_quadratic_minimum():
    # min x^2 - 4x -> 2   (tests/sco_osqp/test_variable.py:39-67 of the reference)
    r = o.solve(np.array([[2.0]]), [-4.0], np.array([[1.0]]), [-INF], [INF])
    assert r.info.status_val == 1
    assert np.allclose(r.x, [2.0])


def test_scalar_quadratic_in_trust_box():
    # same objective inside [3, 5] -> 3   (test_variable.py:69-96)
    r = o.solve(np.array([[2.0]]), [-4.0], np.array([[1.0]]), [3.0], [5.0])
    assert r.info.status_val == 1
    assert np.allclose(r.x, [3.0])
    assert np.allclose(r.y, [-2.0], atol=1e-5)      # multiplier of the active lower bound


def test_projection_onto_box_and_equalities():
    # closest point to the origin with x <= c   (test_prob.py:48-76)
    for c, want in (([1.0, 1.0], [0.0, 0.0]), ([-1.0, 1.0], [-1.0, 0.0]), ([-1.0, -1.0], [-1.0, -1.0])):
        A = np.vstack([np.eye(2), np.eye(2)])
        l = [-INF, -INF, -INF, -INF]; u = [c[0], c[1], INF, INF]
        r = o.solve(2 * np.eye(2), [0.0, 0.0], A, l, u)
        assert r.info.status_val == 1 and np.allclose(r.x, want, atol=1e-6)
    # x == (5, -10)   (test_prob.py:77-93)
    A = np.vstack([np.eye(2), np.eye(2)])
    r = o.solve(2 * np.eye(2), [0.0, 0.0], A, [5.0, -10.0, -INF, -INF], [5.0, -10.0, INF, INF])
    assert np.allclose(r.x, [5.0, -10.0], atol=1e-6)


def test_l1_penalty_qp():
    # min x^2 + |x - 4| via slacks p, n -> x = 0.5   (test_prob.py:315-349)
    P = np.diag([2.0, 0.0, 0.0]); q = [0.0, 1.0, 1.0]
    A = np.array([[1.0, -1.0, 1.0], [1, 0, 0], [0, 1, 0], [0, 0, 1]])
    r = o.solve(P, q, A, [4.0, -INF, 0.0, 0.0], [4.0, INF, INF, INF])
    assert r.info.status_val == 1 and np.allclose(r.x[0], 0.5, atol=1e-5)


def _random_qp(seed, n=25, m=40):
    rng = np.random.default_rng(seed)
    M = rng.standard_normal((n, n)); P = 0.1 * M @ M.T
    A = rng.standard_normal((m, n)) * (rng.random((m, n)) < 0.3)
    A = np.vstack([A, np.eye(n)])
    l = np.concatenate([-rng.random(m), -np.ones(n)]); u = np.concatenate([rng.random(m), np.ones(n)])
    l[:4] = u[:4]; u[4:8] = INF; l[8:10] = -INF
    return P, rng.standard_normal(n), A, l, u


@pytest.mark.parametrize("seed", range(4))
def test_kkt_conditions_hold_at_the_answer(seed):
    P, q, A, l, u = _random_qp(seed)
    r = o.solve(P, q, A, l, u)
    assert r.info.status_val == 1
    prim, stat, comp = o.kkt_violation(P, q, A, l, u, r.x, r.y)
    assert prim < 5e-6 and stat < 5e-6 and comp < 5e-5


@pytest.mark.parametrize("seed", range(3))
def test_kkt_ldl_and_reduced_cholesky_agree(seed):
    P, q, A, l, u = _random_qp(seed)
    a = o.solve(P, q, A, l, u, linsys=0); b = o.solve(P, q, A, l, u, linsys=1)
    assert a.info.iter == b.info.iter and a.info.status_val == b.info.status_val
    assert np.abs(a.x - b.x).max() < 1e-10 and np.abs(a.y - b.y).max() < 1e-9


@pytest.mark.parametrize("seed", range(3))
def test_row_multiplicity_equals_physical_duplicates(seed):
    P, q, A, l, u = _random_qp(seed)
    w = np.ones(A.shape[0], dtype=int); w[10:25] = 3; w[30] = 5
    phys = o.solve(P, q, A, l, u, w=w, expand_dups=1)
    fold = o.solve(P, q, A, l, u, w=w, expand_dups=0)
    red = o.solve(P, q, A, l, u, w=w, expand_dups=0, linsys=1)
    # physically repeating the rows through the Python side gives the same thing
    rep = np.repeat(np.arange(A.shape[0]), w)
    manual = o.solve(P, q, A[rep], l[rep], u[rep])
    for other in (fold, red, manual):
        assert other.info.iter == phys.info.iter
        assert np.abs(other.x - phys.x).max() < 1e-10


def test_primal_infeasible_is_reported():
    A = np.array([[1.0], [1.0]])
    r = o.solve(np.array([[1.0]]), [0.0], A, [1.0, -INF], [INF, 0.0])      # x >= 1 and x <= 0
    assert r.info.status_val == -3


def test_dual_infeasible_is_reported():
    r = o.solve(np.array([[0.0]]), [1.0], np.array([[1.0]]), [-INF], [0.0])  # min x, x <= 0
    assert r.info.status_val == -4


def test_max_iter_and_inaccurate_statuses():
    P, q, A, l, u = _random_qp(0)
    r = o.solve(P, q, A, l, u, max_iter=30)
    assert r.info.status_val == -2 and r.info.iter == 30
    full = o.solve(P, q, A, l, u)
    # stop a little before convergence: the 10x looser test of OSQP applies
    r2 = o.solve(P, q, A, l, u, max_iter=full.info.iter - 25)
    assert r2.info.status_val in (2, -2)


def test_empty_constraint_matrix():
    r = o.solve(np.array([[2.0, 0.0], [0.0, 4.0]]), [-2.0, -4.0], np.zeros((0, 2)), [], [])
    assert r.info.status_val == 1 and np.allclose(r.x, [1.0, 1.0], atol=1e-6)


def test_termination_trace_is_every_25_iterations():
    P, q, A, l, u = _random_qp(1)
    r = o.solve(P, q, A, l, u, trace_cap=4096)
    assert np.array_equal(r.trace[:, 0], 25 * (1 + np.arange(r.trace.shape[0])))
    assert r.trace[-1, 0] == r.info.iter


def test_adaptive_rho_reaches_the_same_answer():
    """adaptive_rho (solver.py:39 lets a caller turn it on; off by default, osqp_utils.py:13): OSQP's update rule as
    restated in osqp_ref.c.  Same optimum (KKT check), both linear-system back-ends and both duplicate-row forms
    agree with each other, and a badly scaled penalty QP needs a fraction of the iterations."""
    from test_qp_plan import penalty_qp
    rng = np.random.default_rng(29)
    P, q, A, l, u = penalty_qp(rng, 20, 7, 10)
    fixed = o.solve(P, q, A, l, u)
    ad = o.solve(P, q, A, l, u, adaptive_rho=1)
    assert fixed.info.status_val == ad.info.status_val == 1
    assert ad.info.rho_updates >= 1 and ad.info.rho_estimate != 0.1 and fixed.info.rho_updates == 0
    assert ad.info.iter < 0.5 * fixed.info.iter
    assert np.abs(ad.x - fixed.x).max() < 1e-4
    prim, stat, comp = o.kkt_violation(P, q, A, l, u, ad.x, ad.y)
    assert prim < 5e-6 and stat < 5e-5 and comp < 5e-4
    red = o.solve(P, q, A, l, u, adaptive_rho=1, linsys=1)
    assert red.info.iter == ad.info.iter and np.abs(red.x - ad.x).max() < 1e-9
    w = np.ones(A.shape[0], dtype=int); w[7:207] = 2
    phys = o.solve(P, q, A, l, u, w=w, expand_dups=1, adaptive_rho=1)
    fold = o.solve(P, q, A, l, u, w=w, expand_dups=0, adaptive_rho=1)
    assert phys.info.iter == fold.info.iter and phys.info.rho_updates == fold.info.rho_updates
    assert np.abs(phys.x - fold.x).max() < 1e-9


def test_adaptive_rho_interval_and_tolerance():
    P, q, A, l, u = _random_qp(1)
    base = o.solve(P, q, A, l, u, adaptive_rho=1)
    assert o.solve(P, q, A, l, u, adaptive_rho=1, adaptive_rho_interval=100).info.iter == base.info.iter   # 0 = 4 x 25
    never = o.solve(P, q, A, l, u, adaptive_rho=1, adaptive_rho_tolerance=1e9)
    assert never.info.rho_updates == 0 and never.info.iter == o.solve(P, q, A, l, u).info.iter
