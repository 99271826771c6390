"""HIP QP kernels (through the C ABI) against the oracle ADMM: x within 1e-6 (the
north_star tolerance; in practice ~1e-13), identical status and iteration counts."""
import os

import numpy as np
import pytest
import scipy.sparse as sp

from oracle import osqp_ref as o
from sco_py_amd import _lib
from test_qp_plan import penalty_qp

pytestmark = pytest.mark.gpu
TOL = 1e-6          # abs tolerance stated by BASELINE.json north_star


def _stack(probs):
    P0, q0, A0, l0, u0 = probs[0]
    n, m = len(q0), len(l0)
    Pu = sp.triu(sp.csc_matrix(P0), format="csc"); Pu.sort_indices()
    Ac = sp.csc_matrix((A0 != 0).astype(float)); Ac.sort_indices()
    pr, pc = Pu.indices, np.repeat(np.arange(n), np.diff(Pu.indptr))
    ar, ac = Ac.indices, np.repeat(np.arange(n), np.diff(Ac.indptr))
    Pval = np.stack([p[0][pr, pc] for p in probs]); Aval = np.stack([p[2][ar, ac] for p in probs])
    q = np.stack([p[1] for p in probs]); l = np.stack([p[3] for p in probs]); u = np.stack([p[4] for p in probs])
    return n, m, Pu.indptr, Pu.indices, Ac.indptr, Ac.indices, Pval, q, Aval, l, u


def _check(probs, w=None, settings=None, check=None, resid_tol=1e-9, **okw):
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack(probs)
    qp = _lib.BatchedQP(len(probs), n, m, Pp, Pi, Ap, Ai)
    try:
        qp.load(Pval, q, Aval, l, u, w)
        x, y, st, it, res = qp.solve(settings)
        info = qp.info()
        rho, nupd = qp.adaptive_info()
    finally:
        qp.close()
    for b in (range(len(probs)) if check is None else check):
        ref = o.solve(*probs[b], w=None if w is None else w[b], **okw)
        assert st[b] == ref.info.status_val, (b, st[b], ref.info.status_val)
        assert it[b] == ref.info.iter, (b, it[b], ref.info.iter)
        if okw.get("adaptive_rho"):
            assert nupd[b] == ref.info.rho_updates and abs(rho[b] - ref.info.rho_estimate) < 1e-4 * ref.info.rho_estimate
        if ref.info.status_val in (1, 2, -2):
            assert np.abs(x[b] - ref.x).max() < TOL, (b, np.abs(x[b] - ref.x).max())
            assert np.abs(y[b] - ref.y).max() < 1e-5 * (1 + np.abs(ref.y).max())
            assert abs(res[b, 0] - ref.info.pri_res) < resid_tol and abs(res[b, 1] - ref.info.dua_res) < resid_tol
    return info, x, st, it


def test_scalar_known_answers(gpu):
    P = np.array([[2.0]]); A = np.array([[1.0]])
    probs = [(P, np.array([-4.0]), A, np.array([-np.inf]), np.array([np.inf])),
             (P, np.array([-4.0]), A, np.array([3.0]), np.array([5.0]))]
    info, x, st, it = _check(probs)
    assert np.allclose(x[:, 0], [2.0, 3.0], atol=1e-5) and info["n_core"] == 0    # the only variable is eliminated


@pytest.mark.parametrize("shape,B", [((3, 1, 1), 3), ((5, 3, 4), 8), ((6, 2, 3), 5)])
def test_penalty_qps_match_oracle(gpu, shape, B):
    rng = np.random.default_rng(100 + B)
    probs = [penalty_qp(rng, *shape) for _ in range(B)]
    info, *_ = _check(probs)
    T, d, r = shape
    assert info["n_elim"] == T * r and info["n_core"] == T * d


def test_row_multiplicities_match_physically_duplicated_rows(gpu):
    rng = np.random.default_rng(5)
    T, d, r = 5, 3, 4
    probs = [penalty_qp(rng, T, d, r) for _ in range(6)]
    m = len(probs[0][3])
    w = np.ones((6, m), dtype=np.int32)
    w[:, d:d + T * r] = rng.integers(1, 6, size=(6, 1))
    _check(probs, w=w)            # oracle runs with expand_dups=1: rows physically repeated


def test_7x20_shape_batch(gpu):
    rng = np.random.default_rng(9)
    probs = [penalty_qp(rng, 20, 7, 10) for _ in range(32)]
    m = len(probs[0][3])
    w = np.ones((32, m), dtype=np.int32); w[:, 7:7 + 200] = 2
    info, *_ = _check(probs, w=w, check=range(6))
    assert (info["n_elim"], info["n_core"]) == (200, 140)      # reduced SPD system of order n_x
    assert info["lds_admm"] <= 160 * 1024                       # fits the CU's LDS


def test_all_admm_kernel_tiers_agree(gpu, monkeypatch):
    """row-local kernel (default) vs register-offset vs LDS sliced-ELL vs generic kernel."""
    rng = np.random.default_rng(17)
    probs = [penalty_qp(rng, 6, 3, 4) for _ in range(4)] + [penalty_qp(rng, 6, 3, 4, )]
    _, x_rl, st_rl, it_rl = _check(probs)
    monkeypatch.setenv("SCO_QP_NO_RL", "1")
    _, x_reg, st_reg, it_reg = _check(probs)
    assert np.array_equal(st_rl, st_reg) and np.array_equal(it_rl, it_reg) and np.abs(x_rl - x_reg).max() < 1e-10
    monkeypatch.setenv("SCO_QP_NO_REG", "1")
    _, x_sell, st_sell, it_sell = _check(probs)
    monkeypatch.setenv("SCO_QP_NO_FAST", "1")
    _, x_gen, st_gen, it_gen = _check(probs)
    for x, st, it in ((x_sell, st_sell, it_sell), (x_gen, st_gen, it_gen)):
        assert np.array_equal(st, st_reg) and np.array_equal(it, it_reg) and np.abs(x - x_reg).max() < 1e-10


def test_wide_rows_fall_back_to_the_looping_kernels(gpu):
    """A pattern wider than the register-offset caps (row > 8 entries) still solves correctly."""
    rng = np.random.default_rng(23)
    n, m_c = 14, 6
    P = np.diag(rng.random(n) + 0.5)
    A = np.vstack([rng.standard_normal((m_c, n)), np.eye(n)])          # dense rows: 14 entries each
    probs = []
    for _ in range(3):
        l = np.concatenate([-rng.random(m_c) - 0.5, -np.ones(n)]); u = np.concatenate([rng.random(m_c) + 0.5, np.ones(n)])
        probs.append((P, rng.standard_normal(n), A, l, u))
    _check(probs, resid_tol=TOL)


def test_elimination_can_be_disabled_and_agrees(gpu, monkeypatch):
    rng = np.random.default_rng(11)
    probs = [penalty_qp(rng, 4, 2, 3) for _ in range(4)]
    monkeypatch.setenv("SCO_QP_NO_ELIM", "1")
    info, x0, *_ = _check(probs)
    assert info["n_elim"] == 0
    monkeypatch.delenv("SCO_QP_NO_ELIM")
    info, x1, *_ = _check(probs)
    assert info["n_elim"] > 0 and np.abs(x0 - x1).max() < 1e-9


def test_infeasible_and_unbounded_statuses(gpu):
    A = np.array([[1.0], [1.0]])
    prim = (np.array([[1.0]]), np.array([0.0]), A, np.array([1.0, -np.inf]), np.array([np.inf, 0.0]))
    _, _, st, _ = _check([prim])
    assert st[0] == -3
    dual = (np.array([[0.0]]), np.array([1.0]), np.array([[1.0]]), np.array([-np.inf]), np.array([0.0]))
    _, _, st, _ = _check([dual])
    assert st[0] == -4


def test_max_iter_and_settings_are_honoured(gpu):
    rng = np.random.default_rng(3)
    probs = [penalty_qp(rng, 5, 3, 4) for _ in range(3)]
    s = _lib.default_qp_settings(max_iter=40)
    _, _, st, it = _check(probs, settings=s, max_iter=40)
    assert np.all(st == -2) and np.all(it == 40)
    s = _lib.default_qp_settings(rho=0.5, sigma=1e-6, eps_abs=1e-5, eps_rel=1e-5, alpha=1.2, check_termination=10)
    _check(probs, settings=s, rho=0.5, sigma=1e-6, eps_abs=1e-5, eps_rel=1e-5, alpha=1.2, check_termination=10)


def test_set_bounds_changes_only_the_box(gpu):
    rng = np.random.default_rng(21)
    probs = [penalty_qp(rng, 5, 3, 4) for _ in range(2)]
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack(probs)
    qp = _lib.BatchedQP(2, n, m, Pp, Pi, Ap, Ai)
    qp.load(Pval, q, Aval, l, u)
    qp.solve()
    l2, u2 = l.copy(), u.copy()
    box = slice(m - n, m - n + 15)
    mid = 0.5 * (l[:, box] + u[:, box]); l2[:, box] = mid - 0.1; u2[:, box] = mid + 0.1     # shrunk trust region
    qp.set_bounds(l2, u2)
    x, y, st, it, _ = qp.solve()
    qp.close()
    for b in range(2):
        ref = o.solve(probs[b][0], probs[b][1], probs[b][2], l2[b], u2[b])
        assert (st[b], it[b]) == (ref.info.status_val, ref.info.iter) and np.abs(x[b] - ref.x).max() < TOL


def test_api_misuse_is_reported(gpu):
    Pp = np.array([0, 1], dtype=np.int32); Pi = np.array([0], dtype=np.int32)
    qp = _lib.BatchedQP(1, 1, 1, Pp, Pi, Pp, Pi)
    with pytest.raises(_lib.ScoHipError) as e:
        qp.solve()                                   # solve before load
    assert e.value.code == -4
    qp.close()
    bad = np.array([0, 2], dtype=np.int32)
    with pytest.raises(_lib.ScoHipError) as e:
        _lib.BatchedQP(1, 1, 1, Pp, Pi, bad, np.array([0, 0], dtype=np.int32))   # repeated row index
    assert e.value.code == -1


def test_results_are_run_to_run_deterministic(gpu):
    rng = np.random.default_rng(33)
    probs = [penalty_qp(rng, 6, 3, 4) for _ in range(16)]
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack(probs)
    outs = []
    for _ in range(2):
        qp = _lib.BatchedQP(16, n, m, Pp, Pi, Ap, Ai)
        qp.load(Pval, q, Aval, l, u)
        outs.append(qp.solve()); qp.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][3], outs[1][3])


@pytest.mark.parametrize("structured", [False, True])
def test_global_memory_tier_matches_oracle(gpu, monkeypatch, structured):
    """sco_qp_big.hip (forced): the tier BASELINE config 5 (12-DOF x 50) runs on, in its
    dense-inverse form and in the structured form (block-tridiagonal core, dense row blocks)."""
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    monkeypatch.setenv("SCO_QP_NO_BT", "0" if structured else "1")
    rng = np.random.default_rng(41)
    probs = [penalty_qp(rng, 6, 3, 4) for _ in range(3)]
    m = len(probs[0][3])
    w = np.ones((3, m), dtype=np.int32); w[:, 3:3 + 24] = 2
    info, x_big, st_big, it_big = _check(probs, w=w)
    assert (info["lds_admm"] > 0) == structured
    monkeypatch.delenv("SCO_QP_FORCE_BIG")
    _, x_rl, st_rl, it_rl = _check(probs, w=w)
    assert np.array_equal(st_big, st_rl) and np.array_equal(it_big, it_rl) and np.abs(x_big - x_rl).max() < 1e-10


@pytest.mark.parametrize("shape", [(5, 7, 40), (4, 12, 70), (3, 16, 20), (6, 2, 3)])
def test_structured_tier_dense_row_blocks(gpu, monkeypatch, shape):
    """Row blocks long enough to be addressed as dense chunks (>= 16 consecutive rows with the
    same core columns, here T blocks of r rows x d columns), block orders 8, 12, 16 and 4."""
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    T, d, r = shape
    rng = np.random.default_rng(100 + T + d + r)
    probs = [penalty_qp(rng, T, d, r) for _ in range(2)]
    info, x_bt, st_bt, it_bt = _check(probs)
    assert info["lds_admm"] > 0
    monkeypatch.setenv("SCO_QP_NO_BT", "1")
    _, x_d, st_d, it_d = _check(probs)
    assert np.array_equal(st_bt, st_d) and np.array_equal(it_bt, it_d) and np.abs(x_bt - x_d).max() < 1e-10


@pytest.mark.parametrize("structured", [False, True])
def test_global_memory_tier_infeasible_status(gpu, monkeypatch, structured):
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    monkeypatch.setenv("SCO_QP_NO_BT", "0" if structured else "1")
    A = np.array([[1.0], [1.0]])
    prim = (np.array([[1.0]]), np.array([0.0]), A, np.array([1.0, -np.inf]), np.array([np.inf, 0.0]))
    _, _, st, _ = _check([prim])
    assert st[0] == -3


def test_unconstrained_and_single_variable_qps(gpu):
    """m = 0 (no rows at all) and n = 1: the degenerate ends of the ABI."""
    P = np.array([[2.0, 0.5], [0.5, 1.0]]); q = np.array([-1.0, 1.0])
    Pu = sp.triu(sp.csc_matrix(P), format="csc"); Pu.sort_indices()
    Ap = np.zeros(3, dtype=np.int32); Ai = np.zeros(0, dtype=np.int32)
    qp = _lib.BatchedQP(1, 2, 0, Pu.indptr.astype(np.int32), Pu.indices.astype(np.int32), Ap, Ai)
    qp.load(Pu.data[None, :], q[None, :], np.zeros((1, 0)), np.zeros((1, 0)), np.zeros((1, 0)))
    x, y, st, it, _ = qp.solve()
    qp.close()
    assert st[0] == 1 and np.abs(x[0] - np.linalg.solve(P, -q)).max() < 1e-5
    one = (np.array([[4.0]]), np.array([-2.0]), np.array([[1.0]]), np.array([1.0]), np.array([1.0]))
    _, x1, st1, _ = _check([one])
    assert st1[0] == 1 and abs(x1[0, 0] - 1.0) < 1e-6


def test_structured_tier_is_run_to_run_deterministic(gpu, monkeypatch):
    """Partial sums are reduced in a fixed order inside the wavefront and chunks write disjoint data,
    so two runs give bit-identical answers."""
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    rng = np.random.default_rng(77)
    probs = [penalty_qp(rng, 4, 12, 70) for _ in range(3)]
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack(probs)
    outs = []
    for _ in range(2):
        qp = _lib.BatchedQP(3, n, m, Pp, Pi, Ap, Ai)
        qp.load(Pval, q, Aval, l, u)
        outs.append(qp.solve()); qp.close()
    assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    assert np.array_equal(outs[0][3], outs[1][3])


@pytest.mark.parametrize("tier", ["row-local", "structured"])
def test_warm_start_is_opt_in_and_converges_to_the_same_answer(gpu, monkeypatch, tier):
    """sco_qp_settings.warm_start (beyond parity: the reference always starts cold): re-solving from the
    previous solution needs a fraction of the iterations (not zero: z restarts at A x and the triple is
    only converged to the tolerances); a perturbed QP needs fewer iterations than from zero and agrees to
    the QP tolerances."""
    if tier == "structured":
        monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    rng = np.random.default_rng(11)
    probs = [penalty_qp(rng, 5, 3, 20 if tier == "structured" else 4) for _ in range(4)]   # 20 rows per block: dense chunks
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack(probs)
    qp = _lib.BatchedQP(4, n, m, Pp, Pi, Ap, Ai)
    qp.load(Pval, q, Aval, l, u)
    x0, y0, st0, it0, _ = qp.solve()
    xa, _, sta, ita, _ = qp.solve()                                   # default: cold again, identical
    assert np.array_equal(xa, x0) and np.array_equal(ita, it0)
    warm = _lib.default_qp_settings(warm_start=1)
    xw, _, stw, itw, _ = qp.solve(warm)
    assert np.all(stw == 1) and np.all(itw <= it0) and itw.sum() < 0.5 * it0.sum()
    assert np.abs(xw - x0).max() < 1e-5
    q2 = q + 0.05 * rng.standard_normal(q.shape)                      # a nearby QP: cold vs warm
    qp.load(Pval, q2, Aval, l, u)
    xc, _, stc, itc, _ = qp.solve()
    qp.load(Pval, q, Aval, l, u); qp.solve()                          # previous solution = the old QP's
    qp.load(Pval, q2, Aval, l, u)
    xw2, _, stw2, itw2, _ = qp.solve(warm)
    qp.close()
    assert np.array_equal(stc, stw2) and np.abs(xw2 - xc).max() < 1e-4
    assert itw2.sum() < itc.sum()


def _random_qp(rng, n, m, density):
    """Feasible random QP with an arbitrary sparsity pattern (not penalty shaped): P = sparse PSD + diagonal
    on some variables only (so that some variables can be eliminated), bounds around A x0."""
    M = sp.random(n, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 30))).toarray()
    P = M @ M.T
    keep = rng.random(n) < 0.6
    P[~keep, :] = 0.0; P[:, ~keep] = 0.0                      # variables without any P entry
    P += np.diag(np.where(keep, rng.uniform(0.1, 1.0, n), 0.0))
    A = sp.random(m, n, density=density, random_state=np.random.RandomState(rng.integers(1 << 30))).toarray()
    A = np.vstack([A, np.eye(n)])                             # a bound row per variable keeps the QP bounded
    x0 = rng.standard_normal(n)
    ax = A @ x0
    lo = ax - rng.uniform(0.0, 1.0, A.shape[0]); hi = ax + rng.uniform(0.0, 1.0, A.shape[0])
    eq = rng.random(A.shape[0]) < 0.15
    lo[eq] = hi[eq] = ax[eq]
    free = rng.random(A.shape[0]) < 0.1
    free[-n:] = False
    lo[free] = -np.inf; hi[free] = np.inf
    return P, rng.standard_normal(n), A, lo, hi


@pytest.mark.parametrize("seed", range(10))
def test_random_sparsity_patterns_match_oracle(gpu, seed):
    """Arbitrary patterns exercise the symbolic analysis (independent set, coupling pairs, Schur plans)
    and whichever tier the pattern lands on.  Variables without any P entry make the oracle's KKT
    matrix rely on sigma = 5e-10 alone (osqp_utils.py:11), which costs it ~1e-7 of accuracy in x; the
    reduced system the device solves does not have that diagonal, so the residuals are compared to the
    parity tolerance here, not to 1e-9 (status, iteration count, x and y still have to agree)."""
    rng = np.random.default_rng(500 + seed)
    n, m = int(rng.integers(4, 40)), int(rng.integers(0, 40))
    base = _random_qp(rng, n, m, float(rng.uniform(0.05, 0.4)))
    probs = [base]
    for _ in range(2):                                         # same pattern, other values
        P, q, A, lo, hi = base
        s = rng.uniform(0.5, 1.5)
        probs.append((P * s, q + 0.1 * rng.standard_normal(n), A * np.where(A != 0, rng.uniform(0.5, 1.5, A.shape), 0.0), lo, hi))
    probs[1] = (probs[1][0], probs[1][1], probs[1][2],
                np.minimum(probs[1][3], (probs[1][2] @ np.zeros(n)) - 0.0), np.maximum(probs[1][4], 0.0))   # 0 feasible
    _check(probs, resid_tol=TOL)


@pytest.mark.parametrize("seed", range(4))
def test_random_infeasible_qps_get_the_oracles_status(gpu, seed):
    """Primal infeasible (two contradictory equality rows) and dual infeasible (a free direction with a
    linear cost and no curvature) variants of random QPs: certificates are detected at the same check."""
    rng = np.random.default_rng(900 + seed)
    n, m = int(rng.integers(5, 25)), int(rng.integers(3, 20))
    P, q, A, lo, hi = _random_qp(rng, n, m, 0.3)
    a = rng.standard_normal(n)
    A1 = np.vstack([a, a, A]); lo1 = np.concatenate([[1.0, -1.0], lo]); hi1 = np.concatenate([[1.0, -1.0], hi])
    _, _, st, _ = _check([(P, q, A1, lo1, hi1)], resid_tol=np.inf)
    assert st[0] in (-3, 3)
    j = int(np.argmin(np.abs(P).sum(axis=0) + 1e9 * (np.abs(P).sum(axis=0) > 0)))      # a variable without curvature, if any
    P2 = P.copy(); P2[j, :] = 0.0; P2[:, j] = 0.0
    A2 = A.copy(); lo2, hi2 = lo.copy(), hi.copy()
    rows = np.nonzero(A2[:, j])[0]
    lo2[rows] = -np.inf; hi2[rows] = np.inf                                            # nothing bounds x_j any more
    q2 = q.copy(); q2[j] = 1.0
    _, _, st, _ = _check([(P2, q2, A2, lo2, hi2)], resid_tol=np.inf)
    assert st[0] in (-4, 4)


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("no_bt", ["0", "1"])
def test_random_sparsity_patterns_through_the_global_memory_tier(gpu, monkeypatch, seed, no_bt):
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    monkeypatch.setenv("SCO_QP_NO_BT", no_bt)
    rng = np.random.default_rng(700 + seed)
    n, m = int(rng.integers(4, 40)), int(rng.integers(0, 40))
    _check([_random_qp(rng, n, m, float(rng.uniform(0.05, 0.4)))], resid_tol=TOL)


def test_a_qp_that_needs_the_global_memory_tier_by_itself(gpu):
    """No switches: 300 variables with a dense-ish core exceed the LDS tiers and land on the dense form of
    the global-memory tier (the core is not banded)."""
    rng = np.random.default_rng(31)
    prob = _random_qp(rng, 300, 150, 0.02)
    info, x, st, it = _check([prob], resid_tol=TOL)
    assert info["n_core"] > 256 and info["lds_admm"] == 0


@pytest.mark.parametrize("tier", ["row-local", "generic", "register", "sliced-ELL", "structured"])
def test_adaptive_rho_matches_the_oracle_rule(gpu, monkeypatch, tier):
    """adaptive_rho=True (solver.py:39 / osqp_utils.py:13; off in the reference's defaults): OSQP's rho update
    every 4 x check_termination iterations.  The solve is parked at each update point, rho re-estimated from the
    scaled iterates, the reduced system refactored and the solve resumed -- same statuses and iteration counts as
    the oracle with the same rule."""
    if tier == "structured":
        monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    elif tier != "row-local":
        monkeypatch.setenv("SCO_QP_NO_RL", "1")
    if tier == "generic":
        monkeypatch.setenv("SCO_QP_NO_REG", "1"); monkeypatch.setenv("SCO_QP_NO_FAST", "1")
    if tier == "sliced-ELL":
        monkeypatch.setenv("SCO_QP_NO_REG", "1")       # "register": the register-offset kernel parks since r03
    rng = np.random.default_rng(23)
    r = 20 if tier == "structured" else 4                            # 20 rows per block: dense row chunks
    probs = [penalty_qp(rng, 6, 3, r) for _ in range(6)]
    m = len(probs[0][3])
    w = np.ones((6, m), dtype=np.int32); w[:, 3:3 + 6 * r] = 2
    st = _lib.default_qp_settings(adaptive_rho=1)
    info, x_ad, st_ad, it_ad = _check(probs, w=w, settings=st, adaptive_rho=1, expand_dups=0, resid_tol=1e-7)
    if tier == "register":
        assert info["lds_admm"] > 44032                 # static LDS of qp_admm_reg_kernel + its value images
    _, x_fx, st_fx, it_fx = _check(probs, w=w, expand_dups=0)
    assert not np.array_equal(it_ad, it_fx)                          # some of them did change rho
    ok = (st_ad == 1) & (st_fx == 1)
    assert ok.any() and np.abs(x_ad[ok] - x_fx[ok]).max() < 1e-4


def test_adaptive_rho_interval_and_tolerance_are_honoured(gpu):
    rng = np.random.default_rng(29)
    probs = [penalty_qp(rng, 20, 7, 10) for _ in range(4)]
    _, _, _, it_fx = _check(probs)
    for kw in (dict(), dict(adaptive_rho_interval=50), dict(adaptive_rho_interval=250, adaptive_rho_tolerance=2.0)):
        st = _lib.default_qp_settings(adaptive_rho=1, **kw)
        _, _, _, it_ad = _check(probs, settings=st, adaptive_rho=1, resid_tol=1e-7, **kw)
        assert it_ad.sum() < 0.6 * it_fx.sum()                       # 1600 against 3950 with the default interval


def test_adaptive_rho_is_refused_on_the_dense_global_memory_form(gpu, monkeypatch):
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1"); monkeypatch.setenv("SCO_QP_NO_BT", "1")
    rng = np.random.default_rng(31)
    probs = [penalty_qp(rng, 5, 3, 4) for _ in range(2)]
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack(probs)
    qp = _lib.BatchedQP(2, n, m, Pp, Pi, Ap, Ai)
    try:
        qp.load(Pval, q, Aval, l, u)
        with pytest.raises(_lib.ScoHipError) as e:
            qp.solve(_lib.default_qp_settings(adaptive_rho=1))
        assert e.value.code == -5 and "adaptive_rho" in str(e.value)
        x, _, st, _, _ = qp.solve()                   # the handle is still usable
        assert np.all(st == 1)
    finally:
        qp.close()


@pytest.mark.parametrize("seed", range(8))
@pytest.mark.parametrize("tier", ["on-chip", "structured"])
def test_adaptive_rho_on_random_sparsity_patterns(gpu, monkeypatch, seed, tier):
    """Random patterns (not penalty shaped) with adaptive rho: whichever kernel the pattern lands on parks and
    resumes around rho changes; status and iteration count follow the oracle with the same rule.  A rho decision
    within rounding of its threshold may fall differently on the oracle's KKT route, so a pattern whose counts
    differ must at least reach the same status and the same answer."""
    if tier == "structured":
        monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    rng = np.random.default_rng(1300 + seed)
    n, m = int(rng.integers(4, 40)), int(rng.integers(0, 40))
    prob = _random_qp(rng, n, m, float(rng.uniform(0.05, 0.4)))
    npat, m_all, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack([prob])
    qp = _lib.BatchedQP(1, npat, m_all, Pp, Pi, Ap, Ai)
    try:
        qp.load(Pval, q, Aval, l, u)
        try:
            x, y, st, it, res = qp.solve(_lib.default_qp_settings(adaptive_rho=1))
        except _lib.ScoHipError as e:
            assert tier == "structured" and e.code == -5          # not banded: the dense form cannot park
            return
        rho, nupd = qp.adaptive_info()
    finally:
        qp.close()
    ref = o.solve(*prob, adaptive_rho=1)
    assert st[0] == ref.info.status_val
    if it[0] == ref.info.iter:
        assert nupd[0] == ref.info.rho_updates
    assert np.abs(x[0] - ref.x).max() < 1e-4 * (1 + np.abs(ref.x).max())


def test_both_core_inversion_routes_agree(gpu, monkeypatch):
    """W = S^-1 by register-resident Gauss-Jordan sweeps (default) and by Cholesky + triangular inverse
    (SCO_QP_FACTOR_CHOLESKY=1): same statuses and iteration counts, answers within 1e-10; cores of order 3, 15, 60,
    140 (one tile per thread) and 200 (two tiles per thread)."""
    rng = np.random.default_rng(41)
    for shape in ((1, 3, 2), (5, 3, 4), (20, 3, 2), (20, 7, 10), (25, 8, 3)):
        probs = [penalty_qp(rng, *shape) for _ in range(2)]
        monkeypatch.delenv("SCO_QP_FACTOR_CHOLESKY", raising=False)
        info, x_s, st_s, it_s = _check(probs, check=[0])
        assert info["n_core"] == shape[0] * shape[1]
        monkeypatch.setenv("SCO_QP_FACTOR_CHOLESKY", "1")
        _, x_c, st_c, it_c = _check(probs, check=[])
        assert np.array_equal(st_s, st_c) and np.array_equal(it_s, it_c) and np.abs(x_s - x_c).max() < 1e-10


def test_null_index_array_with_a_non_empty_pattern_is_rejected(gpu):
    import ctypes as C
    lib = _lib.load()
    h = C.c_void_p()
    Pp = np.array([0, 1], dtype=np.int32); Ap = np.array([0, 1], dtype=np.int32); Ai = np.array([0], dtype=np.int32)
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    rc = lib.sco_qp_create(0, 1, 1, 1, ip(Pp), None, ip(Ap), ip(Ai), C.byref(h))
    assert rc == -1 and not h.value
    rc = lib.sco_qp_create(0, 1, 1, 1, ip(Pp), ip(Ai), ip(Ap), None, C.byref(h))
    assert rc == -1 and not h.value


@pytest.mark.parametrize("k", [1, 2])
def test_device_admm_against_the_admm_free_solutions_of_the_golden_qps(gpu, k):
    """The device ADMM on the golden 7-DOF x 20 penalty QPs against tests/golden/qp_exact_7x20.npz (interior point +
    active-set polish, oracle/qp_exact.py): the converging QP lands on the optimum, the compounded-penalty QP (Q1) is
    still far from it when max_iter = 100 000 ends the solve -- in both cases with the iteration count of the golden
    reference run (the duplicated rows of the second QP are passed as physical rows here, as the reference does)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    from make_qp_exact import golden_qp
    from oracle import qp_exact as qe
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "trajopt_7x20.npz")); ex = np.load(os.path.join(gold, "qp_exact_7x20.npz"))
    pre = "p0_qp%d" % k
    P, q, A, l, u = golden_qp(g, pre)
    st = _lib.default_qp_settings(rho=0.1, sigma=5e-10, eps_abs=1e-6, eps_rel=1e-9, max_iter=100000)
    n, m, Pp, Pi, Ap, Ai, Pval, qq, Aval, ll, uu = _stack([(P, q, A, l, u)])
    qp = _lib.BatchedQP(1, n, m, Pp, Pi, Ap, Ai)
    try:
        qp.load(Pval, qq, Aval, ll, uu)
        x, y, status, iters, res = qp.solve(st)
    finally:
        qp.close()
    assert iters[0] == int(g[pre + "_iters"]) and status[0] == int(g[pre + "_status"])
    xs, obj = ex[pre + "_x"], float(ex[pre + "_obj"])
    mine = 0.5 * x[0] @ P @ x[0] + q @ x[0]
    if k == 1:
        assert status[0] == 1 and np.abs(x[0] - xs).max() < 2e-4 and abs(mine - obj) < 1e-6 * (1 + abs(obj))
    else:
        r = qe.osqp_residuals(P, q, A, l, u, x[0], y[0])
        assert status[0] == -2 and np.abs(x[0] - xs).max() > 0.1 and mine > obj + 1.0
        assert r["dual"] > 1e-4 or r["primal_lower_bound"] > 1e-4


def test_closed_thread_assignment_gives_the_same_answer(gpu, monkeypatch):
    """SCO_QP_RL_CLOSED=1 (opt-in, slower on the trajectory QPs: DESIGN.md 3.2): every core column in the wavefront of all
    its rows, phase (1) without a workgroup barrier -- same status, iteration count and solution as the default plan."""
    rng = np.random.default_rng(17)
    probs = [penalty_qp(rng, 20, 7, 10) for _ in range(3)]
    base = _check(probs)
    monkeypatch.setenv("SCO_QP_RL_CLOSED", "1")
    closed = _check(probs)
    assert np.array_equal(base[2], closed[2]) and np.array_equal(base[3], closed[3])
    assert np.abs(base[1] - closed[1]).max() < 1e-10


@pytest.mark.parametrize("switch", ["SCO_QP_RL_ALIGNED", "SCO_QP_RL_LAY8"])
def test_opt_in_layouts_of_the_row_local_kernel_give_the_same_answer(gpu, monkeypatch, switch):
    """r03 experiments kept as opt-ins (both measured not faster, profiles/r03_ab.txt): SCO_QP_RL_ALIGNED=1 -- every
    wavefront owns whole timesteps (rows, columns and the W rows of those columns), one barrier per iteration, double-
    buffered right-hand side; SCO_QP_RL_LAY8=1 -- 3 x 18 W tiles with 8 column groups under the open assignment.  Same
    status and iteration count as the default plan and as the oracle, solutions to 1e-10 (the W mat-vec sums in a different
    order).  Parked / resumed solves of these layouts: test_sqp_gpu.py::test_opt_in_layouts_through_the_sliced_sqp_loop."""
    rng = np.random.default_rng(23)
    probs = [penalty_qp(rng, 20, 7, 10) for _ in range(3)]
    base = _check(probs)
    monkeypatch.setenv(switch, "1")
    alt = _check(probs)
    assert np.array_equal(base[2], alt[2]) and np.array_equal(base[3], alt[3])
    assert np.abs(base[1] - alt[1]).max() < 1e-10
    # the plan really is the opt-in one
    import ctypes as C
    n, m, Pp, Pi, Ap, Ai = _stack(probs[:1])[:6]
    lib = _lib.load()
    ip = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))
    sizes = np.zeros(16, dtype=np.int32); info = np.zeros(8, dtype=np.int32)
    lib.sco_debug_plan_build.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.c_int, C.POINTER(C.c_int)]
    lib.sco_debug_rl_plan.argtypes = [C.POINTER(C.c_int)]
    assert lib.sco_debug_plan_build(n, m, ip(Pp), ip(Pi), ip(Ap), ip(Ai), 1, ip(sizes)) == 0
    assert lib.sco_debug_rl_plan(ip(info)) == 0
    assert info[0] == 1 and info[6] == (1 if switch.endswith("ALIGNED") else 2) and info[2] == 3 and info[3] == 18


def test_split_and_unsplit_column_plans_give_the_same_answer(gpu, monkeypatch):
    """r03 default: a core column's operand pairs sit in two neighbouring lanes (owner + helper) and their partial sums are
    joined by one quad_perm step -- phase (1) and the test's A'y read half as many operands per lane.  SCO_QP_RL_SPLIT=0 keeps
    one lane per column (the r02 plan): same status and iteration count, solutions to 1e-10 (another order of the column sums),
    for the 7x20 shape, a wide pattern (8 pairs per column) and small shapes; both agree with the oracle."""
    rng = np.random.default_rng(29)
    for probs in ([penalty_qp(rng, 20, 7, 10) for _ in range(3)], [penalty_qp(rng, 5, 3, 4) for _ in range(2)],
                  [penalty_qp(rng, 12, 4, 14) for _ in range(2)]):
        monkeypatch.delenv("SCO_QP_RL_SPLIT", raising=False)
        base = _check(probs)
        monkeypatch.setenv("SCO_QP_RL_SPLIT", "0")
        alt = _check(probs)
        assert np.array_equal(base[2], alt[2]) and np.array_equal(base[3], alt[3])
        assert np.abs(base[1] - alt[1]).max() < 1e-10


def _tiers(n, m, Pp, Pi, Ap, Ai):
    import ctypes as C
    lib = _lib.load()
    lib.sco_debug_qp_tiers.restype = C.c_int; lib.sco_debug_qp_tiers.argtypes = [C.c_void_p]
    qp = _lib.BatchedQP(1, n, m, Pp, Pi, Ap, Ai)
    try:
        return lib.sco_debug_qp_tiers(qp._h)
    finally:
        qp.close()


@pytest.mark.parametrize("shape", [(6, 3, 4), (8, 2, 3), (9, 5, 6), (20, 7, 10), (12, 7, 10), (20, 2, 4), (16, 5, 8)])
def test_wavefront_tier_agrees_with_the_row_local_tier_and_the_oracle(gpu, monkeypatch, shape):
    """r04: csrc/sco_admm_wv.hip -- one wavefront per problem, twisted block-tridiagonal core solve instead of the dense
    inverse.  Same statuses and iteration counts as the row-local kernel and as the oracle, answers within 1e-10 of the
    former; row weights, second pins, every instantiation (<7,4,3,10> at 7 x 20, <8,1,1,4>, <8,2,2,8> at 7 x 12 and 5 x 16,
    <8,2,2,10> at 2 x 20)."""
    monkeypatch.setenv("SCO_WV_MIN_PER_CU", "0")      # (by default a launch goes to this tier with > 3.3 problems per CU only)
    T, d, r = shape
    rng = np.random.default_rng(100 + T)
    probs = [penalty_qp(rng, T, d, r) for _ in range(6)]
    m = len(probs[0][3])
    w = np.ones((6, m), dtype=np.int32); w[:, d:d + T * r] = rng.integers(1, 4, size=(6, 1))
    n, m, Pp, Pi, Ap, Ai, *_ = _stack(probs)
    assert _tiers(n, m, Pp, Pi, Ap, Ai) & 32, "pattern did not land on the wavefront tier"
    _, x_wv, st_wv, it_wv = _check(probs, w=w, check=range(3))
    monkeypatch.setenv("SCO_QP_NO_WV", "1")
    assert not _tiers(n, m, Pp, Pi, Ap, Ai) & 32
    _, x_rl, st_rl, it_rl = _check(probs, w=w, check=[])
    assert np.array_equal(st_wv, st_rl) and np.array_equal(it_wv, it_rl) and np.abs(x_wv - x_rl).max() < 1e-10


def test_wavefront_tier_leaves_odd_value_structure_to_the_row_local_kernel(gpu, monkeypatch):
    """The tier's kernel assumes the VALUES of a penalty QP (hinge rows l = -inf with one weight, slack rows [0, inf), box
    rows on the base rho).  Problems of a batch that do not have them -- a two-sided hinge row, a slack with an upper bound,
    a box so narrow that OSQP gives it the equality rho, unequal weights -- are found by qp_wv_factor_kernel and solved by
    the row-local kernel in the same launch sequence: statuses, iteration counts and answers of the whole batch = the
    oracle's and = a run with the tier switched off."""
    monkeypatch.setenv("SCO_WV_MIN_PER_CU", "0")
    rng = np.random.default_rng(77)
    T, d, r = 6, 3, 4
    probs = [list(penalty_qp(rng, T, d, r)) for _ in range(6)]
    nx = T * d; m = len(probs[0][3])
    probs[1][3] = probs[1][3].copy(); probs[1][3][d + 2] = -3.0                     # hinge row with a finite lower bound
    probs[2][4] = probs[2][4].copy(); probs[2][4][d + T * r + nx + 1] = 5.0         # slack row with an upper bound
    lo, hi = probs[3][3].copy(), probs[3][4].copy()
    mid = 0.5 * (lo[d + T * r + 4] + hi[d + T * r + 4]); lo[d + T * r + 4] = mid - 2e-5; hi[d + T * r + 4] = mid + 2e-5
    probs[3][3], probs[3][4] = lo, hi                                                # box narrower than 1e-4: rho_eq
    probs = [tuple(p) for p in probs]
    w = np.ones((6, m), dtype=np.int32); w[:, d:d + T * r] = 2; w[4, d + 1] = 3       # problem 4: unequal hinge weights
    _, x_wv, st_wv, it_wv = _check(probs, w=w)
    monkeypatch.setenv("SCO_QP_NO_WV", "1")
    _, x_rl, st_rl, it_rl = _check(probs, w=w, check=[])
    assert np.array_equal(st_wv, st_rl) and np.array_equal(it_wv, it_rl) and np.abs(x_wv - x_rl).max() < 1e-10


@pytest.mark.parametrize("shape", [(6, 3, 4), (20, 7, 10)])
def test_wavefront_tier_infeasibility_certificates(gpu, monkeypatch, shape):
    """r04: the two infeasibility certificates of the wavefront tier run on its structured layout (A' (w dy), P dx and A dx
    from the lane's rows and the block vectors in LDS, no generic loop over the pattern).  A pin outside its variable's box
    (primal infeasible) and a cost that pushes every joint down a direction P does not see with the boxes open below (dual
    infeasible), between ordinary problems: statuses and iteration counts = the oracle's and = the row-local kernel's."""
    monkeypatch.setenv("SCO_WV_MIN_PER_CU", "0")
    T, d, r = shape
    rng = np.random.default_rng(300 + T)
    probs = [list(penalty_qp(rng, T, d, r)) for _ in range(6)]
    nx = T * d; box0 = d + T * r
    for k in (2, 3):                                            # pin of joint k - 2 four above the box of its variable
        lo, hi = probs[k][3].copy(), probs[k][4].copy()
        lo[k - 2] = hi[k - 2] = hi[box0 + k - 2] + 4.0
        probs[k][3], probs[k][4] = lo, hi
    for k in (4, 5):
        q, lo, hi = probs[k][1].copy(), probs[k][3].copy(), probs[k][4].copy()
        lo[:d] = -np.inf; hi[:d] = np.inf                        # no pins
        lo[box0:box0 + nx] = -np.inf                             # boxes open below
        q[:nx] = 1000.0
        probs[k][1], probs[k][3], probs[k][4] = q, lo, hi
    probs = [tuple(p) for p in probs]
    n, m, Pp, Pi, Ap, Ai, *_ = _stack(probs)
    assert _tiers(n, m, Pp, Pi, Ap, Ai) & 32
    # (one 7 x 20 case creeps down its ray until max_iter, in the oracle too: status and count are compared, its x of order 1e5 not)
    _, x_wv, st_wv, it_wv = _check(probs, check=[0, 1, 2, 3, 5])
    ref4 = o.solve(*probs[4])
    assert (st_wv[4], it_wv[4]) == (ref4.info.status_val, ref4.info.iter)
    assert list(st_wv[2:4]) == [-3, -3] and -4 in st_wv[4:6] and list(st_wv[:2]) == [1, 1], st_wv
    monkeypatch.setenv("SCO_QP_NO_WV", "1")
    _, x_rl, st_rl, it_rl = _check(probs, check=[])
    assert np.array_equal(st_wv, st_rl) and np.array_equal(it_wv, it_rl)
    assert np.abs(x_wv[:2] - x_rl[:2]).max() < 1e-10


def test_wavefront_tier_time_slices_and_max_iter(gpu, monkeypatch):
    """Parked and resumed solves (sco_qp_settings.max_iter cut into slices by the SQP loop is covered in test_sqp_gpu.py);
    here: a solve that stops on max_iter between two termination checks, and one with check_termination off."""
    monkeypatch.setenv("SCO_WV_MIN_PER_CU", "0")
    rng = np.random.default_rng(5)
    probs = [penalty_qp(rng, 20, 7, 10) for _ in range(4)]
    for kw in (dict(max_iter=60), dict(max_iter=333), dict(max_iter=500, check_termination=0)):
        st = _lib.default_qp_settings(**kw)
        _check(probs, settings=st, **kw)


@pytest.mark.parametrize("shape", [(6, 12, 20), (5, 12, 9), (4, 16, 33)])
def test_block_normal_matrices_on_the_matrix_cores_change_no_bit(gpu, monkeypatch, shape):
    """r04 (north_star: "MFMA used only for the dense batched Jacobian x step contraction"): on the structured global-memory
    tier the hinge-row part J' R J of every diagonal block of S is formed by v_mfma_f64_16x16x4 for blocks of order 12 .. 16
    (BASELINE configs[4] is 100 x 12 per block), started from the terms the vector path adds first and followed by the ones it
    adds afterwards, the Schur terms included -- the same additions in the same order.  Against SCO_QP_NO_MFMA=1: identical
    bits in x and y, identical iteration counts; row counts that are no multiple of four, row weights."""
    import ctypes as C
    T, d, r = shape
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    rng = np.random.default_rng(31 + d)
    probs = [penalty_qp(rng, T, d, r) for _ in range(3)]
    m = len(probs[0][3])
    w = np.ones((3, m), dtype=np.int32); w[:, d:d + T * r] = rng.integers(1, 4, size=(3, 1))
    n, m, Pp, Pi, Ap, Ai, Pval, q, Aval, l, u = _stack(probs)
    outs = []
    for no in ("0", "1"):
        monkeypatch.setenv("SCO_QP_NO_MFMA", no)
        assert bool(_tiers(n, m, Pp, Pi, Ap, Ai) & 64) == (no == "0")
        qp = _lib.BatchedQP(3, n, m, Pp, Pi, Ap, Ai)
        try:
            qp.load(Pval, q, Aval, l, u, w)
            outs.append(qp.solve())
        finally:
            qp.close()
    (x0, y0, s0, i0, _), (x1, y1, s1, i1, _) = outs
    assert np.array_equal(x0, x1) and np.array_equal(y0, y1) and np.array_equal(i0, i1) and np.array_equal(s0, s1)
    ref = o.solve(*probs[0], w=w[0])
    assert s0[0] == ref.info.status_val and i0[0] == ref.info.iter and np.abs(x0[0] - ref.x).max() < TOL
