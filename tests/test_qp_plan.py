"""Host-side symbolic analysis (csrc/qp_plan.cpp) checked without a GPU: the real
C++ plans drive a NumPy rendition of the device phases (tests/qp_emulator.py) and
the result must equal the oracle's."""
import ctypes as C

import numpy as np
import pytest
import scipy.sparse as sp

import qp_emulator as E
from oracle import osqp_ref as o
from sco_py_amd import _lib


def penalty_qp(rng, T, d, r, feasible_pins=True):
    nx = T * d; ns = T * r; n = nx + ns
    Q = np.zeros((n, n))
    for t in range(T - 1):
        for j in range(d):
            a, b = t * d + j, (t + 1) * d + j
            Q[a, a] += 2; Q[b, b] += 2; Q[a, b] -= 2; Q[b, a] -= 2
    x0 = rng.standard_normal(nx) * 0.3
    rows, lo, hi = [], [], []
    for j in range(d):
        e = np.zeros(n); e[j] = 1; rows.append(e)
        v = x0[j] + 0.1 * rng.standard_normal(); lo.append(v); hi.append(v)
    for t in range(T):
        for k in range(r):
            e = np.zeros(n); e[t * d:(t + 1) * d] = rng.standard_normal(d); e[nx + t * r + k] = -1
            rows.append(e); lo.append(-np.inf); hi.append(rng.standard_normal())
    for j in range(n):
        e = np.zeros(n); e[j] = 1; rows.append(e)
        if j < nx:
            lo.append(x0[j] - 1); hi.append(x0[j] + 1)
        else:
            lo.append(0.0); hi.append(np.inf)
    q = np.zeros(n); q[nx:] = 10.0
    return Q, q, np.array(rows), np.array(lo), np.array(hi)


def _patterns(P, A):
    Pu = sp.triu(sp.csc_matrix(P), format="csc"); Pu.sort_indices()
    Ac = sp.csc_matrix(A); Ac.sort_indices()
    return (Pu.indptr.astype(np.int32), Pu.indices.astype(np.int32), Pu.data,
            Ac.indptr.astype(np.int32), Ac.indices.astype(np.int32), Ac.data)


@pytest.mark.parametrize("shape", [(5, 3, 4), (4, 2, 3), (3, 1, 1)])
@pytest.mark.parametrize("elim", [1, 0])
def test_plans_reproduce_the_oracle(shape, elim):
    lib = _lib.load()
    rng = np.random.default_rng(sum(shape))
    T, d, r = shape
    P, q, A, l, u = penalty_qp(rng, T, d, r)
    n, m = len(q), len(l)
    Pp, Pi, Pv, Ap, Ai, Av = _patterns(P, A)
    w = np.ones(m, dtype=np.int32); w[d:d + T * r] = 3
    ref = o.solve(P, q, A, l, u, w=w)
    plan = E.get_plan(lib, n, m, Pp, Pi, Ap, Ai, elim)
    if elim:
        assert plan["n_e"] == T * r and plan["n_c"] == T * d      # slacks eliminated, x in the core
        assert plan["ncpl"] == T * r * d
    else:
        assert plan["n_e"] == 0 and plan["n_c"] == n
    x, y, st, it = E.emulate(plan, Pv, q, Av, l, u, w=w)
    assert (st, it) == (ref.info.status_val, ref.info.iter)
    assert np.abs(x - ref.x).max() < 1e-11 and np.abs(y - ref.y).max() < 1e-9


def test_eliminated_set_is_independent_and_core_is_complete():
    lib = _lib.load()
    rng = np.random.default_rng(7)
    # abs-penalty shaped rows: p and n slacks share a row, so only one of each pair may be eliminated
    n_x, r = 4, 3
    n = n_x + 2 * r
    rows = []
    for i in range(r):
        e = np.zeros(n); e[:n_x] = rng.standard_normal(n_x); e[n_x + i] = -1; e[n_x + r + i] = 1
        rows.append(e)
    A = np.vstack([np.array(rows), np.eye(n)])
    P = np.zeros((n, n)); P[:n_x, :n_x] = np.eye(n_x) + 0.1
    Pp, Pi, Pv, Ap, Ai, Av = _patterns(P, A)
    plan = E.get_plan(lib, n, A.shape[0], Pp, Pi, Ap, Ai, 1)
    assert plan["n_e"] == r and plan["n_c"] == n - r
    elim = set(plan["elim_var"].tolist())
    for i in range(r):
        assert len(elim & {n_x + i, n_x + r + i}) == 1
    # solving with this plan still matches the oracle
    q = np.concatenate([rng.standard_normal(n_x), np.full(2 * r, 5.0)])
    l = np.concatenate([np.ones(r), np.full(n_x, -2.0), np.zeros(2 * r)])
    u = np.concatenate([np.ones(r), np.full(n_x, 2.0), np.full(2 * r, np.inf)])
    ref = o.solve(P, q, A, l, u)
    x, y, st, it = E.emulate(plan, Pv, q, Av, l, u)
    assert (st, it) == (ref.info.status_val, ref.info.iter) and np.abs(x - ref.x).max() < 1e-10


def test_malformed_patterns_are_rejected():
    lib = _lib.load()
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    sizes = np.zeros(16, dtype=np.int32)
    Pp = np.array([0, 1, 2], dtype=np.int32); Pi = np.array([0, 1], dtype=np.int32)
    Ap = np.array([0, 2, 3], dtype=np.int32)
    bad_rows = np.array([1, 0, 0], dtype=np.int32)           # not increasing inside column 0
    assert lib.sco_debug_plan_build(2, 2, ip(Pp), ip(Pi), ip(Ap), ip(bad_rows), 1, ip(sizes)) != 0
    lower = np.array([1, 1], dtype=np.int32)                  # P entry below the diagonal
    ok_rows = np.array([0, 1, 0], dtype=np.int32)
    assert lib.sco_debug_plan_build(2, 2, ip(Pp), ip(lower), ip(Ap), ip(ok_rows), 1, ip(sizes)) != 0
    assert lib.sco_debug_plan_build(2, 2, ip(Pp), ip(Pi), ip(Ap), ip(ok_rows), 1, ip(sizes)) == 0


@pytest.mark.parametrize("name,kw,cw", [
    ("circles", {}, 12), ("objective terms", dict(ee_cost_weight=1.0), 12),
    ("reach", dict(reach=True), 16), ("velocity limits", dict(vel_limit=0.3), 12),
    ("joint limits", dict(joint_limit=0.2), 12),
    ("velocity + joint limits", dict(vel_limit=0.3, joint_limit=0.2), 12),
    ("reach + velocity limits", dict(reach=True, vel_limit=0.3), 16)])
def test_which_device_families_land_on_the_row_local_tier(name, kw, cw):
    """Host-side plan of the fastest ADMM tier (csrc/sco_admm_rl.hip: rl_plan_build) for the penalty QP of every device
    family at 7-DOF x 20: gather-dots take operands in aligned pairs; CW = 12: a column thread <= 6 pairs, a row <= 4, two
    row slots per thread; 16 (wide): 8 / 5; 20: the three-slot instantiation (10 / 5, up to 1536 rows) that velocity + joint
    limits (1100 rows) needed until r03.  r03: a column's pairs are split over two neighbouring lanes (owner + helper), so the 8 pairs
    of a column with velocity or joint limits are 4 + 4 and fit the default instantiation; reach keeps the wide one for its
    5-pair rows, reach + velocity limits (9 pairs in a last-timestep column: 5 + 4) no longer needs three row slots, and
    velocity + joint limits (1100 rows: three row slots) runs the three-slot kernel with the narrow offsets (5 + 5 pairs)."""
    fits = True
    from oracle import arm_family as af
    from oracle import sco_ref as sr
    out = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(0, **kw)), sr.SolverParams(max_qp_solves=2), record_qps=True)
    q = out.qps[1]
    P = sp.triu(sp.csc_matrix(q["P"] != 0), format="csc"); A = sp.csc_matrix(q["A"] != 0)
    P.sort_indices(); A.sort_indices()
    lib = _lib.load()
    ip = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))
    Pp, Pi, Ap, Ai = (np.ascontiguousarray(a, dtype=np.int32) for a in (P.indptr, P.indices, A.indptr, A.indices))
    sizes = np.zeros(16, dtype=np.int32); info = np.zeros(8, dtype=np.int32)
    lib.sco_debug_plan_build.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.c_int, C.POINTER(C.c_int)]
    lib.sco_debug_rl_plan.argtypes = [C.POINTER(C.c_int)]
    assert lib.sco_debug_plan_build(len(q["q"]), len(q["l"]), ip(Pp), ip(Pi), ip(Ap), ip(Ai), 1, ip(sizes)) == 0
    assert lib.sco_debug_rl_plan(ip(info)) == 0
    assert bool(info[0]) == fits, (name, info)
    assert info[1] == cw and info[2] == 5 and info[4] + 40 * 1024 <= 160 * 1024


@pytest.mark.parametrize("name,kw,want", [
    ("circles 7x20", {}, (1, 7, 20, 3, 7, 4, 3, 10)), ("circles 3x6", dict(d=3, T=6, K=2, O=2), (1, 3, 6, 8, 8, 1, 1, 4)),
    ("objective terms", dict(ee_cost_weight=1.0), (1, 7, 20, 3, 7, 4, 3, 10)),
    ("point robot", dict(d=2, T=8, O=3, point=True), (1, 2, 8, 8, 8, 1, 1, 4)),
    ("7-DOF x 12: the smaller instantiation", dict(T=12), (1, 7, 12, 5, 8, 2, 2, 8)), ("5-DOF x 16", dict(d=5, T=16, K=4, O=2), (1, 5, 16, 4, 8, 2, 2, 8)),
    ("program rows d=2 T=20", dict(d=2, T=20, K=1, program=True), (1, 2, 20, 3, 8, 2, 2, 10)),
    ("4-DOF x 24: three row slots and 12 steps, no instantiation", dict(d=4, T=24, K=3, O=2), (0,)), ("5-DOF x 30: five row slots", dict(d=5, T=30), (0,)),
    ("velocity limits", dict(vel_limit=0.3), (0,)), ("reach", dict(reach=True), (0,)), ("joint limits", dict(joint_limit=0.2), (0,)),
    ("span-2 program", dict(d=2, T=8, K=1, program=True, variant="sweep"), (0,))])
def test_which_penalty_qps_land_on_the_wavefront_tier(name, kw, want):
    """Host-side plan of the wavefront tier (csrc/sco_admm_wv.hip: wv_plan_build): block-tridiagonal core with diagonal
    couplings, every hinge row inside one timestep block with its own slack, at most two single rows per core variable.
    info = fits, block order, blocks, lanes per block, instantiation <BS, NS, NV, NSTEP>.  Rows on two timesteps (velocity
    limits, span-2 blocks), abs rows with a core slack (reach) and three single rows per variable (joint limits) stay on
    the row-local tier."""
    from oracle import arm_family as af
    from oracle import sco_ref as sr
    out = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(0, **kw)), sr.SolverParams(max_qp_solves=2), record_qps=True)
    q = out.qps[1]
    P = sp.triu(sp.csc_matrix(q["P"] != 0), format="csc"); A = sp.csc_matrix(q["A"] != 0)
    P.sort_indices(); A.sort_indices()
    lib = _lib.load()
    ip = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))
    Pp, Pi, Ap, Ai = (np.ascontiguousarray(a, dtype=np.int32) for a in (P.indptr, P.indices, A.indptr, A.indices))
    sizes = np.zeros(16, dtype=np.int32); info = np.zeros(10, dtype=np.int32)
    lib.sco_debug_plan_build.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.c_int, C.POINTER(C.c_int)]
    lib.sco_debug_wv_plan.argtypes = [C.POINTER(C.c_int)]
    assert lib.sco_debug_plan_build(len(q["q"]), len(q["l"]), ip(Pp), ip(Pi), ip(Ap), ip(Ai), 1, ip(sizes)) == 0
    assert lib.sco_debug_wv_plan(ip(info)) == 0
    assert tuple(info[:len(want)]) == want, (name, info.tolist())
    if want[0]:
        assert info[8] <= 40 * 1024        # LDS: four problems per CU (r04: the Jacobian rows live in registers, the test constants in LDS: 38 KB at 7 x 20)


@pytest.mark.parametrize("name,kw,want", [
    ("12-DOF x 50 x 100 hinge rows (the shape of BASELINE configs[4])", (50, 12, 100), (1, 12, 50, 1, 0)),
    ("12 x 8 x 20", (8, 12, 20), (1, 12, 8, 1, 0)),
    ("12-DOF x 8 arm with the pattern taken from the values: a link's rows stop at its joint", dict(d=12, T=8, K=10, O=10), (1, 12, 8, 0, 4)),
    ("16 x 4, 33 rows", (4, 16, 33), (1, 16, 4, 1, 0)),
    ("14 x 5: blocks of 16 cut the time steps", (5, 14, 9), (1, 16, 5, 0, 4)),
    ("7-DOF x 20: block order below 12", dict(d=7, T=20), (1, 8, 18, 0, 2)),
])
def test_which_structured_plans_form_their_blocks_on_the_matrix_cores(name, kw, want, monkeypatch):
    """r04 (N1): the structured global-memory plan forms the hinge-row part of its diagonal blocks with
    v_mfma_f64_16x16x4 when the block order is 12 .. 16 and every entry of a block sums over the same run of hinge rows
    (csrc/sco_qp_big.hip bt_plan_build); SCO_QP_NO_MFMA=1 switches it off.  info: fits, block order, blocks, MFMA, why not."""
    if isinstance(kw, dict):
        from oracle import arm_family as af
        from oracle import sco_ref as sr
        out = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(0, **kw)), sr.SolverParams(max_qp_solves=2), record_qps=True)
        q = out.qps[1]; Pm, Am, n, m = q["P"], q["A"], len(q["q"]), len(q["l"])
    else:
        from test_qp_gpu import penalty_qp
        Pm, qv, Am, lv, uv = penalty_qp(np.random.default_rng(0), *kw); n, m = len(qv), len(lv)
    P = sp.triu(sp.csc_matrix(Pm != 0), format="csc"); A = sp.csc_matrix(Am != 0)
    P.sort_indices(); A.sort_indices()
    lib = _lib.load()
    ip = lambda a: np.ascontiguousarray(a, dtype=np.int32).ctypes.data_as(C.POINTER(C.c_int))
    Pp, Pi, Ap, Ai = (np.ascontiguousarray(a, dtype=np.int32) for a in (P.indptr, P.indices, A.indptr, A.indices))
    sizes = np.zeros(16, dtype=np.int32); info = np.zeros(8, dtype=np.int32)
    lib.sco_debug_plan_build.argtypes = [C.c_int, C.c_int] + [C.POINTER(C.c_int)] * 4 + [C.c_int, C.POINTER(C.c_int)]
    lib.sco_debug_bt_plan.argtypes = [C.POINTER(C.c_int)]
    assert lib.sco_debug_plan_build(n, m, ip(Pp), ip(Pi), ip(Ap), ip(Ai), 1, ip(sizes)) == 0
    assert lib.sco_debug_bt_plan(ip(info)) == 0
    assert tuple(info[:5]) == want, (name, info.tolist())
    if want[3]:
        monkeypatch.setenv("SCO_QP_NO_MFMA", "1")
        assert lib.sco_debug_bt_plan(ip(info)) == 0 and tuple(info[:5]) == want[:3] + (0, 1)
