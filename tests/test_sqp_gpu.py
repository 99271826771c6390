"""Device-resident penalty-SQP loop (sco_sqp_* through the C ABI) against the flat
oracle and against the golden vectors recorded from the reference's own code."""
import os

import numpy as np
import pytest

import conftest as ct
from oracle import arm_family as af
from oracle import sco_ref as sr
from sco_py_amd import _lib, batch as sb

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-6          # abs, BASELINE.json north_star
SMALL = dict(d=3, T=6, K=2, O=2)


def _compare(res, probs, which, oracle_params=None, analytic=False, memo=True, qp_settings=None):
    for b in which:
        ref = sr.penalty_sqp(sr.trajopt_flat(probs[b], analytic_jac=analytic), oracle_params, emulate_memo=memo,
                             qp_settings=qp_settings)
        tr = res.trace[b]
        rt = ref.trace[:64]                       # the device keeps the first 64 decisions of a problem
        assert tr.shape == rt.shape, (b, tr.shape, ref.trace.shape)
        assert np.array_equal(tr[:, 0], rt[:, 0]), (b, tr[:, 0], rt[:, 0])                     # same decisions
        assert np.array_equal(tr[:, 6:8], rt[:, 6:8]), b                                       # QP status + iterations
        assert np.abs(tr[:, 1:4] - rt[:, 1:4]).max() < 1e-7 * (1 + np.abs(rt[:, 1:4]).max()), b
        assert np.array_equal(tr[:, 4:6], rt[:, 4:6]), b                                       # trust, penalty
        assert np.abs(res.x[b] - ref.x).max() < TOL, (b, np.abs(res.x[b] - ref.x).max())
        assert bool(res.success[b]) == ref.success
        assert (res.sqp_iters[b], res.qp_solves[b], res.admm_iters[b]) == (ref.sqp_iters, ref.qp_solves, ref.admm_iters)
        assert abs(res.max_violation[b] - ref.max_violation) < 1e-7


def test_small_batch_parity_mode(gpu):
    arrays, probs = af.make_batch(8, **SMALL)
    _compare(sb.solve_batch(arrays), probs, range(8))


def test_small_batch_intended_mode(gpu):
    arrays, probs = af.make_batch(6, **SMALL)
    p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0)
    _compare(sb.solve_batch(arrays, params=p), probs, range(6),
             sr.SolverParams(compound_penalty=False, duplicate_rows=False))


def test_penalty_escalation_and_custom_knobs(gpu):
    arrays, probs = af.make_batch(6, first=8, **SMALL)
    kw = dict(initial_penalty_coeff=10.0, max_merit_coeff_increases=3, initial_trust_region_size=0.5,
              min_trust_region_size=1e-3, improve_ratio_threshold=0.2)
    p = _lib.default_sqp_params(**kw)
    _compare(sb.solve_batch(arrays, params=p), probs, range(6), sr.SolverParams(**kw))


def test_memoisation_on_rounded_points_is_reproduced_and_can_be_switched_off(gpu):
    """Q3 (expr.py:13, 31-41, 323-332): problems 15 and 16 of the 7x20 workload take a different
    decision (group-converged instead of y-converged) because a block of the point being
    convexified rounds onto an already convexified point and reuses its old affine model."""
    arrays, probs = af.make_batch(2, first=15)
    on = sb.solve_batch(arrays)
    _compare(on, probs, range(2), memo=True)
    off = sb.solve_batch(arrays, params=_lib.default_sqp_params(memoize_rounded=0))
    _compare(off, probs, range(2), memo=False)
    assert [int(t[-1, 0]) for t in on.trace] == [6, 6] and [int(t[-1, 0]) for t in off.trace] == [3, 3]


def test_analytic_jacobian_path(gpu):
    arrays, probs = af.make_batch(4, **SMALL)
    _compare(sb.solve_batch(arrays, analytic_jac=True), probs, range(4), analytic=True)


def test_matches_reference_golden_run_small(gpu):
    """x, success and the per-QP status/iterations recorded from the REFERENCE's own
    modules (tests/golden/make_golden.py)."""
    g = np.load(os.path.join(GOLD, "trajopt_small.npz"))
    arrays, _ = af.make_batch(4, **SMALL)
    res = sb.solve_batch(arrays)
    for i in range(4):
        gq = ct.load_golden_qps(g, "p%d_" % i)
        assert np.abs(res.x[i] - g["p%d_x" % i]).max() < TOL
        assert bool(res.success[i]) == bool(g["p%d_success" % i])
        assert [int(v) for v in res.trace[i][:, 6]] == [q["status"] for q in gq]
        assert [int(v) for v in res.trace[i][:, 7]] == [q["iters"] for q in gq]


def test_matches_reference_golden_run_7x20(gpu):
    g = np.load(os.path.join(GOLD, "trajopt_7x20.npz"))
    arrays, _ = af.make_batch(2)
    res = sb.solve_batch(arrays)
    gq = ct.load_golden_qps(g, "p0_", sparse=True)
    assert np.abs(res.x[0] - g["p0_x"]).max() < TOL
    assert bool(res.success[0]) == bool(g["p0_success"])
    assert [int(v) for v in res.trace[0][:, 7]] == [q["iters"] for q in gq]
    assert abs(res.max_violation[0] - float(g["p0_max_violation"])) < 1e-7


def test_7x20_batch_against_oracle(gpu):
    arrays, probs = af.make_batch(16)
    _compare(sb.solve_batch(arrays), probs, range(3))


def test_repeated_solves_restart_from_the_loaded_state_and_are_deterministic(gpu):
    arrays, _ = af.make_batch(8, **SMALL)
    with sb.TrajOptBatch(8, 3, 6, 2, 2) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        tb.solve(); a = tb.fetch()
        tb.solve(); b = tb.fetch()
    assert np.array_equal(a.x, b.x) and np.array_equal(a.admm_iters, b.admm_iters)


def test_full_size_batch_properties(gpu):
    """BASELINE configs[2] size (1024 problems): size-independent properties."""
    B = 1024
    arrays, _ = af.make_batch(B)
    res = sb.solve_batch(arrays)
    d, T = 7, 20
    x = res.x.reshape(B, T, d)
    # linear constraints hold to QP accuracy at every returned point
    assert np.abs(x[:, 0, :] - arrays["start"]).max() < 1e-4 and np.abs(x[:, -1, :] - arrays["goal"]).max() < 1e-4
    # success implies the non-linear constraints are within tolerance (solver.py:94-101)
    assert np.all(res.max_violation[res.success] <= 1e-4)
    # the reported violation is the violation of the reported trajectory
    for b in range(0, B, 97):
        v = max(np.max(af.arm_dist(x[b, t], arrays["link_len"][b], arrays["point_link"], arrays["point_frac"],
                                   arrays["obstacles"][b])) for t in range(T))
        assert abs(max(v, 0.0) - res.max_violation[b]) < 1e-9
    # every problem did at least the projection and one SQP iteration; counters are consistent
    assert np.all(res.qp_solves >= 2) and np.all(res.sqp_iters >= 1) and np.all(res.qp_solves > res.sqp_iters - 1)
    assert np.all(np.isfinite(res.x))


def test_full_size_batch_sampled_against_the_oracle(gpu):
    """BASELINE configs[2] at full size: every 16th problem of the 1024-problem bench batch against the oracle
    (same decisions, QP statuses, ADMM iteration counts per QP, merits, trajectory to 1e-6) -- the driver-run
    slice of the whole-batch sweeps in scripts/gpu_parity_sweep.py (64 oracle solves, about a minute of CPU)."""
    B = 1024
    arrays, probs = af.make_batch(B)
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(0, B, 16))


def test_descriptor_validation(gpu):
    with pytest.raises(_lib.ScoHipError) as e:
        sb.TrajOptBatch(0, 3, 6, 2, 2)
    assert e.value.code == -1
    with sb.TrajOptBatch(2, 3, 6, 2, 2) as tb:
        with pytest.raises(_lib.ScoHipError) as e:
            tb.solve()                     # solve before load
        assert e.value.code == -4


def test_sqp_through_the_global_memory_tier(gpu, monkeypatch):
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    arrays, probs = af.make_batch(3, **SMALL)
    _compare(sb.solve_batch(arrays), probs, range(3))


def test_12dof_50step_problem_against_stored_oracle_run(gpu):
    """BASELINE configs[4]: 12-DOF x 50 timesteps, 5000 non-linear rows (n = 5600,
    m = 10 624).  The oracle needs minutes on a CPU for this size, so its answer is a
    committed fixture (tests/golden/trajopt_12x50_oracle.npz, made by
    tests/golden/make_big_oracle.py)."""
    path = os.path.join(GOLD, "trajopt_12x50_oracle.npz")
    g = np.load(path)
    arrays, _ = af.make_batch(1, d=12, T=50, K=10, O=10)
    res = sb.solve_batch(arrays)
    tr = res.trace[0]
    assert np.array_equal(tr[:, 0], g["trace"][:, 0])                 # same decisions
    assert np.array_equal(tr[:, 6:8], g["trace"][:, 6:8])             # same QP status / iterations
    assert np.abs(res.x[0] - g["x"]).max() < TOL
    assert bool(res.success[0]) == bool(g["success"])
    # ... and the same problem run by the REFERENCE's own modules (tests/golden/make_golden_12x50.py)
    r = np.load(os.path.join(GOLD, "trajopt_12x50.npz"))
    assert bool(res.success[0]) == bool(r["p0_success"]) and np.abs(res.x[0] - r["p0_x"]).max() < TOL
    assert tr.shape[0] == int(r["p0_n_qp_total"])
    assert np.array_equal(tr[:, 6], r["p0_qp_status"]) and np.array_equal(tr[:, 7], r["p0_qp_iters"])
    assert abs(res.max_violation[0] - float(r["p0_max_violation"])) < 1e-7


def test_three_chunks_in_flight_on_the_structured_tier_change_no_bit(gpu, monkeypatch):
    """r03: where three dense chunks of full width qualify (known per-row constants), the structured kernel issues the loads
    of all three before the arithmetic of the first (csrc/sco_qp_big.hip, BtLean).  The arithmetic is the pairwise loop's:
    SCO_QP_BT_TRIPLE=0 gives the same bits -- 12-DOF x 50 problems, solved in slices (park / resume) and in one piece."""
    arrays, _ = af.make_batch(3, d=12, T=50, K=10, O=10)
    out = {}
    for triple in ("1", "0"):
        monkeypatch.setenv("SCO_QP_BT_TRIPLE", triple)
        for slice_ in (-1, 1500):
            p = _lib.default_sqp_params(max_sqp_iters=2, admm_slice=slice_)
            res = sb.solve_batch(arrays, params=p, qp_settings=_lib.default_qp_settings(max_iter=6000))
            out[triple, slice_] = (res.x.copy(), [t.copy() for t in res.trace], res.admm_iters.copy())
    ref = out["0", -1]
    for key, (x, tr, it) in out.items():
        assert np.array_equal(it, ref[2]), key
        assert all(np.array_equal(a, b) for a, b in zip(tr, ref[1])), key
        assert np.array_equal(x, ref[0]), key


@pytest.mark.parametrize("shape,B", [((1, 2, 1, 1), 1), ((2, 3, 1, 1), 3), ((1, 40, 1, 2), 2), ((5, 2, 3, 1), 5)])
def test_minimal_and_odd_shapes(gpu, shape, B):
    """Smallest legal descriptor (1 joint, 2 steps, 1 point, 1 obstacle, batch 1), odd batch sizes, a
    long thin problem and a two-step one: same decisions and answers as the oracle."""
    d, T, K, O = shape
    arrays, probs = af.make_batch(B, d=d, T=T, K=K, O=O)
    _compare(sb.solve_batch(arrays), probs, range(B))


def test_longest_horizon_is_accepted_and_one_more_is_refused(gpu):
    arrays, probs = af.make_batch(1, d=1, T=256, K=1, O=1)
    res = sb.solve_batch(arrays)
    assert np.all(np.isfinite(res.x)) and res.qp_solves[0] >= 2
    x = res.x.reshape(1, 256, 1)
    assert np.abs(x[:, 0, :] - arrays["start"]).max() < 1e-4 and np.abs(x[:, -1, :] - arrays["goal"]).max() < 1e-4
    with pytest.raises(_lib.ScoHipError) as e:
        sb.TrajOptBatch(1, 1, 257, 1, 1)
    assert e.value.code == -1


REACH = dict(d=3, T=6, K=2, O=2, reach=True)


def test_reach_family_matches_oracle(gpu):
    """SCO_FAM_ARM_REACH: non-linear equality rows (end-effector target) lowered to the abs penalty
    with two slack variables per row (prob.py:280-315), on the device."""
    arrays, probs = af.make_batch(6, **REACH)
    _compare(sb.solve_batch(arrays), probs, range(6))
    p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0)
    _compare(sb.solve_batch(arrays, params=p), probs, range(6),
             sr.SolverParams(compound_penalty=False, duplicate_rows=False))


def test_reach_family_analytic_jacobian_and_no_memo(gpu):
    arrays, probs = af.make_batch(3, first=6, **REACH)
    _compare(sb.solve_batch(arrays, analytic_jac=True), probs, range(3), analytic=True)
    p = _lib.default_sqp_params(memoize_rounded=0)
    _compare(sb.solve_batch(arrays, params=p), probs, range(3), memo=False)


def test_reach_family_matches_reference_golden_run(gpu):
    g = np.load(os.path.join(GOLD, "trajopt_reach.npz"))
    arrays, _ = af.make_batch(3, **REACH)
    res = sb.solve_batch(arrays)
    for i in range(3):
        assert np.abs(res.x[i] - g["p%d_x" % i]).max() < TOL
        assert bool(res.success[i]) == bool(g["p%d_success" % i])
        assert abs(res.max_violation[i] - float(g["p%d_max_violation" % i])) < 1e-7
        assert [int(v) for v in res.trace[i][:, 7]] == [int(g["p%d_qp%d_iters" % (i, k)]) for k in range(int(g["p%d_n_qp" % i]))]


def test_reach_family_7x20_batch(gpu):
    arrays, probs = af.make_batch(4, reach=True)
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(2))
    x = res.x.reshape(4, 20, 7)
    assert np.abs(x[:, 0, :] - arrays["start"]).max() < 1e-4
    for b in range(4):
        assert abs(max(np.abs(af.ee_pos(x[b, -1], arrays["link_len"][b]) - arrays["target"][b]).max(),
                       max(np.max(af.arm_dist(x[b, t], arrays["link_len"][b], arrays["point_link"], arrays["point_frac"],
                                              arrays["obstacles"][b])) for t in range(20)), 0.0)
                   - res.max_violation[b]) < 1e-9


def test_reach_family_needs_its_target(gpu):
    arrays, _ = af.make_batch(1, **REACH)
    with sb.TrajOptBatch(1, 3, 6, 2, 2, reach=True) as tb:
        with pytest.raises(ValueError):
            tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                    arrays["point_frac"], arrays["obstacles"])


def _compare_groups(res, probs, which, oracle_params=None):
    _compare(res, probs, which, oracle_params)
    for b in which:
        ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), oracle_params, emulate_memo=True)
        assert sorted(res.nonconverged_groups[b]) == sorted(ref.nonconverged_groups), (b, res.nonconverged_groups[b])


@pytest.mark.parametrize("scheme,reach", [("halves", False), ("split", True), ("split", False)])
def test_constraint_groups_on_device(gpu, scheme, reach):
    """prob.add_cnt_expr(..., group_ids): per-group merit vectors, overlap graph, nonconverged groups
    (solver.py:155-161, 209-235) in the device loop, against the oracle (itself pinned to reference
    runs with groups, tests/test_golden.py)."""
    kw = dict(d=3, T=6, K=2, O=2, groups=scheme, reach=reach)
    arrays, probs = af.make_batch(12, first=8, **kw)      # problems 10, 13 stall a group
    res = sb.solve_batch(arrays)
    _compare_groups(res, probs, range(12))
    knobs = dict(initial_penalty_coeff=10.0, max_merit_coeff_increases=3)
    _compare_groups(sb.solve_batch(arrays, params=_lib.default_sqp_params(**knobs)), probs, range(12), sr.SolverParams(**knobs))


def test_constraint_groups_match_reference_golden_runs(gpu):
    g = np.load(os.path.join(GOLD, "trajopt_groups.npz"))
    for prefix, i in (("s22_", 22), ("s35_", 35), ("s38_", 38), ("s58_", 58)):
        arrays, _ = af.make_batch(1, first=i, d=3, T=6, K=2, O=2, groups="split", reach=True)
        res = sb.solve_batch(arrays)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL
        assert bool(res.success[0]) == bool(g[prefix + "success"])
        assert sorted(res.nonconverged_groups[0]) == sorted(str(s) for s in g[prefix + "nonconverged"])
        assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(int(g[prefix + "n_qp"]))]


def test_default_group_is_all(gpu):
    arrays, probs = af.make_batch(3, first=10, d=3, T=6, K=2, O=2)
    res = sb.solve_batch(arrays)
    for b in range(3):
        ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), emulate_memo=True)
        assert res.nonconverged_groups[b] == ref.nonconverged_groups


@pytest.mark.parametrize("structured", ["0", "1"])
def test_reach_family_and_groups_through_the_global_memory_tier(gpu, monkeypatch, structured):
    """The equality rows put one abs slack of every pair into the dense core and add rows that are neither
    dense chunks nor single-entry rows: both forms of the global-memory tier must handle them."""
    monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    monkeypatch.setenv("SCO_QP_NO_BT", "0" if structured == "1" else "1")
    arrays, probs = af.make_batch(3, d=3, T=6, K=2, O=2, reach=True, groups="split")
    _compare_groups(sb.solve_batch(arrays), probs, range(3))
    arrays, probs = af.make_batch(2, d=4, T=5, K=5, O=4, reach=True)       # 20 rows per block: dense chunks
    _compare(sb.solve_batch(arrays), probs, range(2))


def test_family_specific_calls_are_validated(gpu):
    arrays, _ = af.make_batch(1, d=3, T=6, K=2, O=2)
    with sb.TrajOptBatch(1, 3, 6, 2, 2) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        with pytest.raises(_lib.ScoHipError) as e:       # a target only exists in the reach family
            _lib.check(_lib.load().sco_sqp_load_target(tb._h, _lib.dptr(np.zeros((1, 2)))))
        assert e.value.code == -1
        with pytest.raises(ValueError):
            tb.set_groups([["all"]] * 5)                  # one entry per constraint block (6 here)
        with pytest.raises(ValueError):
            tb.set_groups([["g%d" % k for k in range(40)]] * 6)     # at most 32 groups
        tb.set_groups([["a"], ["a"], ["a", "b"], ["b"], ["b"], ["b"]])
        tb.solve()
        assert tb.fetch().nonconverged_groups[0] in ([], ["a"], ["b"], ["a", "b"])
    with sb.TrajOptBatch(1, 3, 6, 2, 2, reach=True) as tb:
        r = af.make_batch(1, d=3, T=6, K=2, O=2, reach=True)[0]
        tb.load(r["x0"], r["start"], r["goal"], r["link_len"], r["point_link"], r["point_frac"], r["obstacles"],
                target=r["target"])
        with pytest.raises(ValueError):
            tb.set_groups([["all"]] * 6)                  # the reach block is a seventh block


def test_velocity_limits_match_oracle(gpu):
    """SCO_FAM_FLAG_VEL_LIMITS: linear inequality rows |theta[t+1] - theta[t]| <= vmax in every QP,
    with and without the reach equality."""
    for kw in (dict(vel_limit=0.6), dict(vel_limit=0.6, reach=True), dict(vel_limit=0.3, groups="split", reach=True)):
        arrays, probs = af.make_batch(6, d=3, T=6, K=2, O=2, **kw)
        res = sb.solve_batch(arrays)
        _compare(res, probs, range(6))
        x = res.x.reshape(6, 6, 3)
        ok = res.qp_solves > 1                      # problems whose projection QP was feasible
        assert np.all(np.abs(np.diff(x[ok], axis=1)) <= kw["vel_limit"] + 1e-4)


def test_velocity_limits_match_reference_golden_runs_incl_infeasible_projection(gpu):
    import sys
    sys.path.insert(0, GOLD)
    from vel_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_vel.npz"))
    for prefix, kw, i in CASES:
        arrays, _ = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, prefix
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix
        assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(nq)], prefix
    # x0_: pins and limits contradict each other -> the projection QP is primal infeasible (-3),
    # Solver.solve gives up with the variables untouched (solver.py:81-82)
    arrays, _ = af.make_batch(1, d=3, T=6, K=2, O=2, vel_limit=0.05)
    res = sb.solve_batch(arrays)
    assert not res.success[0] and res.qp_solves[0] == 1 and res.sqp_iters[0] == 0
    assert np.array_equal(res.x[0], arrays["x0"][0])


def test_velocity_limits_7x20_batch(gpu):
    arrays, probs = af.make_batch(4, vel_limit=0.3)
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(2))
    with sb.TrajOptBatch(1, 3, 6, 2, 2, vel_limits=True) as tb:
        a = af.make_batch(1, d=3, T=6, K=2, O=2, vel_limit=0.5)[0]
        with pytest.raises(ValueError):
            tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"])


def test_warm_started_qps_in_the_device_loop(gpu):
    """sco_sqp_params.warm_start_qps (beyond parity): same kind of answer with fewer ADMM iterations;
    repeated solves stay deterministic (the first penalty QP always starts from zero)."""
    arrays, probs = af.make_batch(64)
    cold_p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0, max_sqp_iters=12)
    warm_p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0, max_sqp_iters=12, warm_start_qps=1)
    with sb.TrajOptBatch(64, 7, 20, 5, 2) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        tb.solve(cold_p); cold = tb.fetch()
        tb.solve(warm_p); w1 = tb.fetch()
        tb.solve(warm_p); w2 = tb.fetch()
    assert np.array_equal(w1.x, w2.x) and np.array_equal(w1.admm_iters, w2.admm_iters)
    assert w1.admm_iters.sum() < 0.8 * cold.admm_iters.sum()
    both = cold.success & w1.success
    assert both.sum() >= 0.8 * cold.success.sum()
    assert np.all(w1.max_violation[w1.success] <= 1e-4)
    x = w1.x.reshape(64, 20, 7)
    assert np.abs(x[:, 0, :] - arrays["start"]).max() < 1e-4 and np.abs(x[:, -1, :] - arrays["goal"]).max() < 1e-4


@pytest.mark.parametrize("seed", range(12))
def test_random_shapes_and_feature_mixes_match_oracle(gpu, seed):
    """Random small descriptors with random combinations of reach / velocity limits / groups / analytic
    Jacobians / solver knobs: same decisions, iteration counts and answers as the oracle."""
    rng = np.random.default_rng(4000 + seed)
    d, T, K, O = int(rng.integers(1, 6)), int(rng.integers(2, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    kw = dict(d=d, T=T, K=K, O=O, reach=bool(rng.integers(2)))
    if rng.integers(2):
        kw["vel_limit"] = float(rng.uniform(0.4, 1.5))
    if rng.integers(2):
        kw["groups"] = ["halves", "split"][int(rng.integers(2))]
    analytic = bool(rng.integers(2))
    knobs = dict(initial_penalty_coeff=float(10 ** rng.uniform(0.5, 3)), max_merit_coeff_increases=int(rng.integers(1, 4)),
                 compound_penalty=int(rng.integers(2)), duplicate_rows=int(rng.integers(2)))
    arrays, probs = af.make_batch(3, first=int(rng.integers(0, 50)), **kw)
    res = sb.solve_batch(arrays, params=_lib.default_sqp_params(max_sqp_iters=40, **knobs), analytic_jac=analytic)
    op = sr.SolverParams(initial_penalty_coeff=knobs["initial_penalty_coeff"],
                         max_merit_coeff_increases=knobs["max_merit_coeff_increases"],
                         compound_penalty=bool(knobs["compound_penalty"]), duplicate_rows=bool(knobs["duplicate_rows"]),
                         max_qp_solves=40)
    _compare(res, probs, range(3), op, analytic=analytic)


def test_unknown_initial_values_are_left_out_of_the_projection(gpu):
    """NaN entries of the initial trajectory (Variable values that are not known yet) do not enter
    find_closest_feasible_point's distance (prob.py:394-404): same answer as the oracle."""
    arrays, probs = af.make_batch(4, d=3, T=6, K=2, O=2)
    for b in range(4):
        x0 = arrays["x0"][b].reshape(6, 3)
        x0[2 + (b % 2), :] = np.nan; x0[4, b % 3] = np.nan
        probs[b]["x0"] = arrays["x0"][b].copy()
    res = sb.solve_batch(arrays)
    assert np.all(np.isfinite(res.x))
    _compare(res, probs, range(4))


def test_diagnostic_flags(gpu):
    """fetch().flags: bit 2 = stopped by max_sqp_iters (the reference's loops are unbounded); parity-mode
    runs of the small workload set no flag."""
    arrays, _ = af.make_batch(8, d=3, T=6, K=2, O=2)
    res = sb.solve_batch(arrays)
    assert np.all(res.flags == 0)
    p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0, max_sqp_iters=3)
    res = sb.solve_batch(arrays, params=p)
    capped = (res.flags & 2) != 0
    assert capped.any() and np.all(res.qp_solves[capped] == 3) and not np.any(res.success[capped])
    assert np.all(res.qp_solves[~capped] <= 3)


@pytest.mark.parametrize("tier", ["row-local", "register", "sliced-ELL", "generic", "structured"])
def test_time_slicing_changes_the_schedule_not_the_results(gpu, monkeypatch, tier):
    """sco_sqp_params.admm_slice: parked and resumed ADMM solves continue bit-exactly, so any slice length
    gives the same trajectories, decisions and iteration counts as one launch per QP (row-local, register-offset --
    since r03 --, sliced-ELL, generic and structured global-memory kernels)."""
    if tier == "register":
        monkeypatch.setenv("SCO_QP_NO_RL", "1")
    if tier == "generic":
        for k in ("SCO_QP_NO_RL", "SCO_QP_NO_REG", "SCO_QP_NO_FAST"):
            monkeypatch.setenv(k, "1")
    if tier == "sliced-ELL":
        for k in ("SCO_QP_NO_RL", "SCO_QP_NO_REG"):
            monkeypatch.setenv(k, "1")
    if tier == "structured":
        monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    nb, dims = (48, (7, 20, 5, 2)) if tier == "row-local" else (6, (4, 8, 8, 3))      # 24 rows per block: dense chunks
    arrays, _ = af.make_batch(nb, d=dims[0], T=dims[1], K=dims[2], O=dims[3])
    outs = []
    with sb.TrajOptBatch(nb, *dims) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        for sl in (-1, 0, 1000, 30, 77777):
            tb.solve(_lib.default_sqp_params(admm_slice=sl))
            r = tb.fetch(); r.trace = tb.trace()
            outs.append(r)
    for r in outs[1:]:
        assert np.array_equal(r.x, outs[0].x) and np.array_equal(r.admm_iters, outs[0].admm_iters)
        assert np.array_equal(r.success, outs[0].success) and np.array_equal(r.qp_solves, outs[0].qp_solves)
        assert all(np.array_equal(a, b) for a, b in zip(r.trace, outs[0].trace))


@pytest.mark.parametrize("tier,quirks", [("row-local", True), ("row-local", False), ("structured", True)])
def test_round_selection_and_stream_groups_change_the_schedule_not_the_results(gpu, monkeypatch, tier, quirks):
    """With more live problems than CUs a round runs whole passes only, as a compact launch over the problems with most
    in front of them (sqp_select_kernel; SCO_SQP_SELECT=0: plain lock-step rounds), or -- opt-in, SCO_SQP_GROUPS -- the
    batch is cut into stream groups whose rounds run side by side.  Either way every problem sees the same kernels in the
    same order: trajectories, decisions and iteration counts are bit-identical for every schedule and slice length."""
    if tier == "structured":
        monkeypatch.setenv("SCO_QP_FORCE_BIG", "1")
    nb, dims = (700, (3, 6, 2, 2)) if tier == "row-local" else (300, (4, 8, 8, 3))
    arrays, probs = af.make_batch(nb, d=dims[0], T=dims[1], K=dims[2], O=dims[3])
    # (selection, stream groups, slice); since r03 selection also runs inside every stream group
    schedules = [("0", "1", 400), ("1", "1", 400), ("1", "1", 175), ("0", "2", 400), ("0", "3", 175), ("1", "2", 400), ("1", "2", 175)]
    outs = []
    with sb.TrajOptBatch(nb, *dims) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        # quirks off: up to 20 penalty QPs per problem, accepted steps, shrinks and escalations (many transitions)
        kw = {} if quirks else dict(compound_penalty=0, duplicate_rows=0, max_sqp_iters=20)
        for select, groups, sl in schedules:
            monkeypatch.setenv("SCO_SQP_SELECT", select); monkeypatch.setenv("SCO_SQP_GROUPS", groups)
            tb.solve(_lib.default_sqp_params(admm_slice=sl, **kw))
            r = tb.fetch(); r.trace = tb.trace(); r.timing = tb.last_timing()
            outs.append(r)
    # 700 problems on 256 CUs: at most two groups; 300: one
    assert [r.timing["groups"] for r in outs] == ([1, 1, 1, 2, 2, 2, 2] if nb >= 512 else [1] * 7)
    if nb >= 512:
        assert outs[3].timing["launches"] > outs[0].timing["launches"] >= outs[0].timing["rounds"] - 1
    assert outs[1].timing["rounds"] >= outs[0].timing["rounds"]        # problems that sit rounds out need more of them
    for r in outs[1:]:
        assert np.array_equal(r.x, outs[0].x) and np.array_equal(r.admm_iters, outs[0].admm_iters)
        assert np.array_equal(r.success, outs[0].success) and np.array_equal(r.qp_solves, outs[0].qp_solves)
        assert np.array_equal(r.merit, outs[0].merit) and np.array_equal(r.max_violation, outs[0].max_violation)
        assert all(np.array_equal(a, b) for a, b in zip(r.trace, outs[0].trace))
    _compare(outs[1], probs, range(0, nb, 97),
             None if quirks else sr.SolverParams(compound_penalty=False, duplicate_rows=False, max_qp_solves=20))


@pytest.mark.parametrize("kw", [dict(reach=True, vel_limit=0.6, groups="halves"), dict(joint_limit=0.3, vel_limit=0.6),
                                dict(ee_cost_weight=0.5)], ids=["reach+vel+groups", "jl+vel", "objective terms"])
def test_round_selection_with_every_family_feature(gpu, kw):
    """A batch just above the CU count (one problem sits out every round until the first finishes) and very short
    slices (many rounds, many park / resume cycles): reach equality, velocity and joint limits, constraint groups and
    non-quadratic objective terms under round selection against the oracle."""
    nb = 260
    arrays, probs = af.make_batch(nb, d=3, T=6, K=2, O=2, **kw)
    with sb.TrajOptBatch(nb, 3, 6, 2, 2, reach=bool(kw.get("reach")), vel_limits="vel_limit" in kw,
                         joint_limits="joint_limit" in kw, ee_cost="ee_cost_weight" in kw) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"], target=arrays.get("target"), vmax=arrays.get("vmax"),
                jlo=arrays.get("jlo"), jhi=arrays.get("jhi"), cost_weight=arrays.get("cost_weight"),
                cost_target=arrays.get("cost_target"))
        if arrays.get("groups") is not None:
            tb.set_groups(arrays["groups"])
        tb.solve(_lib.default_sqp_params(admm_slice=75))
        res = tb.fetch(); res.trace = tb.trace(); tm = tb.last_timing()
    assert tm["groups"] == 1 and tm["rounds"] > 20
    if "ee_cost_weight" in kw:
        # numeric Hessians amplify last-bit differences: decisions, statuses and trajectories only
        for b in range(0, nb, 37):
            ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), emulate_memo=True)
            assert np.array_equal(res.trace[b][:, 0], ref.trace[:64, 0]) and np.array_equal(res.trace[b][:, 6], ref.trace[:64, 6])
            assert np.abs(res.x[b] - ref.x).max() < TOL
    else:
        _compare(res, probs, range(0, nb, 37))


def test_adaptive_rho_in_the_device_loop(gpu):
    """sco_qp_settings.adaptive_rho (solver.py:39; reference default off): the device loop parks every QP at each
    rho-update point, re-estimates rho, refactors and resumes.  Not parity mode.  Small problems follow the oracle
    (same rule) decision for decision; on 7-DOF x 20 the ill-conditioned QPs that run to max_iter sit so close to
    the rho thresholds that the oracle's KKT route (sigma = 5e-10 leaves 1e-7 of noise in its residuals) and the
    device's reduced system part ways there, so only the well-conditioned part is compared: the projection and
    the first penalty QP, iteration for iteration."""
    st = _lib.default_qp_settings(adaptive_rho=1)
    arrays, probs = af.make_batch(6, **SMALL)
    res = sb.solve_batch(arrays, qp_settings=st)
    for b in range(6):
        ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), qp_settings=dict(adaptive_rho=1))
        tr, rt = res.trace[b], ref.trace[:64]
        assert tr.shape == rt.shape and np.array_equal(tr[:, 0], rt[:, 0]), b          # same decisions
        assert np.array_equal(tr[:, 6], rt[:, 6]) and np.array_equal(tr[:2, 7], rt[:2, 7]), b
        assert bool(res.success[b]) == ref.success
        assert np.abs(res.x[b] - ref.x).max() < 1e-4, (b, np.abs(res.x[b] - ref.x).max())
        assert abs(res.max_violation[b] - ref.max_violation) < 1e-5
    assert res.admm_iters.sum() < 0.75 * sb.solve_batch(arrays).admm_iters.sum()
    arrays, probs = af.make_batch(3)
    res = sb.solve_batch(arrays, qp_settings=st)
    for b in range(3):
        ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), qp_settings=dict(adaptive_rho=1))
        assert np.array_equal(res.trace[b][:2, [0, 6, 7]], ref.trace[:2, [0, 6, 7]]), b
        assert np.abs(res.trace[b][:2, 1:4] - ref.trace[:2, 1:4]).max() < 1e-6 * (1 + np.abs(ref.trace[:2, 1:4]).max())
    fixed = sb.solve_batch(arrays)
    assert np.all(res.trace[b][1, 7] < 0.2 * fixed.trace[b][1, 7] for b in range(3))   # first penalty QP: 2275 vs 21625 ...
    again = sb.solve_batch(arrays, qp_settings=st)
    assert np.array_equal(again.x, res.x) and np.array_equal(again.admm_iters, res.admm_iters)    # deterministic


def test_12dof_50step_with_adaptive_rho_follows_the_oracle(gpu):
    """BASELINE configs[4] shape with the quirks off and adaptive rho: the structured global-memory kernel parks at
    every rho change, setup + block factorisation run again, the solve resumes.  Same decisions, statuses and ADMM
    iteration counts as the oracle with the same rule (10 650 iterations in all, against > 200 000 with the fixed rho)."""
    arrays, probs = af.make_batch(1, d=12, T=50, K=10, O=10)
    p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0)
    res = sb.solve_batch(arrays, params=p, qp_settings=_lib.default_qp_settings(adaptive_rho=1), analytic_jac=True)
    ref = sr.penalty_sqp(sr.trajopt_flat(probs[0], analytic_jac=True),
                         sr.SolverParams(compound_penalty=False, duplicate_rows=False), qp_settings=dict(adaptive_rho=1))
    tr, rt = res.trace[0], ref.trace[:64]
    assert tr.shape == rt.shape and np.array_equal(tr[:, 0], rt[:, 0])
    assert np.array_equal(tr[:, 6:8], rt[:, 6:8]), (tr[:, 6:8], rt[:, 6:8])
    assert bool(res.success[0]) == ref.success and np.abs(res.x[0] - ref.x).max() < 1e-5
    assert res.admm_iters[0] < 20000


def test_joint_limits_match_oracle(gpu):
    """SCO_FAM_FLAG_JOINT_LIMITS: lo <= theta[t] <= hi as linear inequality rows in every QP (the projection QP
    included), alone and together with velocity limits, the reach equality and constraint groups."""
    for kw in (dict(joint_limit=0.3), dict(joint_limit=0.05), dict(joint_limit=0.3, vel_limit=0.6),
               dict(joint_limit=0.3, reach=True, groups="split")):
        arrays, probs = af.make_batch(6, d=3, T=6, K=2, O=2, **kw)
        res = sb.solve_batch(arrays)
        _compare(res, probs, range(6))
        x = res.x.reshape(6, 6, 3)
        ok = res.qp_solves > 1                      # problems whose projection QP was feasible
        assert np.all(x[ok] <= arrays["jhi"][ok][:, None, :] + 1e-4) and np.all(x[ok] >= arrays["jlo"][ok][:, None, :] - 1e-4)


def test_joint_limits_match_reference_golden_runs(gpu):
    import sys
    sys.path.insert(0, GOLD)
    from jl_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_jl.npz"))
    for prefix, kw, i in CASES:
        arrays, _ = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, prefix
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix
        assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(nq)], prefix


def test_point_robot_family_matches_oracle(gpu):
    """SCO_FAM_POINT_CIRCLES: a point robot in the plane instead of the arm (rows r_o - ||x[0:2] - c_o||; a third state
    coordinate stays unconstrained), alone, with velocity limits, a workspace box and constraint groups; numeric and
    analytic Jacobians."""
    for kw, analytic in ((dict(d=2, T=8, O=3), False), (dict(d=2, T=8, O=3), True), (dict(d=3, T=6, O=2), False),
                         (dict(d=2, T=8, O=3, vel_limit=0.45), False), (dict(d=2, T=8, O=3, joint_limit=0.15, groups="split"), False)):
        arrays, probs = af.make_batch(8, K=1, point=True, **kw)
        res = sb.solve_batch(arrays, analytic_jac=analytic)
        _compare(res, probs, range(8), analytic=analytic)
    arrays, probs = af.make_batch(300, d=2, T=20, K=1, O=3, point=True)           # more problems than CUs: round selection
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(0, 300, 43))
    ok = res.success
    assert ok.sum() > 100
    x = res.x.reshape(300, 20, 2)
    for b in np.nonzero(ok)[0][:50]:
        dist = np.linalg.norm(x[b][:, None, :] - arrays["obstacles"][b][None, :, :2], axis=2)
        assert np.all(dist >= arrays["obstacles"][b][None, :, 2] - 1e-3)          # success = outside every disc


def test_point_robot_family_matches_reference_golden_runs(gpu):
    import sys
    sys.path.insert(0, GOLD)
    from point_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_point.npz"))
    for prefix, kw, i in CASES:
        arrays, _ = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, prefix
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix
        assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(nq)], prefix


def test_point_robot_family_descriptor_rules(gpu):
    for bad in (dict(n_points=2), dict(dof=1), dict(ee_cost=True)):
        kw = dict(batch=2, dof=2, horizon=6, n_points=1, n_obstacles=2, point=True); kw.update(bad)
        with pytest.raises((_lib.ScoHipError, ValueError)):
            sb.TrajOptBatch(kw["batch"], kw["dof"], kw["horizon"], kw["n_points"], kw["n_obstacles"], point=True,
                            ee_cost=bool(kw.get("ee_cost")))


def test_quadratic_row_family_matches_oracle(gpu):
    """SCO_FAM_STATE_QUADRATIC: general quadratic rows on the state of a timestep (keep-out ellipsoids = concave rows, a
    half-space, a keep-in ball; no kinematics), numeric and analytic Jacobians, with velocity limits, a box and groups, in 2
    to 5 dimensions; a batch above the CU count runs under round selection."""
    for kw, analytic in ((dict(d=2, T=8, O=3), False), (dict(d=2, T=8, O=3), True), (dict(d=3, T=6, O=4), False),
                         (dict(d=5, T=5, O=4), False), (dict(d=2, T=8, O=3, vel_limit=0.5), False),
                         (dict(d=2, T=8, O=3, joint_limit=0.2, groups="halves"), False)):
        arrays, probs = af.make_batch(8, K=1, quadratic=True, **kw)
        res = sb.solve_batch(arrays, analytic_jac=analytic)
        _compare(res, probs, range(8), analytic=analytic)
    arrays, probs = af.make_batch(280, d=2, T=12, K=1, O=3, quadratic=True)
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(0, 280, 41))
    for b in np.nonzero(res.success)[0][:40]:                                     # success = every row satisfied
        x = res.x[b].reshape(12, 2)
        g = np.array([af.quad_rows(x[t], arrays["quad_Q"][b], arrays["quad_a"][b], arrays["quad_c"][b]) for t in range(12)])
        assert g.max() < 1e-3


def test_quadratic_row_family_matches_reference_golden_runs(gpu):
    import sys
    sys.path.insert(0, GOLD)
    from quad_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_quad.npz"))
    for prefix, kw, i in CASES:
        arrays, _ = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, prefix
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix
        assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(nq)], prefix


def test_quadratic_row_family_call_order_and_validation(gpu):
    arrays, _ = af.make_batch(2, d=2, T=6, K=1, O=3, quadratic=True)
    with sb.TrajOptBatch(2, 2, 6, 1, 3, quadratic=True) as tb:
        with pytest.raises(ValueError):
            tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                    arrays["point_frac"], arrays["obstacles"])                    # coefficients missing
        lib = _lib.load()
        tb2 = sb.TrajOptBatch(2, 2, 6, 1, 3, quadratic=True)
        assert lib.sco_sqp_load(tb2._h, _lib.dptr(arrays["x0"]), _lib.dptr(arrays["start"]), _lib.dptr(arrays["goal"]),
                                _lib.dptr(arrays["link_len"]), _lib.iptr(arrays["point_link"]), _lib.dptr(arrays["point_frac"]),
                                _lib.dptr(arrays["obstacles"])) == 0
        with pytest.raises(_lib.ScoHipError):
            tb2.solve()                                                           # sco_sqp_load_quadratic not called yet
        Q = arrays["quad_Q"].copy(); Q[0, 0, 0, 1] += 1.0                         # not symmetric
        assert lib.sco_sqp_load_quadratic(tb2._h, _lib.dptr(Q), _lib.dptr(arrays["quad_a"]), _lib.dptr(arrays["quad_c"])) != 0
        tb2.close()
    with sb.TrajOptBatch(2, 7, 6, 2, 2) as arm:                                   # an arm batch has no quadratic rows
        assert _lib.load().sco_sqp_load_quadratic(arm._h, _lib.dptr(arrays["quad_Q"]), _lib.dptr(arrays["quad_a"]),
                                                  _lib.dptr(arrays["quad_c"])) != 0
    for bad in (dict(n_points=2), dict(dof=17)):
        kw = dict(dof=2, n_points=1); kw.update(bad)
        with pytest.raises(_lib.ScoHipError):
            sb.TrajOptBatch(2, kw["dof"], 6, kw["n_points"], 3, quadratic=True)


def test_program_family_matches_oracle(gpu):
    """SCO_FAM_STATE_PROGRAM: rows as closed-form postfix programs over the state and per-problem parameters (rippled
    discs, a wavy wall, an exponential bump; sco_py_amd.rowexpr), Jacobians by the device's central differences -- in 2
    and 3 dimensions, with velocity limits, a box and groups, and above the CU count under round selection."""
    for kw in (dict(d=2, T=8), dict(d=3, T=6), dict(d=2, T=8, vel_limit=0.5), dict(d=2, T=8, joint_limit=0.25, groups="split")):
        arrays, probs = af.make_batch(8, K=1, program=True, **kw)
        res = sb.solve_batch(arrays)
        _compare(res, probs, range(8))
    arrays, probs = af.make_batch(270, d=2, T=10, K=1, program=True)
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(0, 270, 53))
    prog = arrays["row_program"]
    for b in np.nonzero(res.success)[0][:30]:                                     # success = every row satisfied
        x = res.x[b].reshape(10, 2)
        assert max(prog.evaluate(x[t], arrays["row_params"][b]).max() for t in range(10)) < 1e-3


def test_program_family_matches_reference_golden_runs(gpu):
    import sys
    sys.path.insert(0, GOLD)
    from prog_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_prog.npz"))
    for prefix, kw, i in CASES:
        arrays, _ = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, prefix
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix
        assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(nq)], prefix


def test_program_family_validation(gpu):
    import ctypes as C
    arrays, _ = af.make_batch(2, d=2, T=6, K=1, program=True)
    prog = arrays["row_program"]
    with sb.TrajOptBatch(2, 2, 6, 1, prog.n_rows, program=True) as tb:
        lib = _lib.load()
        assert lib.sco_sqp_load(tb._h, _lib.dptr(arrays["x0"]), _lib.dptr(arrays["start"]), _lib.dptr(arrays["goal"]),
                                _lib.dptr(arrays["link_len"]), _lib.iptr(arrays["point_link"]), _lib.dptr(arrays["point_frac"]),
                                _lib.dptr(arrays["obstacles"])) == 0
        with pytest.raises(_lib.ScoHipError):
            tb.solve()                                                            # no program yet
        par = np.ascontiguousarray(arrays["row_params"])

        def load(words, row_ptr=prog.row_ptr, npar=prog.n_params):
            w = np.ascontiguousarray(words, dtype=np.int32)
            return lib.sco_sqp_load_program(tb._h, len(w), _lib.iptr(np.ascontiguousarray(w.ravel())), _lib.iptr(np.ascontiguousarray(row_ptr, dtype=np.int32)),
                                            len(prog.consts), _lib.dptr(prog.consts), npar, _lib.dptr(par))
        bad = prog.words.copy(); bad[0] = (1, 7)                                  # x[7] of a 2-dimensional state
        assert load(bad) != 0
        bad = prog.words.copy(); bad[0] = (4, 0)                                  # ADD on an empty stack
        assert load(bad) != 0
        bad = prog.words.copy(); bad[prog.row_ptr[1] - 1] = (8, 0)                # a row without its END
        assert load(bad) != 0
        bad = prog.words.copy(); bad[1] = (99, 0)                                 # unknown opcode
        assert load(bad) != 0
        assert load(prog.words, npar=3) != 0                                      # the program reads parameter 12
        # row_ptr is checked as a whole before any word is read through it (r02 advisor): not increasing, beyond the words
        rp = prog.row_ptr.copy(); rp[1] = 10 ** 6
        assert load(prog.words, row_ptr=rp) != 0
        rp = prog.row_ptr.copy(); rp[2] = rp[1]
        assert load(prog.words, row_ptr=rp) != 0
        assert load(prog.words) == 0
        tb.solve()
        first = tb.fetch().x.copy()
        for _ in range(3):                                                        # reloads reuse the handle's buffers
            assert load(prog.words) == 0
        tb.solve()
        assert np.array_equal(tb.fetch().x, first)


def test_joint_limits_7x20_batch_and_validation(gpu):
    arrays, probs = af.make_batch(4, joint_limit=0.2)
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(2))
    # 1100 rows, 9 operand pairs per column: the three-row-slot instantiation of the row-local tier (r02)
    arrays, probs = af.make_batch(2, joint_limit=0.2, vel_limit=0.3)
    _compare(sb.solve_batch(arrays), probs, range(1))
    with sb.TrajOptBatch(1, 3, 6, 2, 2, joint_limits=True) as tb:
        a = af.make_batch(1, d=3, T=6, K=2, O=2, joint_limit=0.5)[0]
        with pytest.raises(ValueError):
            tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"])
        tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
                jlo=a["jlo"], jhi=a["jhi"])
        with pytest.raises(_lib.ScoHipError) as e:
            tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
                    jlo=a["jhi"], jhi=a["jlo"])                           # lo >= hi
        assert e.value.code == -1
    with sb.TrajOptBatch(1, 3, 6, 2, 2) as tb:
        with pytest.raises(_lib.ScoHipError) as e:                           # no joint limits in this family
            _lib.check(_lib.load().sco_sqp_load_joint_limits(tb._h, _lib.dptr(np.zeros((1, 3))), _lib.dptr(np.ones((1, 3)))))
        assert e.value.code == -1


@pytest.mark.parametrize("seed", range(6))
def test_random_feature_mixes_with_joint_limits(gpu, seed):
    rng = np.random.default_rng(5000 + seed)
    d, T, K, O = int(rng.integers(1, 6)), int(rng.integers(2, 9)), int(rng.integers(1, 4)), int(rng.integers(1, 4))
    kw = dict(d=d, T=T, K=K, O=O, reach=bool(rng.integers(2)), joint_limit=float(rng.uniform(0.05, 0.6)))
    if rng.integers(2):
        kw["vel_limit"] = float(rng.uniform(0.4, 1.5))
    if rng.integers(2):
        kw["groups"] = ["halves", "split"][int(rng.integers(2))]
    knobs = dict(compound_penalty=int(rng.integers(2)), duplicate_rows=int(rng.integers(2)))
    arrays, probs = af.make_batch(3, first=int(rng.integers(0, 50)), **kw)
    res = sb.solve_batch(arrays, params=_lib.default_sqp_params(max_sqp_iters=40, **knobs))
    op = sr.SolverParams(compound_penalty=bool(knobs["compound_penalty"]), duplicate_rows=bool(knobs["duplicate_rows"]),
                         max_qp_solves=40)
    _compare(res, probs, range(3), op)


def test_handles_are_independent_also_across_host_threads(gpu):
    """Two handles alive at once, solved from two host threads at the same time (ctypes releases the GIL; every
    handle has its own stream and its own device buffers): same results as one after the other."""
    import threading
    a1, _ = af.make_batch(24, **SMALL)
    a2, _ = af.make_batch(5)
    want1, want2 = sb.solve_batch(a1), sb.solve_batch(a2)
    got = {}
    with sb.TrajOptBatch(24, 3, 6, 2, 2) as t1, sb.TrajOptBatch(5, 7, 20, 5, 2) as t2:
        t1.load(a1["x0"], a1["start"], a1["goal"], a1["link_len"], a1["point_link"], a1["point_frac"], a1["obstacles"])
        t2.load(a2["x0"], a2["start"], a2["goal"], a2["link_len"], a2["point_link"], a2["point_frac"], a2["obstacles"])

        def run(key, tb):
            for _ in range(3):
                tb.solve()
            got[key] = tb.fetch()
        th = [threading.Thread(target=run, args=(1, t1)), threading.Thread(target=run, args=(2, t2))]
        for t in th: t.start()
        for t in th: t.join()
    for want, res in ((want1, got[1]), (want2, got[2])):
        assert np.array_equal(want.x, res.x) and np.array_equal(want.admm_iters, res.admm_iters)
        assert np.array_equal(want.success, res.success)


def test_non_finite_problem_data_fails_that_problem_only(gpu):
    """A NaN obstacle makes every constraint value of its problem NaN: its QPs are reported non-convex / unsolved,
    the problem ends unsuccessful, the other problems of the batch are not touched and nothing hangs."""
    arrays, probs = af.make_batch(4, **SMALL)
    arrays["obstacles"] = arrays["obstacles"].copy()
    arrays["obstacles"][2, 0, 0] = np.nan
    res = sb.solve_batch(arrays, params=_lib.default_sqp_params(max_sqp_iters=30))
    assert not bool(res.success[2])
    _compare(res, probs, [0, 1, 3])


def test_failed_solve_leaves_no_stale_result_and_bad_settings_fail_before_any_launch(gpu):
    """A call that is going to fail touches nothing on the device and does not leave an older result readable."""
    arrays, probs = af.make_batch(2, **SMALL)
    with sb.TrajOptBatch(2, SMALL["d"], SMALL["T"], SMALL["K"], SMALL["O"]) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        tb.solve()
        good = tb.fetch()
        for bad in (dict(max_iter=0), dict(rho=0.0), dict(sigma=-1.0), dict(alpha=2.5),
                    dict(adaptive_rho=1, adaptive_rho_tolerance=0.5)):
            with pytest.raises(_lib.ScoHipError):
                tb.solve(None, _lib.default_qp_settings(**bad))
            with pytest.raises(_lib.ScoHipError):      # the earlier result is gone, not mixed with a half-run one
                tb.fetch()
        with pytest.raises(_lib.ScoHipError):
            tb.solve(_lib.default_sqp_params(initial_trust_region_size=0.0))
        tb.solve()
        again = tb.fetch()
        assert np.array_equal(good.x, again.x) and np.array_equal(good.sqp_iters, again.sqp_iters)


def test_entry_points_put_the_callers_device_back(gpu):
    """Every ABI call runs on its handle's device and restores the caller's current device (one device here: the
    check is that the current device is unchanged and HIP state stays usable around create/solve/destroy)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    cur = C.c_int(-1)
    assert hip.hipGetDevice(C.byref(cur)) == 0
    before = cur.value
    arrays, _ = af.make_batch(1, **SMALL)
    sb.solve_batch(arrays)
    assert hip.hipGetDevice(C.byref(cur)) == 0 and cur.value == before


# ---- non-quadratic objective terms (SCO_FAM_FLAG_EE_COST): degree-2 convexification on the device --------------
def _obj_cases():
    import sys
    sys.path.insert(0, GOLD)
    from obj_cases import CASES
    return CASES


def _params_from(attrs):
    return (_lib.default_sqp_params(**attrs) if attrs else None), (sr.SolverParams(**attrs) if attrs else None)


@pytest.mark.parametrize("case", range(5))
def test_non_quadratic_objective_matches_reference_golden_run(gpu, case):
    """Prob.add_obj_expr on a plain Expr (prob.py:88-104): numeric Hessian, eigenvalue shift, numeric gradient, the model
    in P and q, the true value in the merit (expr.py:102-156; prob.py:532-534, 571-573, 625-626) -- on the device, against
    runs of the reference's own modules (tests/golden/make_golden_obj.py): same number of QPs with the same statuses,
    trajectory within the 1e-6 of north_star."""
    name, kw, attrs = _obj_cases()[case]
    g = np.load(os.path.join(GOLD, "trajopt_obj.npz"))
    kw = dict(kw); first = kw.pop("i")
    arrays, probs = af.make_batch(1, first=first, **kw)
    dp, _ = _params_from(attrs)
    res = sb.solve_batch(arrays, params=dp)
    n_qp = int(g[name + "_n_qp"])
    assert res.qp_solves[0] == n_qp
    assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%s_qp%d_status" % (name, k)]) for k in range(n_qp)]
    assert bool(res.success[0]) == bool(g[name + "_success"])
    assert np.abs(res.x[0] - g[name + "_x"]).max() < TOL
    assert abs(res.max_violation[0] - float(g[name + "_max_violation"])) < 1e-6


def test_non_quadratic_objective_batch_follows_the_oracle(gpu):
    """A batch of 12 problems with objective terms against the flat oracle: decisions, QP statuses, merits; ADMM
    iteration counts may differ by one termination check where a residual sits on the tolerance (the numeric Hessian
    amplifies last-bit differences of f by 1/h^2, expr.py:102-109), so they are compared per QP to within 25."""
    arrays, probs = af.make_batch(12, first=20, ee_cost_weight=1.5, **SMALL)
    kw = dict(max_merit_coeff_increases=2, initial_penalty_coeff=10.0)
    res = sb.solve_batch(arrays, params=_lib.default_sqp_params(**kw))
    for b in range(12):
        ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), sr.SolverParams(**kw), emulate_memo=True)
        tr, rt = res.trace[b], ref.trace[:64]
        assert tr.shape == rt.shape and np.array_equal(tr[:, 0], rt[:, 0]), (b, tr[:, 0], rt[:, 0])
        assert np.array_equal(tr[:, 6], rt[:, 6]) and np.abs(tr[:, 7] - rt[:, 7]).max() <= 25, b
        assert np.abs(tr[:, 1:4] - rt[:, 1:4]).max() < 1e-6 * (1 + np.abs(rt[:, 1:4]).max()), b
        assert np.abs(res.x[b] - ref.x).max() < TOL and bool(res.success[b]) == ref.success


def test_non_quadratic_objective_at_7x20(gpu):
    """BASELINE configs[2] shape with the objective terms: 4 problems against the oracle (the QP's P now holds a dense
    7 x 7 block per timestep; the row-local ADMM tier still takes it)."""
    arrays, probs = af.make_batch(4, first=3, ee_cost_weight=0.5)
    kw = dict(compound_penalty=0, duplicate_rows=0, max_sqp_iters=6)
    res = sb.solve_batch(arrays, params=_lib.default_sqp_params(**kw))
    for b in range(4):
        ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), sr.SolverParams(compound_penalty=False, duplicate_rows=False,
                                                                       max_qp_solves=6), emulate_memo=True)
        n = min(len(res.trace[b]), len(ref.trace))
        assert n >= 3 and np.array_equal(res.trace[b][:n, 0], ref.trace[:n, 0]), (b, res.trace[b][:, 0], ref.trace[:, 0])
        assert np.abs(res.trace[b][:n, 1:4] - ref.trace[:n, 1:4]).max() < 1e-6 * (1 + np.abs(ref.trace[:n, 1:4]).max())


def test_objective_family_call_order_and_limits(gpu):
    with sb.TrajOptBatch(1, 3, 6, 2, 2, ee_cost=True) as tb:
        arrays, _ = af.make_batch(1, ee_cost_weight=1.0, **SMALL)
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"], cost_weight=arrays["cost_weight"], cost_target=arrays["cost_target"])
        tb.solve()
        with pytest.raises(_lib.ScoHipError):
            tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                    arrays["point_frac"], arrays["obstacles"], cost_weight=-1.0, cost_target=arrays["cost_target"])
    with pytest.raises(_lib.ScoHipError):
        sb.TrajOptBatch(1, 17, 4, 1, 1, ee_cost=True)          # the block Hessian is handled per thread: dof <= 16


@pytest.mark.parametrize("switch", ["SCO_QP_RL_ALIGNED", "SCO_QP_RL_LAY8"])
def test_opt_in_layouts_through_the_sliced_sqp_loop(gpu, monkeypatch, switch):
    """The opt-in layouts of the row-local ADMM kernel (r03, csrc/sco_admm_rl.hip: aligned closed assignment with one
    barrier per iteration; 3 x 18 W tiles) under the device loop: parked / resumed solves (time slices of 1000 iterations),
    cold and slice-free runs agree bit for bit with each other, and with the oracle decision for decision."""
    monkeypatch.setenv(switch, "1")
    arrays, probs = af.make_batch(6)
    outs = []
    with sb.TrajOptBatch(6, 7, 20, 5, 2) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        for sl in (-1, 1000):
            tb.solve(_lib.default_sqp_params(admm_slice=sl))
            r = tb.fetch(); r.trace = tb.trace()
            outs.append(r)
    assert np.array_equal(outs[0].x, outs[1].x) and np.array_equal(outs[0].admm_iters, outs[1].admm_iters)
    _compare(outs[1], probs, range(2))


@pytest.mark.parametrize("variant,kw", [("sweep", dict(d=2, T=8)), ("sweep", dict(d=3, T=6)), ("dynamics", dict(d=3, T=8)),
                                        ("curve", dict(d=3, T=8)), ("attract", dict(d=2, T=8)),
                                        ("sweep", dict(d=2, T=8, vel_limit=0.6, groups="split")),
                                        ("dynamics", dict(d=3, T=8, joint_limit=0.4, groups="halves"))])
@pytest.mark.parametrize("analytic", [False, True])
def test_wide_program_blocks_match_oracle(gpu, variant, kw, analytic):
    """r03 extensions of SCO_FAM_STATE_PROGRAM on the device: constraint blocks on two consecutive timesteps ("sweep":
    swept-volume style keep-outs; "dynamics": the unicycle step as EQUALITY rows -> abs penalty with two slacks per row,
    /root/reference/sco_py/sco_osqp/prob.py:280-315), an equality row on one timestep ("curve"), a non-quadratic objective
    program per timestep ("attract": prob.py:88-104, expr.py:143-153), each with numeric (Richardson central differences)
    and with forward-mode analytic Jacobians (expr.py:86-100) -- decision for decision against the oracle."""
    if variant == "attract" and analytic:
        pytest.skip("the objective term is always differentiated numerically (as in the reference: Expr without grad / hess)")
    arrays, probs = af.make_batch(6, K=1, program=True, variant=variant, **kw)
    res = sb.solve_batch(arrays, analytic_jac=analytic)
    _compare(res, probs, range(6), analytic=analytic)


def test_wide_program_blocks_match_reference_golden_runs(gpu):
    """The same extensions against runs of the REFERENCE's own modules (tests/golden/make_golden_prog.py ->
    trajopt_prog2.npz): trajectory to 1e-6, success flag, and the status of every QP."""
    import sys
    sys.path.insert(0, GOLD)
    from prog_cases import CASES2
    g = np.load(os.path.join(GOLD, "trajopt_prog2.npz"))
    for prefix, kw, i, aj in CASES2:
        arrays, _ = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays, analytic_jac=aj)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, (prefix, np.abs(res.x[0] - g[prefix + "x"]).max())
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix
        if arrays["row_program"].n_eq == 0 and not arrays["row_program"].objective:
            # (equality rows: the reference's slack column order is another one, an ADMM run that stops on max_iter may
            # end one termination check apart; objective terms: numeric Hessians)
            assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(nq)], prefix


def test_wide_program_blocks_above_the_cu_count_and_descriptor_rules(gpu):
    """300 dynamics problems (round selection, time slices) sampled against the oracle; success means every row of every
    block holds at the returned trajectory; descriptor rules of the extensions."""
    arrays, probs = af.make_batch(300, d=3, T=8, K=1, program=True, variant="dynamics")
    res = sb.solve_batch(arrays)
    _compare(res, probs, range(0, 300, 61))
    prog = arrays["row_program"]
    for b in np.nonzero(res.success)[0][:40]:
        x = res.x[b]
        for t in range(8 - 1):
            g = prog.evaluate(x[t * 3:(t + 2) * 3], arrays["row_params"][b])
            assert g[prog.ineq_rows].max() < 1e-3 and np.abs(g[prog.eq_rows]).max() < 1e-3
    from sco_py_amd.rowexpr import X, compile_rows
    with pytest.raises(_lib.ScoHipError):                       # span x dof beyond 32 state coordinates
        sb.TrajOptBatch(1, 20, 6, 1, 1, program=compile_rows([X(0) + X(39)], span=2))       # 2 x 20 > 32 state coordinates
    with pytest.raises(ValueError):
        compile_rows([X(0)], objective=X(1), span=2)           # objective terms live on one timestep
    import ctypes as C
    lib = _lib.load()
    for fam, span, neq in ((sb.SCO_FAM_ARM_CIRCLES, 2, 0), (sb.SCO_FAM_ARM_CIRCLES, 1, 1), (sb.SCO_FAM_STATE_QUADRATIC, 2, 0),
                           (sb.SCO_FAM_STATE_PROGRAM, 5, 0), (sb.SCO_FAM_STATE_PROGRAM, 1, 5),       # (span 3 and 4: allowed since r04)
                           (sb.SCO_FAM_STATE_PROGRAM | sb.SCO_FAM_FLAG_OBJ_PROGRAM, 2, 0)):
        h = C.c_void_p()
        desc = _lib.TrajoptDesc(1, 3, 6, 1, 2, fam, 0, 2, span, neq)
        assert lib.sco_sqp_create(0, C.byref(desc), C.byref(h)) == -1, (fam, span, neq)      # SCO_ERR_ARG
    h = C.c_void_p()                                           # equality rows are open to the quadratic-row family too
    desc = _lib.TrajoptDesc(1, 3, 6, 1, 2, sb.SCO_FAM_STATE_QUADRATIC, 0, 2, 1, 1)
    assert lib.sco_sqp_create(0, C.byref(desc), C.byref(h)) == 0
    assert lib.sco_sqp_destroy(h) == 0


def test_quadratic_rows_with_an_equality_row(gpu):
    """SCO_FAM_STATE_QUADRATIC with n_eq_rows = 1 (r03): the last row of every timestep is an EQUALITY (the state stays on a
    sphere through start and goal; EqExpr on a quadratic Expr -> abs penalty with two slacks per row,
    /root/reference/sco_py/sco_osqp/prob.py:280-315) -- against the oracle decision for decision (numeric and analytic
    Jacobians) and against runs of the reference's own modules (tests/golden/trajopt_quad2.npz)."""
    for kw, aj in ((dict(d=2, T=8, K=1, O=3), False), (dict(d=3, T=6, K=1, O=4), True), (dict(d=2, T=8, K=1, O=3, groups="split"), False)):
        arrays, probs = af.make_batch(6, quadratic=True, n_eq=1, **kw)
        res = sb.solve_batch(arrays, analytic_jac=aj)
        _compare(res, probs, range(6), analytic=aj)
    import sys
    sys.path.insert(0, GOLD)
    from quad_cases import CASES2
    g = np.load(os.path.join(GOLD, "trajopt_quad2.npz"))
    for prefix, kw, i, aj in CASES2:
        arrays, _ = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays, analytic_jac=aj)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, prefix
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix


def test_wavefront_rounds_then_row_local_tail_agree_with_the_row_local_loop(gpu, monkeypatch):
    """r04: with at least 3.3 live problems per CU a round of the device loop runs on the wavefront tier (four problems per
    CU, csrc/sco_admm_wv.hip), below that on the row-local kernel; a QP changes kernel at a slice boundary through the
    common parked state.  The two kernels agree to rounding, so against the all-row-local loop (SCO_QP_NO_WV=1) and the
    oracle: same decisions, QP statuses and ADMM iteration counts, trajectories within 1e-9 (the north_star bar is 1e-6)."""
    import ctypes as C
    lib = _lib.load()
    lib.sco_debug_sqp_wv_rounds.restype = C.c_int; lib.sco_debug_sqp_wv_rounds.argtypes = [C.c_void_p]
    nb, dims = 1280, (3, 6, 2, 2)
    arrays, probs = af.make_batch(nb, d=dims[0], T=dims[1], K=dims[2], O=dims[3])
    outs, wvr = [], []
    for no_wv in ("0", "1"):
        monkeypatch.setenv("SCO_QP_NO_WV", no_wv)
        with sb.TrajOptBatch(nb, *dims) as tb:
            tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                    arrays["point_frac"], arrays["obstacles"])
            tb.solve(_lib.default_sqp_params(admm_slice=400))
            r = tb.fetch(); r.trace = tb.trace(); r.timing = tb.last_timing()
            outs.append(r); wvr.append(lib.sco_debug_sqp_wv_rounds(tb._h))
    assert wvr[0] > 0 and wvr[1] == 0 and outs[0].timing["rounds"] > wvr[0]       # wavefront rounds first, then the tail
    a, b = outs
    assert np.array_equal(a.admm_iters, b.admm_iters) and np.array_equal(a.success, b.success) and np.array_equal(a.qp_solves, b.qp_solves)
    assert all(np.array_equal(x[:, [0, 4, 5, 6, 7]], y[:, [0, 4, 5, 6, 7]]) for x, y in zip(a.trace, b.trace))
    assert np.abs(a.x - b.x).max() < 1e-9
    _compare(a, probs, range(0, nb, 160))


@pytest.mark.parametrize("kw,params", [
    (dict(d=3, T=6, K=2, O=2), {}), (dict(d=3, T=6, K=2, O=2), dict(compound_penalty=0, duplicate_rows=0, max_sqp_iters=20)),
    (dict(), {}), (dict(d=2, T=8, O=3, point=True), {}), (dict(d=2, T=8, O=3, quadratic=True), {}),
    (dict(d=2, T=8, K=1, program=True), {}), (dict(d=3, T=6, K=2, O=2, ee_cost_weight=0.5), {}),
    (dict(T=12), {}), (dict(d=2, T=20, K=1, program=True, per_step=True, obj_weights=True), {})],
    ids=["3x6", "3x6 quirks off", "7x20", "point", "quadratic rows", "program rows", "objective terms", "7x12", "program 2x20 steps weights"])
def test_wavefront_tier_forced_through_the_device_loop(gpu, monkeypatch, kw, params):
    """By default a round goes to the wavefront tier only with > 3.3 live problems per CU; SCO_WV_MIN_PER_CU=0 sends every
    round of every batch there: decisions, QP statuses, iteration counts and trajectories of the flat oracle for the
    families whose penalty QP the tier takes (dense P blocks of the objective terms included), time slices of 300."""
    monkeypatch.setenv("SCO_WV_MIN_PER_CU", "0")
    nb = 6 if (not kw or kw.get("T") == 12) else 12
    arrays, probs = af.make_batch(nb, **kw)
    dp = _lib.default_sqp_params(admm_slice=300, **params)
    op = sr.SolverParams(compound_penalty=False, duplicate_rows=False, max_qp_solves=20) if params else None
    res = sb.solve_batch(arrays, params=dp)
    if kw.get("ee_cost_weight") is None:
        _compare(res, probs, range(0, nb, 2), op)
    else:       # numeric Hessians: iteration counts of QPs that stop on max_iter may differ by one check
        for b in range(0, nb, 2):
            ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), emulate_memo=True)
            assert np.array_equal(res.trace[b][:, 0], ref.trace[:64, 0]) and np.abs(res.x[b] - ref.x).max() < TOL


def test_program_reload_with_another_size_and_the_n_eq_rows_check(gpu):
    """sco_sqp_load_program may be called again with a program of ANOTHER size (buffers are reallocated; r03 advisor: between
    the free and the last upload the handle has no program -- a solve is refused, not run on freed buffers); the result is
    that of a fresh handle.  TrajOptBatch refuses an n_eq_rows that contradicts the compiled program."""
    from sco_py_amd.rowexpr import X, P, sqrt, compile_rows
    d, T, B = 2, 8, 4
    arrays, _ = af.make_batch(B, d=d, T=T, K=1, program=True)
    short = compile_rows([P(2) - sqrt((X(0) - P(0)) ** 2 + (X(1) - P(1)) ** 2 + 1e-12)] * 4)        # 4 rows again, fewer words
    assert len(short.words) != len(arrays["row_program"].words)
    par = arrays["row_params"][:, :3].copy()
    outs = []
    with sb.TrajOptBatch(B, d, T, 1, 4, program=arrays["row_program"]) as tb:
        args = (arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
        tb.load(*args, row_program=arrays["row_program"], row_params=arrays["row_params"])
        tb.solve(); first = tb.fetch()
        tb.load(*args, row_program=short, row_params=par)              # another word count: reallocation
        tb.solve(); outs.append(tb.fetch())
        tb.load(*args, row_program=arrays["row_program"], row_params=arrays["row_params"])
        tb.solve(); again = tb.fetch()
    with sb.TrajOptBatch(B, d, T, 1, 4, program=short) as tb2:
        tb2.load(*args, row_program=short, row_params=par)
        tb2.solve(); outs.append(tb2.fetch())
    assert np.array_equal(outs[0].x, outs[1].x) and np.array_equal(first.x, again.x) and not np.array_equal(first.x, outs[0].x)
    with pytest.raises(ValueError):
        sb.TrajOptBatch(B, d, T, 1, 4, program=arrays["row_program"], n_eq_rows=2)


# ---- r04: the wider device template: weighted smoothing objective, per-timestep program parameters ----
@pytest.mark.parametrize("kw,analytic", [
    (dict(d=3, T=6, K=2, O=2, obj_weights=True), False), (dict(d=3, T=6, K=2, O=2, obj_weights=True, reach=True, vel_limit=0.6), False),
    (dict(d=3, T=6, K=2, O=2, obj_weights=True, ee_cost_weight=1.0), False), (dict(d=2, T=8, K=1, O=3, point=True, obj_weights=True, joint_limit=0.2), True),
    (dict(d=2, T=8, K=1, program=True, per_step=True), False), (dict(d=2, T=8, K=1, program=True, variant="sweep", per_step=True), True),
    (dict(d=3, T=8, K=1, program=True, variant="dynamics", per_step=True), False),
    (dict(d=2, T=8, K=1, program=True, variant="attract", per_step=True, obj_weights=True), False),
    (dict(obj_weights=True), False),
    (dict(d=2, T=8, K=1, program=True, variant="accel"), False), (dict(d=3, T=8, K=1, program=True, variant="accel", per_step=True, vel_limit=0.6), True),
    (dict(d=2, T=9, K=1, program=True, variant="jerk"), False), (dict(d=2, T=9, K=1, program=True, variant="jerk", groups="split", obj_weights=True), True),
    (dict(d=3, T=6, K=2, O=2, lin_rows=True), False), (dict(d=3, T=6, K=2, O=2, lin_rows=True, vel_limit=0.6, joint_limit=0.3, obj_weights=True, reach=True), True),
    (dict(d=2, T=8, K=1, program=True, lin_rows=True, per_step=True), False), (dict(lin_rows=True), False),
    (dict(d=2, T=8, K=1, program=True, circles=2), False), (dict(d=2, T=8, K=1, program=True, variant="attract", circles=1, per_step=True, groups="split"), False),
    (dict(d=3, T=6, K=1, program=True, circles=2, lin_rows=True, obj_weights=True), True), (dict(d=2, T=20, K=1, program=True, circles=3), False),
    (dict(d=3, T=6, K=2, O=2, acc_weights=True), False), (dict(d=3, T=6, K=2, O=2, acc_weights=True, obj_weights=True, ee_cost_weight=1.0, vel_limit=0.6), False),
    (dict(d=2, T=8, K=1, program=True, acc_weights=True, per_step=True), True), (dict(acc_weights=True), False)],
    ids=["weights", "weights-reach-vel", "weights-objterm", "weights-point-jl-analytic", "steps", "steps-sweep-analytic", "steps-dynamics",
         "steps-attract-weights", "weights-7x20", "span3", "span3-steps-vel-analytic", "span4", "span4-groups-weights-analytic",
         "rows", "rows-vel-jl-weights-reach-analytic", "rows-program-steps", "rows-7x20",
         "two-kinds", "two-kinds-attract-steps-groups", "two-kinds-rows-weights-analytic", "two-kinds-2x20",
         "acc", "acc-weights-objterm-vel", "acc-program-steps-analytic", "acc-7x20"])
def test_wider_template_matches_the_oracle(gpu, kw, analytic):
    """sco_sqp_load_obj_weights: sum_t sum_j w_j (x[t+1][j] - x[t][j])^2 with per-problem, per-joint weights (objective value, P
    of every QP, the degree-2 model of an objective term on top of it); sco_sqp_load_program_steps: block t and the objective
    term of timestep t evaluated with params[problem][t] (values, finite-difference and forward-mode Jacobians, numeric
    Hessians); span 3 and 4: constraint blocks on three / four consecutive timesteps (acceleration and jerk limits, a keep-out
    on the centroid of four points); sco_sqp_create_rows / sco_sqp_load_linear_rows: general affine rows (a shared CSR pattern,
    coefficients and right-hand sides per problem; inequalities and an equality) in the projection QP and in every penalty QP;
    sco_sqp_set_circle_rows: two kinds of non-linear rows per timestep (the point's keep-out discs in front of program rows);
    SCO_FAM_FLAG_ACC_COST / sco_sqp_load_acc_weights: an acceleration term in the quadratic objective (second super-diagonal block
    of P; with an objective term on top of it).
    Every decision, QP status, iteration count, merit and the answer against the oracle, which reproduces the
    reference's own runs of these cases (tests/test_golden.py, trajopt_wide.npz)."""
    n = 4 if kw.get("d", 7) == 7 else 8
    arrays, probs = af.make_batch(n, **kw)
    res = sb.solve_batch(arrays, analytic_jac=analytic)
    _compare(res, probs, range(n), analytic=analytic)
    # the extension is live: the same problems without it end elsewhere
    if kw.get("lin_rows"):
        g0 = probs[0]["lin_gen"]
        for b in np.nonzero(res.qp_solves > 1)[0]:                      # (projection feasible) the rows hold at every returned point
            v = probs[b]["lin_gen"]["A"] @ res.x[b] - probs[b]["lin_gen"]["rhs"]
            assert v[g0["is_eq"] == 0].max() < 1e-5 and np.abs(v[g0["is_eq"] != 0]).max() < 1e-5
    if kw.get("obj_weights") or kw.get("per_step") or kw.get("acc_weights"):
        plain = {k: v for k, v in arrays.items() if k not in ("obj_w", "acc_w")}
        if kw.get("per_step"):
            plain["row_params"] = arrays["row_params"][:, 0, :].copy()
        res0 = sb.solve_batch(plain, analytic_jac=analytic)
        assert np.abs(res0.x - res.x).max() > 1e-4


def test_wider_template_matches_reference_golden_runs(gpu):
    """The same against runs of the REFERENCE's own modules (tests/golden/make_golden_wide.py -> trajopt_wide.npz):
    trajectory to 1e-6, success flag, status of every QP, iteration counts where no equality row / objective term is in play."""
    import sys
    sys.path.insert(0, GOLD)
    from wide_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_wide.npz"))
    for prefix, kw, i, aj in CASES:
        arrays, probs = af.make_batch(1, first=i, **kw)
        res = sb.solve_batch(arrays, analytic_jac=aj)
        assert np.abs(res.x[0] - g[prefix + "x"]).max() < TOL, (prefix, np.abs(res.x[0] - g[prefix + "x"]).max())
        assert bool(res.success[0]) == bool(g[prefix + "success"]), prefix
        nq = int(g[prefix + "n_qp"])
        assert [int(v) for v in res.trace[0][:, 6]] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(nq)], prefix
        prog = arrays.get("row_program")
        if not (kw.get("reach") or kw.get("ee_cost_weight") or (prog is not None and (prog.n_eq or prog.objective))):
            assert [int(v) for v in res.trace[0][:, 7]] == [int(g["%sqp%d_iters" % (prefix, k)]) for k in range(nq)], prefix


def test_wider_template_reloads_and_argument_checks(gpu):
    """A handle switches between weighted and plain, shared and per-timestep parameters from one load to the next (results =
    fresh handles'); shapes and values are checked before anything is uploaded."""
    kw = dict(d=2, T=8, K=1, program=True)
    a_plain, _ = af.make_batch(4, **kw)
    a_wide, _ = af.make_batch(4, per_step=True, obj_weights=True, **kw)
    ref_plain, ref_wide = sb.solve_batch(a_plain), sb.solve_batch(a_wide)
    with sb.TrajOptBatch(4, 2, 8, 1, a_plain["O"], program=a_plain["row_program"]) as tb:
        for a, ref in ((a_wide, ref_wide), (a_plain, ref_plain), (a_wide, ref_wide)):
            tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
                    row_program=a["row_program"], row_params=a["row_params"], obj_weights=a.get("obj_w"))
            tb.solve()
            r = tb.fetch()
            assert np.array_equal(r.x, ref.x) and np.array_equal(r.admm_iters, ref.admm_iters)
        a = a_wide
        with pytest.raises(ValueError):
            tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
                    row_program=a["row_program"], row_params=a["row_params"][:, :5, :])
        with pytest.raises(_lib.ScoHipError):
            tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
                    row_program=a["row_program"], row_params=a["row_params"], obj_weights=-np.ones((4, 2)))
        with pytest.raises(ValueError):                                  # no general rows in this handle
            tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
                    row_program=a["row_program"], row_params=a["row_params"], lin_vals=np.ones((4, 3)), lin_rhs=np.ones((4, 2)))
    with pytest.raises(ValueError):                                      # circle rows: program family on single timesteps
        sb.TrajOptBatch(2, 3, 6, 2, 2, circle_rows=1)
    am, _ = af.make_batch(2, d=2, T=8, K=1, program=True, circles=2)
    with pytest.raises(ValueError):                                      # the program supplies O - circle_rows rows
        with sb.TrajOptBatch(2, 2, 8, 1, am["O"], program=am["row_program"], circle_rows=1) as tbm:
            tbm.load(am["x0"], am["start"], am["goal"], am["link_len"], am["point_link"], am["point_frac"], am["obstacles"],
                     row_program=am["row_program"], row_params=am["row_params"])
    # patterns are checked before anything is built: a column out of range, columns not increasing, an empty row
    for rp, ci in (([0, 2], [0, 99]), ([0, 2], [3, 3]), ([0, 0, 1], [1])):
        with pytest.raises(_lib.ScoHipError):
            sb.TrajOptBatch(2, 3, 6, 2, 2, lin_rows=(np.array(rp), np.array(ci), np.zeros(len(rp) - 1, dtype=np.int32)))
    arr, _ = af.make_batch(2, d=3, T=6, K=2, O=2, lin_rows=True)
    with sb.TrajOptBatch(2, 3, 6, 2, 2, lin_rows=arr["lin_rows"]) as tb2:
        tb2.load(arr["x0"], arr["start"], arr["goal"], arr["link_len"], arr["point_link"], arr["point_frac"], arr["obstacles"],
                 lin_vals=arr["lin_vals"], lin_rhs=arr["lin_rhs"])
        with pytest.raises(ValueError):
            tb2.load(arr["x0"], arr["start"], arr["goal"], arr["link_len"], arr["point_link"], arr["point_frac"], arr["obstacles"])
