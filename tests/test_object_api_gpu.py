"""The reference's OBJECT API on the device-resident loop (verdict r03, B2).

A caller of sco_py builds a ``Prob`` from ``Variable`` / ``BoundExpr`` objects and calls ``Solver().solve(prob)``
(/root/reference/sco_py/sco_osqp/solver.py:30-59, 107-253; prob.py:88-144; expr.py:413-437).  These tests do exactly
that with the mirror classes (tests/trajopt_build.py, the construction code the golden generators ran against the
REFERENCE's modules), with the non-linear expressions given as ``sco_py_amd.devexpr`` objects, and check that

* the solve ran in the resident loop (``sco_sqp_*``: rounds > 0) and the Python ``f`` / ``grad`` of no expression was
  ever called,
* the results written back into the ``Variable`` objects, the return value, the status of every QP and
  ``prob.nonconverged_groups`` are those of the golden runs of the reference's own modules.
"""
import os
import sys

import numpy as np
import pytest

import conftest as ct
import trajopt_build as tb
from oracle import arm_family as af
from oracle import sco_ref as sr
from sco_py_amd import devexpr as dx
from sco_py_amd.sco_osqp import batching, compile as sco_compile

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
TOL = 1e-6          # abs, BASELINE.json north_star
SMALL = dict(d=3, T=6, K=2, O=2)


def _device_exprs(prob):
    out = [be.expr.expr for be in prob._nonlin_cnt_exprs] + [be.expr for be in prob._nonquad_obj_exprs]
    assert out and all(isinstance(e, dx.DeviceExpr) for e in out)
    return out


def _solve(pr, attrs=None, analytic=False, **solve_kw):
    mods = ct.mirror_mods()
    prob, traj, step_vars, atoms = tb.build_prob(mods, pr, analytic_jac=analytic, device_exprs=True)
    solver = mods.Solver()
    for k, v in (attrs or {}).items():
        setattr(solver, k, v)
    ok = solver.solve(prob, method="penalty_sqp", **solve_kw)
    assert solver.last_path == "device" and solver.last_device["rounds"] > 0          # the resident loop ran ...
    assert sum(e.host_evals for e in _device_exprs(prob)) == 0                           # ... and Python evaluated nothing
    return ok, prob, traj, step_vars, solver


def _against_golden(g, prefix, ok, traj, solver, iters=True, sep=""):
    nq = int(g[prefix + sep + "n_qp"])
    tr = solver.last_device["traces"][0]
    assert np.abs(traj.get_value().ravel() - g[prefix + sep + "x"]).max() < TOL, prefix
    assert ok == bool(g[prefix + sep + "success"]), prefix
    assert [int(v) for v in tr[:, 6]] == [int(g["%s%sqp%d_status" % (prefix, sep, k)]) for k in range(nq)], prefix
    if iters:
        assert [int(v) for v in tr[:, 7]] == [int(g["%s%sqp%d_iters" % (prefix, sep, k)]) for k in range(nq)], prefix


@pytest.mark.parametrize("i", range(4))
def test_plain_solver_solve_reproduces_reference_golden_run_small(gpu, i):
    """Twin of test_mirror_api_reproduces_reference_qp_sequence (tests/test_golden.py): same Prob construction, plain
    Solver().solve(prob) -- on the device."""
    g = np.load(os.path.join(GOLD, "trajopt_small.npz"))
    ok, prob, traj, step_vars, solver = _solve(af.make_problem(i, **SMALL))
    _against_golden(g, "p%d_" % i, ok, traj, solver)
    # every Variable holding an atom sees the returned point (variable.py:47-60), value == saved value (solver.py:197-251)
    x = traj.get_value().ravel()
    for t, sv in enumerate(step_vars):
        assert np.array_equal(sv.get_value().ravel(), x[t * 3:(t + 1) * 3]) and np.array_equal(sv._saved_value, sv._value)
    assert [a.val for a in traj.get_osqp_vars().ravel()] == x.tolist()
    # the host loop's trace format: (code, merit, model merit, new merit, trust, penalty), first row the projection
    assert solver.trace[0][0] == 0 and all(len(r) == 6 for r in solver.trace)


def test_plain_solver_solve_7x20_golden_and_the_host_loop_agree(gpu):
    """BASELINE configs[1] (one 7-DOF x 20 problem) through Solver.solve: the golden run of the reference's modules, and
    the SAME Prob class solved by the host loop (device_loop = False: Python evaluates, one device QP per optimize)."""
    g = np.load(os.path.join(GOLD, "trajopt_7x20.npz"))
    pr = af.make_problem(0)
    ok, prob, traj, _, solver = _solve(pr)
    assert np.abs(traj.get_value().ravel() - g["p0_x"]).max() < TOL and ok == bool(g["p0_success"])
    assert [int(v) for v in solver.last_device["traces"][0][:, 7]] == [int(g["p0_qp%d_iters" % k]) for k in range(int(g["p0_n_qp"]))]
    assert abs(float(prob.get_max_cnt_violation()) - float(g["p0_max_violation"])) < 1e-7      # evaluated on the written-back point
    mods = ct.mirror_mods()
    prob2, traj2, _, _ = tb.build_prob(mods, pr, device_exprs=True)
    host = mods.Solver(); host.device_loop = False
    ok2 = host.solve(prob2, method="penalty_sqp")
    assert host.last_path == "host" and sum(e.host_evals for e in _device_exprs(prob2)) > 0
    assert ok2 == ok and np.abs(traj2.get_value() - traj.get_value()).max() < TOL
    assert [r[0] for r in host.trace] == [r[0] for r in solver.trace]


def test_reach_velocity_and_joint_limit_goldens(gpu):
    from jl_cases import CASES as JL
    from vel_cases import CASES as VEL
    g = np.load(os.path.join(GOLD, "trajopt_reach.npz"))
    for i in range(3):
        ok, _, traj, _, solver = _solve(af.make_problem(i, reach=True, **SMALL))
        _against_golden(g, "p%d_" % i, ok, traj, solver, iters=False)
    ok, _, traj, _, solver = _solve(af.make_problem(1, reach=True, **SMALL), analytic=True)
    _against_golden(g, "p1a_", ok, traj, solver, iters=False)
    for cases, name in ((VEL, "trajopt_vel.npz"), (JL, "trajopt_jl.npz")):
        g = np.load(os.path.join(GOLD, name))
        for prefix, kw, i in cases:
            pr = af.make_problem(i, **kw)
            ok, prob, traj, step_vars, solver = _solve(pr)
            _against_golden(g, prefix, ok, traj, solver, iters=not kw.get("reach"))
            if prefix in ("x0_", "x1_"):
                # pins and limits contradict each other: the projection QP fails, solve returns False and the variables
                # are untouched (solver.py:81-82)
                assert not ok and np.array_equal(traj.get_value().ravel(), pr["x0"]) and traj._saved_value is None


def test_constraint_group_goldens_incl_the_group_report(gpu):
    from group_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_groups.npz"))
    for prefix, kw, i, knobs in CASES:
        ok, prob, traj, _, solver = _solve(af.make_problem(i, **kw), attrs=knobs)
        _against_golden(g, prefix, ok, traj, solver, iters=not kw.get("reach"))
        want = [str(s) for s in g[prefix + "nonconverged"]]
        assert sorted(set(prob.nonconverged_groups)) == sorted(want), prefix
        # the reference lists the stalled groups first and then every violated group under the threshold
        # (solver.py:209-234): the flat oracle keeps the set, the mirror's host loop the list
        mods = ct.mirror_mods()
        prob2, _, _, _ = tb.build_prob(mods, af.make_problem(i, **kw), device_exprs=True)
        host = mods.Solver(); host.device_loop = False
        for k, v in (knobs or {}).items():
            setattr(host, k, v)
        host.solve(prob2, method="penalty_sqp")
        assert prob.nonconverged_groups == prob2.nonconverged_groups, prefix


def test_objective_term_goldens(gpu):
    from obj_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_obj.npz"))
    for name, kw, attrs in CASES:
        kw = dict(kw); i = kw.pop("i")
        ok, prob, traj, _, solver = _solve(af.make_problem(i, **kw), attrs=attrs)
        _against_golden(g, name, ok, traj, solver, iters=False, sep="_")


def test_point_quadratic_and_program_family_goldens(gpu):
    from point_cases import CASES as PT
    from prog_cases import CASES as PG, CASES2 as PG2
    from quad_cases import CASES as QD, CASES2 as QD2
    for cases, name in ((PT, "trajopt_point.npz"), (QD, "trajopt_quad.npz"), (PG, "trajopt_prog.npz"),
                        (QD2, "trajopt_quad2.npz"), (PG2, "trajopt_prog2.npz")):
        g = np.load(os.path.join(GOLD, name))
        for case in cases:
            prefix, kw, i = case[:3]
            aj = case[3] if len(case) > 3 else False
            pr = af.make_problem(i, **kw)
            ok, prob, traj, _, solver = _solve(pr, analytic=aj)
            prog = pr.get("row_program")
            exact = not (pr.get("quad_n_eq") or (prog is not None and (prog.n_eq or prog.objective)))
            _against_golden(g, prefix, ok, traj, solver, iters=exact)


def test_solver_knobs_tol_and_qp_settings_reach_the_device(gpu):
    """Solver attributes -> sco_sqp_params, solve() keyword arguments -> sco_qp_settings, tol overwrites the three
    thresholds (Q8): the device run follows the flat oracle run with the same numbers."""
    pr = af.make_problem(9, **SMALL)
    attrs = dict(initial_penalty_coeff=10.0, max_merit_coeff_increases=3, initial_trust_region_size=0.5,
                 improve_ratio_threshold=0.2, trust_shrink_ratio=0.2, trust_expand_ratio=1.3, merit_coeff_increase_ratio=5.0)
    ok, prob, traj, _, solver = _solve(pr, attrs=attrs, tol=1e-3, osqp_eps_abs=1e-7, osqp_max_iter=20000, rho=0.2, sigma=1e-9)
    assert solver.min_trust_region_size == solver.min_approx_improve == solver.cnt_tolerance == 1e-3
    ref = sr.penalty_sqp(sr.trajopt_flat(pr), sr.SolverParams(min_trust_region_size=1e-3, min_approx_improve=1e-3,
                                                               cnt_tolerance=1e-3, **attrs),
                         emulate_memo=True, qp_settings=dict(eps_abs=1e-7, max_iter=20000, rho=0.2, sigma=1e-9))
    tr = solver.last_device["traces"][0]
    assert np.array_equal(tr[:, 0], ref.trace[:, 0]) and np.array_equal(tr[:, 6:8], ref.trace[:, 6:8])
    assert ok == ref.success and np.abs(traj.get_value().ravel() - ref.x).max() < TOL


def test_what_is_not_recognised_keeps_the_host_loop(gpu):
    """A callback, a Solver subclass with its own predicate, a plain Expr among the blocks: compile_prob declines and
    Solver.solve runs the reference's Python loop with one device QP per optimize."""
    mods = ct.mirror_mods()
    pr = af.make_problem(2, **SMALL)
    calls = []
    prob, traj, _, _ = tb.build_prob(mods, pr, device_exprs=True)
    prob._callback = lambda: calls.append(1)
    s = mods.Solver()
    ok = s.solve(prob, method="penalty_sqp")
    assert s.last_path == "host" and calls and "callback" in sco_compile.last_reason()

    class Patient(mods.Solver):
        def _y_converged(self, approx_merit_improve):
            return approx_merit_improve < 1e-6
    prob2, traj2, _, _ = tb.build_prob(mods, pr, device_exprs=True)
    s2 = Patient()
    s2.solve(prob2, method="penalty_sqp")
    assert s2.last_path == "host"
    prob3, traj3, _, _ = tb.build_prob(mods, pr, device_exprs=False)
    s3 = mods.Solver()
    ok3 = s3.solve(prob3, method="penalty_sqp")
    assert s3.last_path == "host" and "device expressions" in sco_compile.last_reason()
    ok4, _, traj4, _, _ = _solve(pr)
    assert ok3 == ok4 and np.abs(traj3.get_value() - traj4.get_value()).max() < TOL


def test_solve_many_buckets_by_structure_and_keeps_the_rest_on_the_host(gpu):
    """solve_many([...probs]): Probs of equal structure share ONE device batch each; a Prob with a plain Expr runs the
    host loop beside them.  Results = one Solver().solve per Prob."""
    mods = ct.mirror_mods()
    recs = [af.make_problem(i, **SMALL) for i in range(6)] + [af.make_problem(i, reach=True, **SMALL) for i in range(3)] + \
           [af.make_problem(i, d=2, T=8, K=1, program=True) for i in range(3)]
    built = [tb.build_prob(mods, pr, device_exprs=True) for pr in recs]
    plain = tb.build_prob(mods, af.make_problem(7, **SMALL), device_exprs=False)
    probs = [b[0] for b in built[:4]] + [plain[0]] + [b[0] for b in built[4:]]
    oks, stats = batching.solve_many(probs)
    assert stats["device_problems"] == 12 and stats["device_batches"] == 3 and stats["qps"] > 0
    single = [_solve(pr) for pr in recs]
    k = 0
    for j, p in enumerate(probs):
        if p is plain[0]:
            ref = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(7, **SMALL)), emulate_memo=True)
            assert oks[j] == ref.success and np.abs(plain[1].get_value().ravel() - ref.x).max() < TOL
            continue
        ok1, _, traj1, _, _ = single[k]
        assert oks[j] == ok1 and np.array_equal(built[k][1].get_value(), traj1.get_value())
        assert sum(e.host_evals for e in _device_exprs(p)) == 0
        k += 1


def test_wider_template_goldens_through_plain_solver_solve(gpu):
    """r04: a QuadExpr with per-joint weights and ProgramExpr objects that close over per-timestep parameters are recognised
    by compile_prob; the reference's runs of these cases (trajopt_wide.npz) through plain Solver().solve(prob), no Python f."""
    from wide_cases import CASES
    g = np.load(os.path.join(GOLD, "trajopt_wide.npz"))
    for prefix, kw, i, aj in CASES:
        pr = af.make_problem(i, **kw)
        ok, prob, traj, _, solver = _solve(pr, analytic=aj)
        prog = pr.get("row_program")
        exact = not (kw.get("reach") or kw.get("ee_cost_weight") or (prog is not None and (prog.n_eq or prog.objective)))
        _against_golden(g, prefix, ok, traj, solver, iters=exact)
        from sco_py_amd.sco_osqp import compile as cc
        rec = cc.compile_prob(tb.build_prob(ct.mirror_mods(), pr, analytic_jac=aj, device_exprs=True)[0]).pr
        assert (rec.get("obj_w") is not None) == bool(kw.get("obj_weights")) and (np.ndim(rec.get("row_params")) == 2) == bool(kw.get("per_step"))
