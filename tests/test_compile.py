"""Host side of the object-API -> device-loop compile step (sco_py_amd/sco_osqp/compile.py, sco_py_amd/devexpr.py).

No GPU here: what compile_prob reads out of a Prob, what it declines and why, that device expressions ARE reference-style
Expr objects (the host loop on them reproduces the golden runs of the reference's modules), and that without a GPU a
recognised Prob fails loudly instead of falling back."""
import os

import numpy as np
import pytest

import conftest as ct
import trajopt_build as tb
from oracle import arm_family as af
from sco_py_amd import _lib, devexpr as dx
from sco_py_amd.sco_osqp import batching, compile as cc

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = dict(d=3, T=6, K=2, O=2)

FAMILIES = [
    dict(SMALL), dict(SMALL, reach=True), dict(SMALL, vel_limit=0.6, joint_limit=0.3, groups="halves"),
    dict(SMALL, ee_cost_weight=0.5), dict(SMALL, reach=True, groups="split"), dict(d=2, T=8, O=3, point=True),
    dict(d=2, T=8, O=3, quadratic=True), dict(d=2, T=8, O=3, quadratic=True, n_eq=1), dict(d=2, T=8, program=True),
    dict(d=3, T=6, K=1, program=True, variant="dynamics"), dict(d=2, T=6, K=1, program=True, variant="sweep", groups="split"),
    dict(d=2, T=6, K=1, program=True, variant="attract"), dict(d=3, T=6, K=1, program=True, variant="curve"), dict(),
    # r04, the wider template: weighted smoothing objective; program parameters per timestep
    dict(SMALL, obj_weights=True), dict(SMALL, obj_weights=True, reach=True, vel_limit=0.6), dict(d=2, T=8, program=True, per_step=True),
    dict(d=2, T=6, K=1, program=True, variant="sweep", per_step=True), dict(d=2, T=6, K=1, program=True, variant="attract", per_step=True, obj_weights=True),
    dict(d=2, T=8, K=1, program=True, variant="accel"), dict(d=2, T=9, K=1, program=True, variant="jerk", per_step=True),      # span 3, span 4
    dict(SMALL, lin_rows=True), dict(SMALL, lin_rows=True, vel_limit=0.6, joint_limit=0.3, obj_weights=True), dict(d=2, T=8, K=1, O=3, point=True, lin_rows=True),
    dict(d=2, T=8, K=1, program=True, circles=2), dict(d=2, T=8, K=1, program=True, variant="attract", circles=1, per_step=True, groups="split"),   # two kinds of rows
    dict(SMALL, acc_weights=True), dict(SMALL, acc_weights=True, obj_weights=True, reach=True), dict(d=2, T=8, K=1, program=True, acc_weights=True),       # acceleration term
]


@pytest.mark.parametrize("kw", FAMILIES, ids=lambda kw: "-".join("%s=%s" % kv for kv in kw.items()) or "7x20")
def test_compile_prob_reads_the_problem_record_back_out_of_the_object_api(kw):
    """build_prob(record) -> Prob -> compile_prob -> the record the array API takes: every field identical."""
    pr = af.make_problem(1, **kw)
    prob, traj, _, _ = tb.build_prob(ct.mirror_mods(), pr, device_exprs=True)
    cp = cc.compile_prob(prob)
    assert cp is not None, cc.last_reason()
    for k, v in pr.items():
        if k == "goal" and pr.get("reach"):
            continue                                   # the reach variant has no goal pin
        if k == "row_program":
            assert cp.pr[k] is v
        elif k == "groups":
            assert cp.pr[k] == v
        elif isinstance(v, bool) or v is None:
            assert bool(cp.pr.get(k)) == bool(v), k
        elif k == "obj_w" and pr.get("acc_w") is not None:      # read back through -2 w - 4 a: one rounding
            assert np.allclose(cp.pr[k], v, rtol=4e-16, atol=0)
        elif k == "lin_gen":
            assert all(np.array_equal(cp.pr[k][f], v[f]) for f in ("A", "rhs", "is_eq"))
        elif k == "row_params" and np.ndim(v) == 2:
            nb = pr["T"] - pr["row_program"].span + 1          # rows of the blocks; the rest belongs to objective terms (or to nobody)
            assert np.shape(cp.pr[k]) == np.shape(v) and np.array_equal(np.asarray(cp.pr[k])[:nb], np.asarray(v)[:nb]), k
        else:
            assert np.array_equal(np.asarray(cp.pr[k]), np.asarray(v)), k
    assert cp.key[6] == 2                              # prox_count: the trajectory Variable + one block Variable per atom
    assert len(cp.holders) == len(prob._vars) and [a.var_name for a in cp.atoms] == sorted(a.var_name for a in prob._osqp_vars)


def test_problems_of_equal_structure_share_a_key_and_others_do_not():
    mods = ct.mirror_mods()
    key = lambda i, **kw: cc.compile_prob(tb.build_prob(mods, af.make_problem(i, **kw), device_exprs=True)[0]).key
    assert key(0, **SMALL) == key(5, **SMALL)
    others = [key(0, **dict(SMALL, T=7)), key(0, **dict(SMALL, reach=True)), key(0, **dict(SMALL, vel_limit=0.5)),
              key(0, **dict(SMALL, groups="halves")), key(0, **dict(SMALL, K=3)), key(0, d=2, T=8, program=True)]
    assert len(set(others + [key(0, **SMALL)])) == 7
    aj = cc.compile_prob(tb.build_prob(mods, af.make_problem(0, **SMALL), analytic_jac=True, device_exprs=True)[0]).key
    assert aj != key(0, **SMALL)


def test_what_compile_prob_declines_and_why():
    mods = ct.mirror_mods()
    pr = af.make_problem(0, **SMALL)

    def fresh(**kw):
        return tb.build_prob(mods, pr, device_exprs=True, **kw)

    def declined(prob, word):
        assert cc.compile_prob(prob) is None and word in cc.last_reason(), cc.last_reason()

    prob, traj, step_vars, atoms = fresh()
    assert cc.compile_prob(prob) is not None and cc.last_reason() == ""
    declined(tb.build_prob(mods, pr)[0], "device expressions")                       # plain Expr(f)
    prob, *_ = fresh(); prob._callback = lambda: None
    declined(prob, "callback")
    prob, traj, *_ = fresh()
    prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(mods.AffExpr(np.ones((1, 18)), np.zeros((1, 1))), np.ones((1, 1))), traj))
    cp = cc.compile_prob(prob)                                                        # a general affine row: in the template since r04
    assert cp is not None and cp.pr["lin_gen"]["A"].shape == (1, 18) and cp.pr["lin_gen"]["is_eq"].tolist() == [0] and cp.key[-1] is not None
    prob, traj, *_ = fresh()
    extra = mods.OSQPVar("zz_extra"); prob.add_osqp_var(extra)
    prob._osqp_lin_cnt_exprs.append(mods.OSQPLinearConstraint(np.array([extra]), np.array([1.0]), -1.0, 1.0))
    assert cc.compile_prob(prob) is None                                              # (an atom outside the trajectory)
    prob, traj, *_ = fresh()
    prob.add_obj_expr(mods.BoundExpr(mods.QuadExpr(np.eye(18), np.zeros((1, 18)), np.zeros((1, 1))), traj))
    declined(prob, "one quadratic objective")
    prob, traj, step_vars, atoms = fresh()
    prob._quad_obj_exprs[0].expr.Q[0, 0] += 1.0
    declined(prob, "sum_t")
    prob, traj, step_vars, atoms = fresh()
    atoms[3, 0].set_lower_bound(-1.0)
    declined(prob, "own bounds")
    prob, traj, step_vars, atoms = fresh()
    step_vars[2]._value[0, 0] += 0.5
    declined(prob, "disagree")
    prob, traj, step_vars, atoms = fresh()
    prob.add_var(mods.Variable(atoms[:3, :], pr["x0"][:3].reshape(3, 1)))           # one more holder of timestep 0 only
    declined(prob, "different numbers of Variables")
    prob, traj, step_vars, atoms = fresh()
    e = prob._nonlin_cnt_exprs[2].expr.expr
    e.obstacles = e.obstacles + 0.01
    declined(prob, "per-timestep")
    prob, *_ = fresh()
    prob._nonlin_cnt_exprs.reverse()
    declined(prob, "start at timestep 0")
    prob, *_ = fresh()
    prob._nonlin_cnt_exprs[0].expr.val[0, 0] = 0.1
    declined(prob, "val = 0")
    prob, traj, step_vars, atoms = fresh()
    extra = dx.ArmCirclesExpr(pr["link_len"], pr["point_link"], pr["point_frac"], pr["obstacles"])
    prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(extra, np.zeros((4, 1))), step_vars[1]))
    declined(prob, "timestep order")
    prob, traj, step_vars, atoms = fresh()
    prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(extra, np.zeros((4, 1))), step_vars[5]))
    declined(prob, "exactly one LEqExpr")
    prob, *_ = fresh()
    prob.convexify(); prob.update_obj(1.0)
    declined(prob, "already been lowered")


def test_device_expressions_are_reference_style_exprs(oracle_qp_backend):
    """The host loop on device expressions = the golden runs of the reference's modules on the equivalent closures
    (expr.py:22-156 semantics: eval memo, numeric / analytic Jacobian, numeric Hessian, convexify)."""
    mods = ct.mirror_mods()
    for name, prefix, kw, i, aj in (("trajopt_small.npz", "p1_", SMALL, 1, False), ("trajopt_reach.npz", "p1a_", dict(SMALL, reach=True), 1, True),
                                    ("trajopt_obj.npz", "o0_", dict(SMALL, ee_cost_weight=0.5), 0, False),
                                    ("trajopt_prog2.npz", "ja_dy_", dict(K=1, program=True, d=3, T=8, variant="dynamics"), 0, True),
                                    ("trajopt_quad2.npz", "qe0_", dict(d=2, T=8, K=1, O=3, quadratic=True, n_eq=1), 0, False)):
        g = np.load(os.path.join(GOLD, name))
        del oracle_qp_backend[:]
        prob, traj, _, _ = tb.build_prob(mods, af.make_problem(i, **kw), analytic_jac=aj, device_exprs=True)
        s = mods.Solver(); s.device_loop = False
        ok = s.solve(prob, method="penalty_sqp")
        assert s.last_path == "host"
        assert ok == bool(g[prefix + "success"]) and np.abs(traj.get_value().ravel() - g[prefix + "x"]).max() < 2e-7, prefix
        assert [r["status"] for r in oracle_qp_backend] == [int(g["%sqp%d_status" % (prefix, k)]) for k in range(int(g[prefix + "n_qp"]))]
        assert sum(be.expr.expr.host_evals for be in prob._nonlin_cnt_exprs) > 0


def test_a_recognised_prob_without_a_gpu_fails_loudly():
    """No CPU path in the product: Solver.solve on a compiled Prob needs the device."""
    if _lib.device_count() > 0:
        pytest.skip("a GPU is visible: tests/test_object_api_gpu.py runs this path for real")
    mods = ct.mirror_mods()
    prob, _, _, _ = tb.build_prob(mods, af.make_problem(0, **SMALL), device_exprs=True)
    with pytest.raises(_lib.ScoHipError):
        mods.Solver().solve(prob, method="penalty_sqp")
    with pytest.raises(_lib.ScoHipError):
        batching.solve_many([prob])


def test_overridden_y_converged_decides_the_group_report(oracle_qp_backend):
    """solver.py:233 builds the reported list with self._y_converged; the stall test (:213-221) uses the raw threshold.
    A subclass with a looser _y_converged reports more groups than the base class on the same trial."""
    from sco_py_amd.sco_osqp import solver as sv
    trial = sv.Trial(10.0, 9.0, 9.5, np.array([1.0, 1.0, 1.0]), np.array([1.0, 1.0 - 5e-7, 0.0]))
    thr = sv.Thresholds(0.25, 1e-8, 1e-4)
    gi = {"a": 0, "b": 1, "c": 2}
    ov = {"a": set(), "b": set(), "c": set()}
    base = sv.classify_trial(trial, thr, gi, ov, ["a", "b", "c"])
    assert base.code == sv.STEP_GROUP and base.stalled == ["a"] and base.reported == ["a", "a"]
    loose = sv.classify_trial(trial, thr, gi, ov, ["a", "b", "c"],
                              predicates=(lambda a: a < -1e-5, lambda a: a < 1e-6, lambda e, r: e < 0 or r < 0.25))
    assert loose.stalled == ["a"] and loose.reported == ["a", "a", "b"]
