"""Known-answer tests of the mirror API (sco_py_amd.expr / sco_py_amd.sco_osqp.*).

Modelled on the reference's own suite (tests/sco_osqp/test_{expr,variable,prob,solver}.py):
analytic optima checked with np.allclose.  Every test that solves a QP runs twice:
``backend="oracle"`` (CPU, host logic only, oracle ADMM patched in at the seam) and
``backend="hip"`` (marked gpu: the real path through the C ABI).
"""
import numpy as np
import pytest

import conftest as ct

BACKENDS = ["oracle", pytest.param("hip", marks=pytest.mark.gpu)]


@pytest.fixture(params=BACKENDS)
def backend(request):
    if request.param == "oracle":
        request.getfixturevalue("oracle_qp_backend")
    else:
        request.getfixturevalue("gpu")
    return request.param


M = ct.mirror_mods()


def _one_var_prob(value=None, **var_kw):
    prob = M.Prob()
    atom = M.OSQPVar("x", **var_kw)
    prob.add_osqp_var(atom)
    var = M.Variable(np.array([[atom]]), None if value is None else np.array([[value]]))
    prob.add_var(var)
    return prob, var, atom


def _two_var_prob(x0):
    prob = M.Prob()
    a1, a2 = M.OSQPVar("x1"), M.OSQPVar("x2")
    prob.add_osqp_var(a1); prob.add_osqp_var(a2)
    var = M.Variable(np.array([[a1], [a2]]), x0)
    prob.add_var(var)
    return prob, var


# ------------------------------------------------------------------ expr (host only)
def test_numeric_and_analytic_derivatives_agree():
    cases = [(lambda x: x, lambda x: np.array([[1.0]])),
             (lambda x: x ** 2, lambda x: 2 * x),
             (lambda x: x ** 3, lambda x: 3 * x ** 2)]
    for f, df in cases:
        for x0 in (1.0, 2.0, -1.0, 0.0):
            x = np.array([[x0]])
            assert np.allclose(M.Expr(f).grad(x), df(x))
            assert np.allclose(M.Expr(f, df).grad(x, num_check=True), df(x))
    with pytest.raises(Exception) as e:
        M.Expr(cases[0][0]).grad(np.zeros((1, 1, 1)))
    assert "Input shape not supported" in str(e.value)


def test_convexify_degree_one_and_two():
    f = lambda x: x ** 3
    x = np.array([[2.0]])
    aff = M.Expr(f).convexify(x, degree=1)
    assert isinstance(aff, M.AffExpr)
    assert np.allclose(aff.A, 12.0) and np.allclose(aff.b, 8.0 - 24.0) and np.allclose(aff.eval(x), 8.0)
    quad = M.Expr(f).convexify(x, degree=2)
    assert isinstance(quad, M.QuadExpr) and np.allclose(quad.Q, 12.0) and np.allclose(quad.eval(x), 8.0)
    # negative curvature is shifted to zero (expr.py:145-148)
    assert np.allclose(M.Expr(lambda x: -(x ** 2)).convexify(np.zeros((1, 1)), degree=2).Q, 0.0)


def test_eval_is_memoised_on_the_rounded_point():
    calls = []

    def f(x):
        calls.append(1)
        return x * 2.0

    e = M.Expr(f)
    x = np.array([[1.0]])
    first = e.eval(x); second = e.eval(x + 1e-8)       # rounds to the same 6-decimal key (Q3)
    assert len(calls) == 1 and np.allclose(first, second)
    assert e.eval(x + 1e-5) is not None and len(calls) == 2


def test_comparison_expressions():
    aff = M.AffExpr(np.eye(2), np.zeros((2, 1)))
    x = np.array([[1.0], [2.0]])
    assert M.EqExpr(aff, x.copy()).eval(x, tol=0.0)
    assert not M.EqExpr(aff, x + 0.1).eval(x, tol=0.01)
    assert M.LEqExpr(aff, x + 0.1).eval(x) and not M.LEqExpr(aff, x - 0.1).eval(x, tol=0.01)
    val = np.array([[1.0]])
    comp = M.CompExpr(M.Expr(lambda x: x), val)
    val[0, 0] = 5.0
    assert comp.val[0, 0] == 1.0                         # val is copied (expr.py:273)
    for bad in (lambda: comp.eval(0), lambda: comp.convexify(0)):
        with pytest.raises(NotImplementedError):
            bad()
    with pytest.raises(Exception):
        comp.grad(0)
    hinge = M.LEqExpr(M.Expr(lambda x: x ** 2), np.array([[1.0]])).convexify(np.array([[2.0]]))
    assert isinstance(hinge, M.HingeExpr) and np.allclose(hinge.expr.A, 4.0) and np.allclose(hinge.expr.b, -5.0)
    absx = M.EqExpr(M.Expr(lambda x: x ** 2), np.array([[1.0]])).convexify(np.array([[2.0]]))
    assert isinstance(absx, M.AbsExpr) and np.allclose(absx.eval(np.array([[0.0]])), 5.0)


def test_variable_copy_semantics_and_errors():
    atom = M.OSQPVar("x")
    atoms = np.array([[atom]])
    val = np.array([[2.0]])
    var = M.Variable(atoms, val)
    val[0, 0] = 7.0
    assert var.get_value()[0, 0] == 2.0
    got = var.get_value(); got[0, 0] = 9.0
    assert var.get_value()[0, 0] == 2.0
    with pytest.raises(ValueError):
        var.update()                                     # atom has no solver value yet
    var.save(); var._value = np.array([[3.0]]); var.restore()
    assert var.get_value()[0, 0] == 2.0
    var.add_trust_region(0.5)
    assert atom.get_lower_bound() == 1.5 and atom.get_upper_bound() == 2.5
    with pytest.raises(AssertionError):
        atom.set_lower_bound(1)                          # ints are rejected (Q19)


def test_error_messages_of_the_lowering():
    prob = M.Prob()
    with pytest.raises(Exception) as e:
        prob._add_osqp_objs_and_cnts_from_expr(M.BoundExpr(M.CompExpr(M.AffExpr(np.ones((1, 1)), np.zeros((1, 1))),
                                                                     np.zeros((1, 1))), None))
    assert "Comparison" in str(e.value)
    with pytest.raises(Exception) as e:
        prob._add_osqp_objs_and_cnts_from_expr(M.BoundExpr(lambda x: x, None))
    assert "Expression cannot be converted" in str(e.value)
    with pytest.raises(Exception) as e:
        M.Solver().solve(prob, method="nope")
    assert "not supported" in str(e.value)


# ------------------------------------------------------------------ QP seam
def test_optimize_seam_scalar_minimum(backend):
    atom = M.OSQPVar("x")
    var = M.Variable(np.array([[atom]]))
    res, index = M.osqp_utils.optimize(
        [atom], [var], [M.OSQPQuadraticObj(np.array([atom]), np.array([atom]), np.array([2.0]))],
        [M.OSQPLinearObj(atom, -4.0)], [])
    assert res.info.status_val in (1, 2)
    M.osqp_utils.update_osqp_vars(index, res.x); var.update()
    assert np.allclose(var.get_value(), 2.0)


def test_optimize_seam_respects_trust_box(backend):
    atom = M.OSQPVar("x")
    var = M.Variable(np.array([[atom]]))
    var._saved_value = np.array([[4.0]]); var.add_trust_region(1.0)
    res, index = M.osqp_utils.optimize(
        [atom], [var], [M.OSQPQuadraticObj(np.array([atom]), np.array([atom]), np.array([2.0]))],
        [M.OSQPLinearObj(atom, -4.0)], [])
    M.osqp_utils.update_osqp_vars(index, res.x); var.update()
    assert np.allclose(var.get_value(), 3.0)


def test_affine_objective_is_scaled_by_the_penalty_coefficient(backend):
    # Q14: min x^2 - 2x (+ an affine -2x that update_obj(0) wipes out) -> 1.0
    prob, var, _ = _one_var_prob()
    prob.add_obj_expr(M.BoundExpr(M.QuadExpr(2 * np.eye(1), -2 * np.ones((1, 1)), np.zeros((1, 1))), var))
    prob.add_obj_expr(M.BoundExpr(M.AffExpr(-2 * np.ones((1, 1)), np.zeros((1, 1))), var))
    prob.update_obj(penalty_coeff=0)
    assert prob.optimize()
    assert np.allclose(var.get_value(), 1.0)


def test_closest_feasible_point(backend):
    for cnt_val, want in (([1.0, 1.0], [0.0, 0.0]), ([-1.0, 1.0], [-1.0, 0.0]), ([-1.0, -1.0], [-1.0, -1.0])):
        prob, var = _two_var_prob(np.zeros((2, 1)))
        cnt = M.LEqExpr(M.AffExpr(np.eye(2), np.zeros((2, 1))), np.array(cnt_val).reshape(2, 1))
        prob.add_cnt_expr(M.BoundExpr(cnt, var))
        assert prob.find_closest_feasible_point()
        assert np.allclose(var.get_value().ravel(), want)
    prob, var = _two_var_prob(np.zeros((2, 1)))
    target = np.array([[5.0], [-10.0]])
    prob.add_cnt_expr(M.BoundExpr(M.EqExpr(M.AffExpr(np.eye(2), np.zeros((2, 1))), target), var))
    prob.find_closest_feasible_point()
    assert np.allclose(var.get_value(), target)


def test_hinge_and_abs_lowering(backend):
    # min max(0, x + 1) s.t. x == -4 and s.t. x == 1
    for pin in (-4.0, 1.0):
        prob, var, _ = _one_var_prob()
        prob._add_to_lin_objs_and_cnts_from_hinge_expr(M.HingeExpr(M.AffExpr(np.ones((1, 1)), np.ones((1, 1)))), var)
        prob.add_cnt_expr(M.BoundExpr(M.EqExpr(M.AffExpr(np.ones((1, 1)), np.zeros((1, 1))), np.array([[pin]])), var))
        prob.optimize(); var.update()
        assert np.allclose(var.get_value(), pin)
    # min |x + 1| with x <= -4 through the variable's own bound
    prob, var, _ = _one_var_prob(ub=-4.0)
    prob._add_to_lin_objs_and_cnts_from_abs_expr(M.AbsExpr(M.AffExpr(np.ones((1, 1)), np.ones((1, 1)))), var)
    prob.optimize(add_convexified_terms=True); var.update()
    assert np.allclose(var.get_value(), -4.0)


def test_merit_values_of_an_l1_penalised_equality(backend):
    # min x^2 s.t. x == 4: penalty 1 -> x = 0.5, merit 3.75; penalty 2 (compounded: Q1) -> x = 1, merit 7
    prob, var, _ = _one_var_prob()
    prob.add_obj_expr(M.BoundExpr(M.QuadExpr(2 * np.eye(1), np.zeros((1, 1)), np.zeros((1, 1))), var))
    prob.add_cnt_expr(M.BoundExpr(M.EqExpr(M.Expr(lambda x: np.array([[x]]).reshape(1, 1)), np.array([[4.0]])), var))
    prob.optimize()
    prob.convexify(); prob.update_obj(penalty_coeff=1.0); prob.optimize()
    assert np.allclose(var.get_value(), 0.5)
    assert np.allclose(prob.get_value(1.0), 3.75) and np.allclose(prob.get_approx_value(1.0), 3.75)
    prob.update_obj(penalty_coeff=2.0); prob.optimize()
    assert np.allclose(var.get_value(), 1.0)
    assert np.allclose(prob.get_value(2.0), 7.0) and np.allclose(prob.get_approx_value(2.0), 7.0)


def test_model_and_true_merit_differ_for_a_nonlinear_constraint(backend):
    # min x^2 - 2x + 1 s.t. x^2 == 4, convexified at x = 1, penalty 0.5 -> x = 1.5
    prob, var, _ = _one_var_prob(1.0)
    prob.add_obj_expr(M.BoundExpr(M.QuadExpr(2 * np.eye(1), -2 * np.ones((1, 1)), np.ones((1, 1))), var))
    prob.add_cnt_expr(M.BoundExpr(M.EqExpr(M.QuadExpr(2 * np.eye(1), np.zeros((1, 1)), np.zeros((1, 1))),
                                           np.array([[4.0]])), var))
    prob.convexify(); prob.update_obj(penalty_coeff=0.5); prob.optimize()
    assert np.allclose(var.get_value(), 1.5)
    assert np.allclose(prob.get_approx_value(0.5), 1.25) and np.allclose(prob.get_value(0.5), 1.125)


def test_max_constraint_violation():
    prob = M.Prob()
    dummy = M.Variable(np.zeros((1, 1)), np.zeros((1, 1)))
    prob.add_cnt_expr(M.BoundExpr(M.LEqExpr(M.Expr(lambda x: np.array([[1.0, 3.0]])), np.array([[1.0, 1.0]])), dummy))
    prob.add_cnt_expr(M.BoundExpr(M.EqExpr(M.Expr(lambda x: np.array([[0.0, 0.0]])), np.array([[1.0, 1.0]])), dummy))
    assert np.allclose(prob.get_max_cnt_violation(), 2.0)


# ------------------------------------------------------------------ end to end
def _nlp(x0, x_true, f=None, g=None, h=None, Q=None, q=None, A_ineq=None, b_ineq=None):
    """The reference's test_solver harness in our own words: quadratic + black-box
    objective, one linear and one non-linear inequality, one non-linear equality."""
    zero = lambda x: np.array([[0.0]])
    f = f or zero; h = h or zero
    g = g or (lambda x: np.array([[-1e5]]))
    Q = np.zeros((2, 2)) if Q is None else Q
    q = np.zeros((1, 2)) if q is None else q
    A_ineq = np.zeros((1, 2)) if A_ineq is None else A_ineq
    b_ineq = np.zeros((1, 1)) if b_ineq is None else b_ineq
    prob, var = _two_var_prob(x0)
    prob.add_obj_expr(M.BoundExpr(M.QuadExpr(Q, q, np.zeros((1, 1))), var))
    prob.add_obj_expr(M.BoundExpr(M.Expr(f), var))
    prob.add_cnt_expr(M.BoundExpr(M.LEqExpr(M.AffExpr(A_ineq, -b_ineq), np.zeros(b_ineq.shape)), var))
    shape = g(np.zeros((2, 1))).shape
    prob.add_cnt_expr(M.BoundExpr(M.LEqExpr(M.Expr(g), np.zeros(shape)), var))
    prob.add_cnt_expr(M.BoundExpr(M.EqExpr(M.Expr(h), np.zeros(shape)), var))
    solv = M.Solver()
    solv.min_trust_region_size = 1e-5
    solv.max_merit_coeff_increases = 5
    solv.initial_penalty_coeff = 1.0
    solv.solve(prob, method="penalty_sqp")
    assert np.allclose(var.get_value(), x_true, atol=5e-4), (var.get_value().ravel(), np.ravel(x_true))
    return solv


def test_sqp_quadratic_objective_linear_inequality(backend):
    _nlp(np.array([[1.0], [1.0]]), np.array([[1.5], [1.5]]),
         f=lambda x: np.array([[x[0, 0] ** 2 + x[1, 0] ** 2]]), g=lambda x: np.array([[3 - x[0, 0] - x[1, 0]]]))


def test_sqp_rosenbrock_like(backend):
    _nlp(np.array([[-2.0], [1.0]]), np.array([[1.0], [1.0]]),
         f=lambda x: np.array([[(x[1, 0] - x[0, 0] ** 2) ** 2 + (1 - x[0, 0]) ** 2]]),
         g=lambda x: np.array([[-1.5 - x[1, 0]]]))


def test_sqp_nonlinear_equality(backend):
    _nlp(np.array([[10.0], [1.0]]), np.array([[1.0], [1.0]]),
         f=lambda x: np.array([[(1 - x[0, 0]) ** 2]]), h=lambda x: np.array([[10 * (x[1, 0] - x[0, 0] ** 2)]]))


def test_sqp_log_objective_quartic_equality(backend):
    _nlp(np.array([[2.0], [2.0]]), np.array([[0.0], [np.sqrt(3)]]),
         f=lambda x: np.array([[np.log(1 + x[0, 0] ** 2) - x[1, 0]]]),
         h=lambda x: np.array([[(1 + x[0, 0] ** 2) ** 2 + x[1, 0] ** 2 - 4]]))


def test_sqp_hexagon_lp(backend):
    ang = (np.arange(1, 7) * 2 * np.pi / 6).reshape(6, 1)
    A = np.hstack((np.cos(ang), np.sin(ang)))
    q = -np.array([[np.cos(np.pi / 6), np.sin(np.pi / 6)]])
    _nlp(np.zeros((2, 1)), np.array([[1.0], [np.tan(np.pi / 6)]]), q=q, A_ineq=A, b_ineq=np.ones((6, 1)))
    _nlp(np.zeros((2, 1)), np.array([[1.0], [np.tan(np.pi / 6)]]), Q=0.1 * np.eye(2), q=q,
         g=lambda x: 0.01 * (A.dot(x) - np.ones((6, 1))))


def test_sqp_mixed_constraints_and_nonconvex_set(backend):
    _nlp(np.zeros((2, 1)), np.array([[2.0], [1.0]]),
         f=lambda x: np.array([[x[0, 0] ** 4 + x[1, 0] ** 4]]),
         g=lambda x: np.array([[3 - x[0, 0] - x[1, 0]]]), h=lambda x: np.array([[x[0, 0] - 2 * x[1, 0]]]))
    g = lambda x: np.vstack((x[0, 0] ** 2 + x[1, 0] ** 2 - 4,
                             -((x[0, 0] - 1) ** 2 + (x[1, 0] - 1) ** 2 - 0.25),
                             -((x[0, 0] + 1) ** 2 + (x[1, 0] - 1) ** 2 - 0.25),
                             -((x[0, 0]) ** 2 + 7 * (x[1, 0] + 1 - x[0, 0] ** 2 / 2) ** 2 - 0.8)))
    _nlp(np.array([[5.0], [5.0]]), np.zeros((2, 1)), g=g, Q=np.eye(2))


def test_tol_argument_overwrites_thresholds(backend):
    prob, var, _ = _one_var_prob(0.0)
    prob.add_obj_expr(M.BoundExpr(M.QuadExpr(2 * np.eye(1), -2 * np.ones((1, 1)), np.zeros((1, 1))), var))
    s = M.Solver()
    assert s.solve(prob, method="penalty_sqp", tol=1e-3) in (True, False)
    assert (s.min_trust_region_size, s.min_approx_improve, s.cnt_tolerance) == (1e-3, 1e-3, 1e-3)   # Q8
    assert np.allclose(var.get_value(), 1.0, atol=1e-3)


# ------------------------------------------------------------------ concurrent solves, batched QPs
def _small_trajopt_probs(n):
    import trajopt_build as tb
    from oracle import arm_family as af
    return [tb.build_prob(M, af.make_problem(i, d=3, T=6, K=2, O=2)) for i in range(n)]


def test_solve_many_equals_sequential_solves_and_batches_the_qps(backend):
    from sco_py_amd.sco_osqp import batching
    seq = _small_trajopt_probs(5)
    seq_ok = [M.Solver().solve(p[0], method="penalty_sqp") for p in seq]
    par = _small_trajopt_probs(5)
    par_ok, stats = batching.solve_many([p[0] for p in par])
    assert par_ok == seq_ok
    for a, b in zip(seq, par):
        assert np.abs(a[1].get_value() - b[1].get_value()).max() < 1e-9
    # 5 problems x 3 QPs each, same pattern at every round: 3 launches instead of 15
    assert stats["qps"] == 15 and stats["device_launches"] == 3


def test_solve_many_handles_problems_that_finish_at_different_times(backend):
    from sco_py_amd.sco_osqp import batching

    def make(target):
        prob, var, _ = _one_var_prob(0.0)
        prob.add_obj_expr(M.BoundExpr(M.QuadExpr(2 * np.eye(1), -2 * target * np.ones((1, 1)), np.zeros((1, 1))), var))
        if target > 2:          # a non-linear constraint makes this one run more QPs
            prob.add_cnt_expr(M.BoundExpr(M.LEqExpr(M.Expr(lambda x: x ** 2), np.array([[4.0]])), var))
        return prob, var

    seq = [make(t) for t in (1.0, 3.0, 0.5)]
    seq_ok = [M.Solver().solve(p, method="penalty_sqp") for p, _ in seq]
    items = [make(t) for t in (1.0, 3.0, 0.5)]
    oks, stats = batching.solve_many([p for p, _ in items])
    assert oks == seq_ok
    for (_, a), (_, b) in zip(seq, items):
        assert np.abs(a.get_value() - b.get_value()).max() < 1e-9
    assert np.allclose(items[0][1].get_value(), 1.0, atol=1e-4) and np.allclose(items[2][1].get_value(), 0.5, atol=1e-4)
    assert items[1][1].get_value()[0, 0] ** 2 <= 4.0 + 1e-3          # the constrained one stays feasible
    assert stats["device_launches"] < stats["qps"]


def test_repeated_penalty_rows_are_folded_into_weights(backend):
    log = None
    prob, var, _ = _one_var_prob(1.0)
    prob.add_obj_expr(M.BoundExpr(M.QuadExpr(2 * np.eye(1), -2 * np.ones((1, 1)), np.zeros((1, 1))), var))
    prob.add_cnt_expr(M.BoundExpr(M.EqExpr(M.Expr(lambda x: x ** 2), np.array([[4.0]])), var))
    from sco_py_amd.sco_osqp import osqp_utils
    for k in range(1, 4):
        prob.convexify(); prob.update_obj(1.0)
        assert len(prob._osqp_lin_cnt_exprs) == k                  # the reference's list keeps growing (Q2)
        uniq, counts = osqp_utils.fold_repeated_constraints(prob._osqp_lin_cnt_exprs)
        assert len(uniq) == 1 and counts.tolist() == [k]           # ... the device sees one row of weight k
        assert prob.optimize()


@pytest.mark.gpu
def test_device_handles_are_reused_per_sparsity_pattern(gpu):
    """The seam keeps one device handle per sparsity pattern (the SQP loop sends the same pattern for
    every trust-region retry); least recently used handles are closed."""
    from sco_py_amd.sco_osqp import osqp_utils as ou
    ou.clear_handle_cache()

    def solve_scalar(ub):
        v = ou.OSQPVar("x", ub=ub)
        out, _ = ou.optimize([v], [], [ou.OSQPQuadraticObj(np.array([v]), np.array([v]), np.array([2.0]))],
                             [ou.OSQPLinearObj(v, -4.0)], [])
        return out.x[0]

    assert abs(solve_scalar(np.inf) - 2.0) < 1e-5
    assert abs(solve_scalar(1.5) - 1.5) < 1e-5         # same pattern, other bounds: the handle is reused
    assert len(ou._HANDLE_CACHE) == 1
    first = next(iter(ou._HANDLE_CACHE.values()))
    for k in range(2, 2 + ou._HANDLE_CACHE_MAX):        # other patterns (k variables) push it out
        vs = [ou.OSQPVar("x%02d" % i) for i in range(k)]
        ou.optimize(vs, [], [ou.OSQPQuadraticObj(np.array(vs), np.array(vs), np.full(k, 2.0))],
                    [ou.OSQPLinearObj(v, -4.0) for v in vs], [])
    assert len(ou._HANDLE_CACHE) == ou._HANDLE_CACHE_MAX and first not in ou._HANDLE_CACHE.values()
    ou.clear_handle_cache()
    assert len(ou._HANDLE_CACHE) == 0


@pytest.mark.gpu
def test_warm_start_extension_of_the_seam(gpu):
    """osqp_utils.WARM_START (off by default): same optimum, never more ADMM iterations on a repeated
    pattern (how many fewer depends on how far the multipliers move between two QPs)."""
    import trajopt_build as tb
    from oracle import arm_family as af
    from sco_py_amd.sco_osqp import osqp_utils as ou
    real, log = ou._solve_qp_batch, []

    def logged(reqs):
        out = real(reqs); log.append(out[0][2]); return out

    results = {}
    try:
        ou._solve_qp_batch = logged
        for warm in (False, True):
            ou.clear_handle_cache(); del log[:]
            ou.WARM_START = warm
            pr = af.make_problem(3, d=3, T=6, K=2, O=2)
            prob, traj, _, _ = tb.build_prob(M, pr)
            s = M.Solver(); s.initial_penalty_coeff = 10.0; s.max_merit_coeff_increases = 3
            ok = s.solve(prob, method="penalty_sqp")
            results[warm] = (ok, traj.get_value().ravel().copy(), list(log))
    finally:
        ou._solve_qp_batch = real; ou.WARM_START = False; ou.clear_handle_cache()
    assert results[False][0] == results[True][0]
    assert np.abs(results[False][1] - results[True][1]).max() < 1e-3
    assert sum(results[True][2]) <= sum(results[False][2])


def test_adaptive_rho_is_passed_down_the_seam(backend):
    """Solver.solve(..., adaptive_rho=True) (solver.py:39): the flag reaches the QP solve; same optimum."""
    prob, var, atom = _one_var_prob(0.0)
    prob.add_obj_expr(M.BoundExpr(M.QuadExpr(np.array([[2.0]]), np.array([[-4.0]]), np.zeros((1, 1))), var))
    seen = []
    from sco_py_amd.sco_osqp import osqp_utils
    real = osqp_utils._solve_qp_batch
    osqp_utils._solve_qp_batch = lambda reqs: (seen.extend(r["settings"] for r in reqs), real(reqs))[1]
    try:
        assert M.Solver().solve(prob, method="penalty_sqp", adaptive_rho=True)
    finally:
        osqp_utils._solve_qp_batch = real
    assert np.allclose(var.get_value(), 2.0, atol=1e-4)
    # the projection QP runs with the defaults, the penalty QPs with the caller's settings
    assert [s[5] for s in seen][0] == 0.0 and all(s[5] == 1.0 for s in seen[1:]) and len(seen) > 1


def test_classify_trial_takes_the_exits_in_the_reference_order():
    """solver.py:181-251: bad model -> y-converged -> stalled group -> shrink / accept, on crafted numbers."""
    from sco_py_amd.sco_osqp import solver as sv
    thr = sv.Thresholds(0.25, 1e-8, 1e-4)
    none = ({}, {}, [])
    e = np.zeros(0)

    def code(merit, model, new, mv=e, av=e, groups=none):
        return sv.classify_trial(sv.Trial(merit, model, new, mv, av), thr, *groups)

    assert code(1.0, 1.1, 0.0).code == sv.STEP_BAD                       # model worse than -1e-5 wins over everything
    assert code(1.0, 1.0 - 1e-9, 5.0).code == sv.STEP_YCONV              # under the y threshold, even though new is worse
    v = code(1.0, 1.0, 1.0)                                              # exactly zero improvement is nudged to 1e-12
    assert v.code == sv.STEP_YCONV and v.approx_improve == 1e-12
    assert code(1.0, 0.5, 0.95).code == sv.STEP_SHRINK                   # ratio 0.1 < 0.25
    assert code(1.0, 0.5, 1.2).code == sv.STEP_SHRINK                    # exact improvement negative
    assert code(1.0, 0.5, 0.6).code == sv.STEP_ACCEPT
    # groups "a", "b" overlap, "c" stands alone; vector order is the sorted ids
    gi = {"a": 0, "b": 1, "c": 2}
    ov = {"a": {"b"}, "b": {"a"}, "c": set()}
    groups = (gi, ov, ["a", "b", "c"])
    mv = np.array([1.0, 1.0, 1.0])
    # "a" predicts nothing but its neighbour "b" progresses: no stall; the accept test decides
    v = code(3.0, 2.0, 2.1, mv, np.array([1.0, 0.5, 0.5]), groups)
    assert v.code == sv.STEP_ACCEPT and v.reported == []
    # "c" violated, no progress, no overlap: stall; the report lists it, then every violated group under the threshold
    v = code(3.0, 2.0, 2.1, mv, np.array([1.0, 0.5, 1.0]), groups)
    assert v.code == sv.STEP_GROUP and v.stalled == ["c"] and v.reported == ["c", "a", "c"]
    # a satisfied group (merit below cnt_tolerance) cannot stall
    v = code(3.0, 2.0, 2.1, np.array([1.0, 1.0, 1e-5]), np.array([0.5, 0.5, 1e-5]), groups)
    assert v.code == sv.STEP_ACCEPT


def test_overridden_predicates_of_a_solver_subclass_decide_as_in_the_reference():
    """The reference's Solver decides through its overridable predicate methods (_bad_model, _y_converged,
    _shrink_trust_region: /root/reference/sco_py/sco_osqp/solver.py:255-283); the mirror routes classify_trial through the
    same methods, so a subclass that overrides one changes the decisions here as it would there."""
    from sco_py_amd.sco_osqp import solver as sv
    thr = sv.Thresholds(0.25, 1e-8, 1e-4)
    trial = sv.Trial(10.0, 9.0, 9.9, [], [])          # model improvement 1.0, exact 0.1: ratio 0.1 -> shrink
    assert sv.classify_trial(trial, thr, {}, {}, []).code == sv.STEP_SHRINK
    s = sv.Solver()
    preds = (s._bad_model, s._y_converged, s._shrink_trust_region)
    assert sv.classify_trial(trial, thr, {}, {}, [], predicates=preds).code == sv.STEP_SHRINK

    class Lenient(sv.Solver):
        def _shrink_trust_region(self, exact_merit_improve, merit_improve_ratio):
            return exact_merit_improve < 0            # any improvement is accepted

    class Jumpy(sv.Solver):
        def _y_converged(self, approx_merit_improve):
            return approx_merit_improve < 5.0

    l = Lenient(); j = Jumpy()
    assert sv.classify_trial(trial, thr, {}, {}, [], predicates=(l._bad_model, l._y_converged, l._shrink_trust_region)).code == sv.STEP_ACCEPT
    assert sv.classify_trial(trial, thr, {}, {}, [], predicates=(j._bad_model, j._y_converged, j._shrink_trust_region)).code == sv.STEP_YCONV
