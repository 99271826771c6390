"""Build the SURVEY 8(d) planar-arm trajectory problem through the sco_py object API.

Test infrastructure shared by tests/golden/make_golden.py (which passes the
REFERENCE's modules) and the parity tests (which pass sco_py_amd's mirror): the
construction code is identical, only the module namespace differs.
"""
import numpy as np

from oracle import arm_family as af


def build_prob(mods, pr, analytic_jac=False, device_exprs=False):
    """mods: namespace with Expr, AffExpr, QuadExpr, EqExpr, LEqExpr, BoundExpr,
    Variable, OSQPVar, Prob.  pr: dict from oracle.arm_family.make_problem.
    device_exprs (mirror API only): the non-linear expressions are sco_py_amd.devexpr classes -- the same functions as the
    closures below, as Expr objects the device can evaluate itself, so that plain Solver().solve(prob) runs the resident loop."""
    dx = None
    if device_exprs:
        from sco_py_amd import devexpr as dx
    d, T = pr["d"], pr["T"]
    n_x = d * T
    prob = mods.Prob()
    atoms = np.empty((n_x, 1), dtype=object)
    for t in range(T):
        for j in range(d):
            v = mods.OSQPVar("q%03d_%02d" % (t, j))     # lexicographic == creation order
            prob.add_osqp_var(v)
            atoms[t * d + j, 0] = v
    traj = mods.Variable(atoms, pr["x0"].reshape(n_x, 1).copy())
    prob.add_var(traj)

    Q = af.smooth_Q(d, T, pr.get("obj_w"), pr.get("acc_w"))     # r04: weighted smoothing objective, acceleration term
    prob.add_obj_expr(mods.BoundExpr(mods.QuadExpr(Q, np.zeros((1, n_x)), np.zeros((1, 1))), traj))

    reach = bool(pr.get("reach"))
    n_pin = d if reach else 2 * d
    pins = np.zeros((n_pin, n_x))
    for j in range(d):
        pins[j, j] = 1.0
        if not reach:
            pins[d + j, (T - 1) * d + j] = 1.0
    rhs = (pr["start"] if reach else np.concatenate([pr["start"], pr["goal"]])).reshape(-1, 1)
    prob.add_cnt_expr(mods.BoundExpr(mods.EqExpr(mods.AffExpr(pins, np.zeros((n_pin, 1))), rhs), traj))

    if pr.get("vmax") is not None:
        V = af.velocity_rows(d, T)
        prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(mods.AffExpr(V, np.zeros((V.shape[0], 1))),
                                                      np.full((V.shape[0], 1), pr["vmax"])), traj))

    if pr.get("jlo") is not None:
        eye = np.eye(n_x)
        prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(mods.AffExpr(eye, np.zeros((n_x, 1))),
                                                      np.tile(pr["jhi"], T).reshape(-1, 1)), traj))
        prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(mods.AffExpr(-eye, np.zeros((n_x, 1))),
                                                      -np.tile(pr["jlo"], T).reshape(-1, 1)), traj))

    if pr.get("lin_gen") is not None:
        # r04: general affine rows -- the inequalities as one LEqExpr(AffExpr), the equalities as one EqExpr(AffExpr), on the
        # whole trajectory (prob.py:126-131, 317-346)
        g = pr["lin_gen"]
        for cls, sel in ((mods.LEqExpr, g["is_eq"] == 0), (mods.EqExpr, g["is_eq"] != 0)):
            if np.any(sel):
                prob.add_cnt_expr(mods.BoundExpr(cls(mods.AffExpr(g["A"][sel], np.zeros((int(sel.sum()), 1))), g["rhs"][sel].reshape(-1, 1)), traj))

    R = pr["K"] * pr["O"]
    step_vars = []
    prog = pr.get("row_program")
    span = prog.span if prog is not None else 1
    wide = prog is not None and (span > 1 or prog.n_eq > 0)
    if wide:
        # r03 program blocks: one Variable per block of `span` consecutive timesteps (the reference binds any Expr to any
        # Variable, expr.py:413-437); the block's inequality rows are one LEqExpr, its equality rows one EqExpr with val 0
        # (-> abs penalty, prob.py:280-315), both on that Variable, inequalities first
        for t in range(T - span + 1):
            sv = mods.Variable(atoms[t * d:(t + span) * d, :], pr["x0"][t * d:(t + span) * d].reshape(span * d, 1).copy())
            step_vars.append(sv)
            gids = pr["groups"][t] if pr.get("groups") is not None else None
            for cls, rows in ((mods.LEqExpr, prog.ineq_rows), (mods.EqExpr, prog.eq_rows)):
                if not rows:
                    continue

                def f(x, pr=pr, rows=rows, t=t):
                    return pr["row_program"].evaluate(x.ravel(), af.step_params(pr, t), rows).reshape(-1, 1)

                def grad(x, pr=pr, rows=rows, t=t):
                    return pr["row_program"].jacobian(x.ravel(), af.step_params(pr, t), rows)
                e = mods.Expr(f, grad) if analytic_jac else mods.Expr(f)
                if dx is not None:
                    e = dx.ProgramExpr(prog, af.step_params(pr, t), rows=rows, analytic=analytic_jac)
                prob.add_cnt_expr(mods.BoundExpr(cls(e, np.zeros((len(rows), 1))), sv), gids)
    if pr.get("quad_n_eq"):
        # quadratic rows with equality rows (r03): per timestep one LEqExpr and one EqExpr (val 0) on the same Variable
        ne = int(pr["quad_n_eq"])
        for t in range(T):
            sv = mods.Variable(atoms[t * d:(t + 1) * d, :], pr["x0"][t * d:(t + 1) * d].reshape(d, 1).copy())
            step_vars.append(sv)
            gids = pr["groups"][t] if pr.get("groups") is not None else None
            for cls, sl in ((mods.LEqExpr, slice(0, R - ne)), (mods.EqExpr, slice(R - ne, R))):
                def f(x, pr=pr, sl=sl):
                    return af.quad_rows(x.ravel(), pr["quad_Q"][sl], pr["quad_a"][sl], pr["quad_c"][sl]).reshape(-1, 1)

                def grad(x, pr=pr, sl=sl):
                    return af.quad_rows_jac(x.ravel(), pr["quad_Q"][sl], pr["quad_a"][sl], pr["quad_c"][sl])
                e = mods.Expr(f, grad) if analytic_jac else mods.Expr(f)
                if dx is not None:
                    e = dx.QuadRowsExpr(pr["quad_Q"][sl], pr["quad_a"][sl], pr["quad_c"][sl], analytic=analytic_jac)
                prob.add_cnt_expr(mods.BoundExpr(cls(e, np.zeros((len(range(R)[sl]), 1))), sv), gids)
        wide = True
    for t in range(0 if wide else T):
        sv = mods.Variable(atoms[t * d:(t + 1) * d, :], pr["x0"][t * d:(t + 1) * d].reshape(d, 1).copy())
        step_vars.append(sv)
        if pr.get("circle_rows"):
            # r04: a second kind of non-linear rows on the same Variable, added first: the point's keep-out discs
            nc = pr["circle_rows"]

            def fc_(x, pr=pr, nc=nc):
                return af.point_dist(x.ravel(), pr["obstacles"][:nc]).reshape(-1, 1)

            def gc_(x, pr=pr, nc=nc):
                return af.point_dist_jac(x.ravel(), pr["obstacles"][:nc])
            ec = mods.Expr(fc_, gc_) if analytic_jac else mods.Expr(fc_)
            if dx is not None:
                ec = dx.PointCirclesExpr(pr["obstacles"][:nc], analytic=analytic_jac)
            prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(ec, np.zeros((nc, 1))), sv), pr["groups"][t] if pr.get("groups") is not None else None)

        def f(x, pr=pr, t=t):
            if pr.get("row_program") is not None:    # program family: the compiled rows as a NumPy callable
                return pr["row_program"].evaluate(x.ravel(), af.step_params(pr, t)).reshape(-1, 1)
            if pr.get("quad_Q") is not None:    # quadratic-row family
                return af.quad_rows(x.ravel(), pr["quad_Q"], pr["quad_a"], pr["quad_c"]).reshape(-1, 1)
            if pr.get("point"):                 # point-robot family: distance of the point itself to the discs
                return af.point_dist(x.ravel(), pr["obstacles"]).reshape(-1, 1)
            return af.arm_dist(x.ravel(), pr["link_len"], pr["point_link"], pr["point_frac"],
                               pr["obstacles"]).reshape(-1, 1)

        grad = None
        if analytic_jac:
            def grad(x, pr=pr, t=t):
                if pr.get("row_program") is not None:       # forward-mode derivative of the compiled rows
                    return pr["row_program"].jacobian(x.ravel(), af.step_params(pr, t))
                if pr.get("quad_Q") is not None:
                    return af.quad_rows_jac(x.ravel(), pr["quad_Q"], pr["quad_a"], pr["quad_c"])
                if pr.get("point"):
                    return af.point_dist_jac(x.ravel(), pr["obstacles"])
                return af.arm_dist_jac(x.ravel(), pr["link_len"], pr["point_link"], pr["point_frac"],
                                       pr["obstacles"])
        e = mods.Expr(f, grad) if analytic_jac else mods.Expr(f)
        if dx is not None:
            e = (dx.ProgramExpr(pr["row_program"], af.step_params(pr, t), analytic=analytic_jac) if pr.get("row_program") is not None else
                 dx.QuadRowsExpr(pr["quad_Q"], pr["quad_a"], pr["quad_c"], analytic=analytic_jac) if pr.get("quad_Q") is not None else
                 dx.PointCirclesExpr(pr["obstacles"], analytic=analytic_jac) if pr.get("point") else
                 dx.ArmCirclesExpr(pr["link_len"], pr["point_link"], pr["point_frac"], pr["obstacles"], analytic=analytic_jac))
        gids = pr["groups"][t] if pr.get("groups") is not None else None
        prob.add_cnt_expr(mods.BoundExpr(mods.LEqExpr(e, np.zeros((R - int(pr.get("circle_rows") or 0), 1))), sv), gids)
    if prog is not None and prog.objective:
        # the program family's objective term: one plain Expr per timestep Variable (prob.py:88-104), degree-2 convexified
        for t in range(T):
            def fo(x, pr=pr, t=t):
                return np.array([[pr["row_program"].evaluate(x.ravel(), af.step_params(pr, t), rows=[pr["row_program"].n_rows])[0]]])
            prob.add_obj_expr(mods.BoundExpr(dx.ProgramObjExpr(prog, af.step_params(pr, t)) if dx is not None else mods.Expr(fo), step_vars[t]))
    if pr.get("cost_weight") is not None:
        # non-quadratic objective terms, one Expr per timestep Variable: numeric gradient and Hessian, degree-2
        # convexification with the eigenvalue shift (expr.py:102-156; prob.py:88-104)
        for t in range(T):
            def fc(x, pr=pr):
                return np.array([[af.ee_cost(x.ravel(), pr["link_len"], pr["cost_target"], pr["cost_weight"])]])
            prob.add_obj_expr(mods.BoundExpr(dx.ArmEECostExpr(pr["link_len"], pr["cost_target"], pr["cost_weight"]) if dx is not None
                                             else mods.Expr(fc), step_vars[t]))
    if reach:
        # end-effector target as a non-linear equality on the last timestep (abs penalty, prob.py:280-315)
        def fe(x, pr=pr):
            return af.ee_pos(x.ravel(), pr["link_len"]).reshape(-1, 1)

        def ge(x, pr=pr):
            return af.ee_jac(x.ravel(), pr["link_len"])
        e = mods.Expr(fe, ge) if analytic_jac else mods.Expr(fe)
        if dx is not None:
            e = dx.ArmReachExpr(pr["link_len"], analytic=analytic_jac)
        gids = pr["groups"][T] if pr.get("groups") is not None else None
        prob.add_cnt_expr(mods.BoundExpr(mods.EqExpr(e, pr["target"].reshape(-1, 1)), step_vars[-1]), gids)
    return prob, traj, step_vars, atoms


def canonical_qp(P, q, A, l, u, n_x, with_rows=False):
    """Bring a QP assembled in the reference's (partly arbitrary) order
    into the canonical order used by the oracle and the device path:
      columns  x in name order (already so), then slack columns ordered by the
               first row they appear in (ties: the -1 column before the +1 column);
      rows     constraint rows as given, then the n bound rows in column order.
    P, A dense arrays.  Returns (P, q, A, l, u, column permutation[, row permutation])."""
    n = q.shape[0]
    m_c = A.shape[0] - n
    slack = list(range(n_x, n))

    def key(c):
        rows = np.nonzero(A[:m_c, c])[0]
        first = rows[0] if rows.size else 1 << 30
        sign = A[first, c] if rows.size else 0.0
        return (first, 0 if sign < 0 else 1)

    slack.sort(key=key)
    perm = np.array(list(range(n_x)) + slack, dtype=np.int64)
    A2 = A[:, perm]
    P2 = P[np.ix_(perm, perm)]
    P2 = np.triu(P2) + np.triu(P2.T, 1)       # keep it upper triangular after the permutation
    q2 = q[perm]
    brow = A2[m_c:, :]
    order = np.argsort(np.argmax(brow != 0, axis=1), kind="stable")
    rows = np.concatenate([np.arange(m_c), m_c + order])
    if with_rows:
        return P2, q2, A2[rows, :], l[rows], u[rows], perm, rows
    return P2, q2, A2[rows, :], l[rows], u[rows], perm
