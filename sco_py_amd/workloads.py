"""Synthetic trajectory-optimisation workload (inputs only; nothing here solves anything).

The reference ships no benchmark problems; this module defines the
seeded planar-arm workload of SURVEY.md 8(d) in NumPy so that bench.py, the tests,
the oracle and the golden-vector generators all hand the device kernels the same
inputs.  The NumPy forward kinematics below are the definition of the device
constraint families (SCO_FAM_ARM_CIRCLES / SCO_FAM_ARM_REACH / SCO_FAM_POINT_CIRCLES, csrc/sco_sqp.hip) and
what a host-side `Expr(f)` of the same constraint evaluates.

Problem i (``make_problem(i, ...)``):
  variables   joint angles theta[t][j], t < T, j < d, flattened time-major
  objective   sum_t || theta[t+1] - theta[t] ||^2            (QuadExpr)
  linear      theta[0] = start, theta[T-1] = goal            (EqExpr(AffExpr))
              (reach variant: only the start pin; ee(theta[T-1]) = target as EqExpr(Expr))
  nonlinear   per timestep, K link points x O circular obstacles (LEqExpr, val = 0):
                g[k*O + o](theta_t) = r_o - || p_k(theta_t) - c_o ||
              p_k = planar forward kinematics of a serial arm (cumulative angles)
"""
import numpy as np


def link_points(theta, link_len, point_link, point_frac):
    """Positions (K, 2) of the K link points for joint angles theta (d,)."""
    phi = np.cumsum(theta)
    cx, cy = np.cos(phi), np.sin(phi)
    jx = np.concatenate(([0.0], np.cumsum(link_len * cx)))   # joint positions
    jy = np.concatenate(([0.0], np.cumsum(link_len * cy)))
    pl = np.asarray(point_link)
    px = jx[pl] + point_frac * link_len[pl] * cx[pl]
    py = jy[pl] + point_frac * link_len[pl] * cy[pl]
    return np.stack([px, py], axis=1)


def arm_dist(theta, link_len, point_link, point_frac, obstacles):
    """g (K*O,) with g[k*O + o] = r_o - ||p_k - c_o||."""
    p = link_points(np.asarray(theta, dtype=np.float64).ravel(), link_len, point_link, point_frac)
    diff = p[:, None, :] - obstacles[None, :, :2]
    dist = np.sqrt(diff[..., 0] ** 2 + diff[..., 1] ** 2)
    return (obstacles[None, :, 2] - dist).ravel()


def arm_dist_jac(theta, link_len, point_link, point_frac, obstacles):
    """Analytic Jacobian (K*O, d) of arm_dist."""
    theta = np.asarray(theta, dtype=np.float64).ravel()
    d = theta.shape[0]
    phi = np.cumsum(theta)
    cx, cy = np.cos(phi), np.sin(phi)
    p = link_points(theta, link_len, point_link, point_frac)
    K, O = p.shape[0], obstacles.shape[0]
    # d p_k / d theta_j = sum over links i >= j (up to the point's link) of the
    # rotated link vector
    dp = np.zeros((K, d, 2))
    for k in range(K):
        lk = point_link[k]
        for j in range(lk + 1):
            sx = sy = 0.0
            for i in range(j, lk + 1):
                L = link_len[i] * (point_frac[k] if i == lk else 1.0)
                sx += -L * cy[i]
                sy += L * cx[i]
            dp[k, j, 0], dp[k, j, 1] = sx, sy
    J = np.zeros((K * O, d))
    for k in range(K):
        for o in range(O):
            dx = p[k, 0] - obstacles[o, 0]
            dy = p[k, 1] - obstacles[o, 1]
            dist = np.sqrt(dx * dx + dy * dy)
            J[k * O + o, :] = -(dx * dp[k, :, 0] + dy * dp[k, :, 1]) / dist
    return J


def ee_pos(theta, link_len):
    """End-effector position (2,) of the arm."""
    phi = np.cumsum(np.asarray(theta, dtype=np.float64).ravel())
    return np.array([np.sum(link_len * np.cos(phi)), np.sum(link_len * np.sin(phi))])


def ee_jac(theta, link_len):
    """Analytic Jacobian (2, d) of ee_pos."""
    phi = np.cumsum(np.asarray(theta, dtype=np.float64).ravel())
    d = phi.shape[0]
    J = np.zeros((2, d))
    for j in range(d):
        J[0, j] = -np.sum(link_len[j:] * np.sin(phi[j:]))
        J[1, j] = np.sum(link_len[j:] * np.cos(phi[j:]))
    return J


def point_dist(x, obstacles):
    """SCO_FAM_POINT_CIRCLES: g (O,) with g[o] = r_o - ||x[:2] - c_o|| for a point robot at x[:2] (further coordinates of the
    state, if any, are unconstrained)."""
    x = np.asarray(x, dtype=np.float64).ravel()
    dx, dy = x[0] - obstacles[:, 0], x[1] - obstacles[:, 1]
    return obstacles[:, 2] - np.sqrt(dx * dx + dy * dy)


def point_dist_jac(x, obstacles):
    """Analytic Jacobian (O, d) of point_dist."""
    x = np.asarray(x, dtype=np.float64).ravel()
    dx, dy = x[0] - obstacles[:, 0], x[1] - obstacles[:, 1]
    dist = np.sqrt(dx * dx + dy * dy)
    J = np.zeros((obstacles.shape[0], x.shape[0]))
    J[:, 0] = -dx / dist; J[:, 1] = -dy / dist
    return J


def quad_rows(x, Q, a, c):
    """SCO_FAM_STATE_QUADRATIC: g (R,) with g[r] = 1/2 x' Q_r x + a_r' x + c_r."""
    x = np.asarray(x, dtype=np.float64).ravel()
    return 0.5 * np.einsum("i,rij,j->r", x, Q, x) + a @ x + c


def quad_rows_jac(x, Q, a, c):
    """Analytic Jacobian (R, d) of quad_rows (Q_r symmetric)."""
    x = np.asarray(x, dtype=np.float64).ravel()
    return a + np.einsum("rij,j->ri", Q, x)


def velocity_rows(d, T):
    """V (2 d (T-1), d T): rows theta[t+1][j] - theta[t][j] (t-major), then their negatives."""
    D = np.zeros((d * (T - 1), d * T))
    for t in range(T - 1):
        for j in range(d):
            D[t * d + j, t * d + j] = -1.0
            D[t * d + j, (t + 1) * d + j] = 1.0
    return np.vstack([D, -D])


def joint_limit_rows(d, T):
    """(2 d T, d T): +I (theta <= hi), then -I (-theta <= -lo)."""
    eye = np.eye(d * T)
    return np.vstack([eye, -eye])


def default_points(d, K):
    """K link points spread over the links: point k sits at the END of link
    (k * d) // K ... evenly, fraction 1.0 for the last point of a link."""
    link = np.array([min(d - 1, ((k + 1) * d - 1) // K) for k in range(K)], dtype=np.int32)
    frac = np.ones(K)
    # several points on the same link are spread along it
    for l in range(d):
        idx = np.where(link == l)[0]
        for r, k in enumerate(idx):
            frac[k] = (r + 1) / len(idx)
    return link, frac


def block_groups(T, reach, scheme):
    """Constraint-group ids per non-linear block (T timestep blocks, then the reach block):
    scheme "halves" puts every block in "all" and the first / second half of the horizon in
    "head" / "tail" (so "all" overlaps both), the reach block in "all" and "reach"; scheme
    "split" uses the disjoint groups "head", "tail" (and "reach") only."""
    if scheme is None:
        return None
    assert scheme in ("halves", "split")
    if scheme == "split":            # disjoint groups, no "all": one stalled half ends the minimisation
        return [["head"] if t < T // 2 else ["tail"] for t in range(T)] + ([["reach"]] if reach else [])
    g = [["all", "head"] if t < T // 2 else ["all", "tail"] for t in range(T)]
    if reach:
        g.append(["all", "reach"])
    return g


def ee_cost(theta, link_len, target, weight):
    """Non-quadratic objective term of one timestep: weight * || ee(theta) - target ||^2 (task-space attraction).
    Its Hessian in theta is indefinite away from the target, so the reference's degree-2 convexification
    (expr.py:143-153) has to shift its eigenvalues."""
    e = ee_pos(theta, link_len) - np.asarray(target, dtype=np.float64)
    return float(weight) * float(e[0] * e[0] + e[1] * e[1])


def make_point_problem(i, d=2, T=20, O=3, noise=0.03, groups=None, vel_limit=None, joint_limit=None):
    """Seeded problem i of the point-robot family (SCO_FAM_POINT_CIRCLES): a point in the plane (state dimension d >= 2, the
    first two coordinates are its position) goes from start to goal past O discs that sit on its straight path.  Same
    dictionary layout as make_problem (K = 1: the point itself; link data are placeholders the family does not read)."""
    rng = np.random.default_rng(5000 + i)
    start = rng.uniform(-1.0, 1.0, size=d)
    goal = -start + 0.3 * rng.standard_normal(d)                 # roughly across the origin
    s = np.linspace(0.0, 1.0, T)[:, None]
    x0 = (1 - s) * start[None, :] + s * goal[None, :] + noise * rng.standard_normal((T, d))
    along = np.sort(rng.uniform(0.2, 0.8, size=O))               # discs near the straight path, off-centre
    centre = (1 - along[:, None]) * start[None, :2] + along[:, None] * goal[None, :2] + 0.08 * rng.standard_normal((O, 2))
    radius = rng.uniform(0.08, 0.2, size=O)
    obstacles = np.concatenate([centre, radius[:, None]], axis=1)
    out = dict(d=d, T=T, K=1, O=O, x0=x0.ravel(), start=start, goal=goal, link_len=np.ones(d),
               point_link=np.zeros(1, dtype=np.int32), point_frac=np.ones(1), obstacles=obstacles, reach=False, point=True)
    if groups is not None:
        out["groups"] = block_groups(T, False, groups)
    if vel_limit is not None:
        out["vmax"] = float(vel_limit)
    if joint_limit is not None:                                   # workspace box around the straight path
        out["jlo"] = np.minimum(start, goal) - float(joint_limit)
        out["jhi"] = np.maximum(start, goal) + float(joint_limit)
    return out


def make_quadratic_problem(i, d=2, T=20, O=3, noise=0.03, groups=None, vel_limit=None, joint_limit=None, n_eq=0):
    """Seeded problem i of the quadratic-row family (SCO_FAM_STATE_QUADRATIC): a state x_t in R^d moves from start to goal;
    per timestep O rows 1/2 x' Q_r x + a_r' x + c_r <= 0: keep-OUT ellipsoids on the straight path (concave rows,
    1 - (x - m)' M (x - m) <= 0), for O >= 3 the last row a keep-IN ball around the path's midpoint (convex) and for O >= 4
    the row before it a half-space.  Same dictionary layout as make_problem (K = 1; link data and obstacles are placeholders).
    ``n_eq`` = 1 (r03) appends an EQUALITY row: the state stays on a sphere through start and goal, |x - m|^2 = rad^2 (an
    EqExpr on a quadratic Expr -> abs penalty, prob.py:280-315); the dictionary then has O + 1 rows and quad_n_eq = 1."""
    rng = np.random.default_rng(7000 + i)
    start = rng.uniform(-1.0, 1.0, size=d)
    goal = -start + 0.3 * rng.standard_normal(d)
    s = np.linspace(0.0, 1.0, T)[:, None]
    x0 = (1 - s) * start[None, :] + s * goal[None, :] + noise * rng.standard_normal((T, d))
    Q = np.zeros((O, d, d)); a = np.zeros((O, d)); c = np.zeros(O)
    n_out = O - (1 if O >= 3 else 0) - (1 if O >= 4 else 0)
    along = np.sort(rng.uniform(0.25, 0.75, size=n_out))
    for r in range(n_out):
        m = (1 - along[r]) * start + along[r] * goal + 0.06 * rng.standard_normal(d)
        axes = rng.uniform(0.1, 0.25, size=d)                         # semi-axes of the keep-out ellipsoid
        R_, _ = np.linalg.qr(rng.standard_normal((d, d)))
        M = R_ @ np.diag(1.0 / axes ** 2) @ R_.T
        M = 0.5 * (M + M.T)
        Q[r] = -2.0 * M; a[r] = 2.0 * M @ m; c[r] = 1.0 - m @ M @ m    # 1 - (x - m)' M (x - m)
    if O >= 4:                                                          # half-space n' x <= n' mid + 0.6
        nrm = rng.standard_normal(d); nrm /= np.linalg.norm(nrm)
        a[O - 2] = nrm; c[O - 2] = -(nrm @ (0.5 * (start + goal)) + 0.6)
    if O >= 3:                                                          # keep-in ball |x - mid|^2 <= rad^2
        mid = 0.5 * (start + goal); rad = 0.75 * np.linalg.norm(goal - start) + 0.4
        Q[O - 1] = 2.0 * np.eye(d); a[O - 1] = -2.0 * mid; c[O - 1] = mid @ mid - rad ** 2
    if n_eq:
        assert n_eq == 1 and d >= 2
        chord = goal - start
        perp = rng.standard_normal(d); perp -= (perp @ chord) / (chord @ chord) * chord; perp /= np.linalg.norm(perp)
        m = 0.5 * (start + goal) + rng.uniform(0.8, 1.5) * np.linalg.norm(chord) * perp       # on the bisector: |start - m| = |goal - m|
        rad2 = float((start - m) @ (start - m))
        Q = np.concatenate([Q, 2.0 * np.eye(d)[None]]); a = np.vstack([a, -2.0 * m]); c = np.append(c, m @ m - rad2)
        O = O + 1
    out = dict(d=d, T=T, K=1, O=O, x0=x0.ravel(), start=start, goal=goal, link_len=np.ones(d),
               point_link=np.zeros(1, dtype=np.int32), point_frac=np.ones(1), obstacles=np.zeros((O, 3)), reach=False,
               quad_Q=Q, quad_a=a, quad_c=c)
    if n_eq:
        out["quad_n_eq"] = int(n_eq)
    if groups is not None:
        out["groups"] = block_groups(T, False, groups)
    if vel_limit is not None:
        out["vmax"] = float(vel_limit)
    if joint_limit is not None:
        out["jlo"] = np.minimum(start, goal) - float(joint_limit)
        out["jhi"] = np.maximum(start, goal) + float(joint_limit)
    return out


_PROGRAMS = {}


def corridor_program(d):
    """The rows of the program-family workload, compiled once per state dimension: two keep-out discs whose radius
    breathes with the angle around them (parameters: centre, radius, ripple), a wavy wall above the path and a soft
    exponential bump from below -- nothing of it quadratic, all of it closed form."""
    if d not in _PROGRAMS:
        from .rowexpr import X, P, sin, cos, sqrt, exp, compile_rows
        rows = []
        for o in range(2):
            cx, cy, rad, rip = P(4 * o), P(4 * o + 1), P(4 * o + 2), P(4 * o + 3)
            dx, dy = X(0) - cx, X(1) - cy
            dist = sqrt(dx ** 2 + dy ** 2 + 1e-12)
            rows.append(rad * (1.0 + rip * (dx / dist) * (dy / dist)) - dist)          # disc with a sin(2 phi) ripple
        rows.append(X(1) - (P(8) + P(9) * sin(P(10) * X(0))))                          # below a wavy wall
        rows.append(P(11) - 0.5 * exp(-((X(0) - P(12)) ** 2) * 4.0) - X(1))            # above a floor with a bump
        if d > 2:
            rows[-1] = rows[-1] + 0.05 * cos(X(2))                                     # a third coordinate enters one row
        _PROGRAMS[d] = compile_rows(rows)
    return _PROGRAMS[d]


def variant_program(variant, d):
    """Programs of the r03 variants of the program family (compiled once per (variant, dof)):

    "sweep"     span 2, dof >= 2: the MIDPOINT of the step (x_t, x_t+1) keeps out of two discs whose radius grows with the
                squared step length (a swept-volume style keep-out: faster motion needs more clearance), and a step-length
                limit that depends on the height of the midpoint -- inequality rows on two timesteps;
    "dynamics"  span 2, dof = 3 (px, py, heading): the unicycle step px' - px = v cos(phi), py' - py = v sin(phi) as two
                EQUALITY rows on (x_t, x_t+1) (v = p[0]: the reference's EqExpr on an Expr, lowered to the abs penalty),
                plus a keep-out disc on x_t and a turn-rate limit |phi' - phi| <= p[5] as inequality rows;
    "curve"     span 1, dof = 3: the third coordinate follows z = p[8] sin(p[9] x) (an equality row on ONE timestep) and the
                two rippled discs of the corridor program keep out;
    "attract"   span 1, dof >= 2: the corridor rows plus a NON-QUADRATIC OBJECTIVE TERM per timestep, a Gaussian well
                -p[13] exp(-|x - g|^2 / 0.18) around g = p[14:16] (non-convex away from g: the eigenvalue shift acts);
    "accel"     (r04) span 3, dof >= 2: the squared second difference |x_t - 2 x_t+1 + x_t+2|^2 <= p[6]^2 (1 + p[7] y_mid)^2 -- an
                acceleration limit that tightens with height -- and two keep-out discs on the MIDDLE timestep;
    "jerk"      (r04) span 4, dof >= 2: the squared third difference |x_t+3 - 3 x_t+2 + 3 x_t+1 - x_t|^2 <= p[6]^2, a keep-out disc
                on the centroid of the four points whose radius grows with the squared chord |x_t+3 - x_t|^2, one on x_t+1."""
    key = (variant, d)
    if key in _PROGRAMS:
        return _PROGRAMS[key]
    from .rowexpr import X, P, sin, cos, sqrt, exp, compile_rows
    if variant == "sweep":
        rows = []
        mx, my = 0.5 * (X(0) + X(d)), 0.5 * (X(1) + X(d + 1))
        step2 = (X(d) - X(0)) ** 2 + (X(d + 1) - X(1)) ** 2
        for o in range(2):
            dx, dy = mx - P(3 * o), my - P(3 * o + 1)
            rows.append(P(3 * o + 2) * (1.0 + P(6) * step2) - sqrt(dx ** 2 + dy ** 2 + 1e-12))
        rows.append(step2 - (P(7) + P(8) * my) ** 2)                  # shorter steps where the midpoint is low
        prog = compile_rows(rows, span=2)
    elif variant == "accel":
        rows = []
        for o in range(2):
            dx, dy = X(d) - P(3 * o), X(d + 1) - P(3 * o + 1)
            rows.append(P(3 * o + 2) - sqrt(dx ** 2 + dy ** 2 + 1e-12))
        ax, ay = X(0) - 2.0 * X(d) + X(2 * d), X(1) - 2.0 * X(d + 1) + X(2 * d + 1)
        rows.append(ax ** 2 + ay ** 2 - (P(6) * (1.0 + P(7) * X(d + 1))) ** 2)
        prog = compile_rows(rows, span=3)
    elif variant == "jerk":
        jx, jy = X(3 * d) - 3.0 * X(2 * d) + 3.0 * X(d) - X(0), X(3 * d + 1) - 3.0 * X(2 * d + 1) + 3.0 * X(d + 1) - X(1)
        cx, cy = 0.25 * (X(0) + X(d) + X(2 * d) + X(3 * d)), 0.25 * (X(1) + X(d + 1) + X(2 * d + 1) + X(3 * d + 1))
        chord2 = (X(3 * d) - X(0)) ** 2 + (X(3 * d + 1) - X(1)) ** 2
        rows = [P(2) * (1.0 + P(7) * chord2) - sqrt((cx - P(0)) ** 2 + (cy - P(1)) ** 2 + 1e-12),
                P(5) - sqrt((X(d) - P(3)) ** 2 + (X(d + 1) - P(4)) ** 2 + 1e-12),
                jx ** 2 + jy ** 2 - P(6) ** 2]
        prog = compile_rows(rows, span=4)
    elif variant == "dynamics":
        assert d == 3
        ineq = [P(3) - sqrt((X(0) - P(1)) ** 2 + (X(1) - P(2)) ** 2 + 1e-12),            # keep-out disc on x_t
                (X(5) - X(2)) ** 2 - P(5) ** 2]                                            # turn-rate limit
        eq = [X(3) - X(0) - P(0) * cos(X(2)), X(4) - X(1) - P(0) * sin(X(2))]             # unicycle step, speed p[0]
        prog = compile_rows(ineq, eq_rows=eq, span=2)
    elif variant == "curve":
        assert d == 3
        rows = []
        for o in range(2):
            cx, cy, rad, rip = P(4 * o), P(4 * o + 1), P(4 * o + 2), P(4 * o + 3)
            dx, dy = X(0) - cx, X(1) - cy
            dist = sqrt(dx ** 2 + dy ** 2 + 1e-12)
            rows.append(rad * (1.0 + rip * (dx / dist) * (dy / dist)) - dist)
        prog = compile_rows(rows, eq_rows=[X(2) - P(8) * sin(P(9) * X(0))])
    elif variant == "attract":
        base = corridor_program(d)
        # (rebuild the corridor rows as expressions: compile_rows wants Nodes, so the rows are written out again)
        rows = []
        for o in range(2):
            cx, cy, rad, rip = P(4 * o), P(4 * o + 1), P(4 * o + 2), P(4 * o + 3)
            dx, dy = X(0) - cx, X(1) - cy
            dist = sqrt(dx ** 2 + dy ** 2 + 1e-12)
            rows.append(rad * (1.0 + rip * (dx / dist) * (dy / dist)) - dist)
        rows.append(X(1) - (P(8) + P(9) * sin(P(10) * X(0))))
        rows.append(P(11) - 0.5 * exp(-((X(0) - P(12)) ** 2) * 4.0) - X(1))
        if d > 2:
            rows[-1] = rows[-1] + 0.05 * cos(X(2))
        well = -(P(13) * exp(-((X(0) - P(14)) ** 2 + (X(1) - P(15)) ** 2) / 0.18))
        prog = compile_rows(rows, objective=well)
        assert prog.n_rows == base.n_rows
    else:
        raise ValueError("unknown program variant %r" % (variant,))
    _PROGRAMS[key] = prog
    return prog


def make_program_problem(i, d=2, T=20, noise=0.03, groups=None, vel_limit=None, joint_limit=None, variant=None):
    """Seeded problem i of the program family (SCO_FAM_STATE_PROGRAM): the rows of corridor_program(d) with per-problem
    parameters.  Same dictionary layout as make_problem (K = 1, O = 4 rows per timestep).  ``variant``: see variant_program."""
    if variant is not None:
        return make_program_variant(i, variant, d=d, T=T, noise=noise, groups=groups, vel_limit=vel_limit, joint_limit=joint_limit)
    rng = np.random.default_rng(9000 + i)
    start = np.concatenate([[-1.0, rng.uniform(-0.2, 0.2)], rng.uniform(-0.3, 0.3, size=d - 2)])
    goal = np.concatenate([[1.0, rng.uniform(-0.2, 0.2)], rng.uniform(-0.3, 0.3, size=d - 2)])
    s = np.linspace(0.0, 1.0, T)[:, None]
    x0 = (1 - s) * start[None, :] + s * goal[None, :] + noise * rng.standard_normal((T, d))
    par = np.zeros(13)
    for o in range(2):
        a = rng.uniform(0.25, 0.75)
        par[4 * o:4 * o + 2] = (1 - a) * start[:2] + a * goal[:2] + 0.05 * rng.standard_normal(2)
        par[4 * o + 2] = rng.uniform(0.12, 0.22); par[4 * o + 3] = rng.uniform(-0.3, 0.3)
    par[8:11] = (rng.uniform(0.55, 0.75), rng.uniform(0.05, 0.15), rng.uniform(2.0, 4.0))
    par[11:13] = (-rng.uniform(0.6, 0.8), rng.uniform(-0.3, 0.3))
    prog = corridor_program(d)
    out = dict(d=d, T=T, K=1, O=prog.n_rows, x0=x0.ravel(), start=start, goal=goal, link_len=np.ones(d),
               point_link=np.zeros(1, dtype=np.int32), point_frac=np.ones(1), obstacles=np.zeros((prog.n_rows, 3)), reach=False,
               row_program=prog, row_params=par)
    if groups is not None:
        out["groups"] = block_groups(T, False, groups)
    if vel_limit is not None:
        out["vmax"] = float(vel_limit)
    if joint_limit is not None:
        out["jlo"] = np.minimum(start, goal) - float(joint_limit)
        out["jhi"] = np.maximum(start, goal) + float(joint_limit)
    return out


def make_program_variant(i, variant, d=2, T=12, noise=0.03, groups=None, vel_limit=None, joint_limit=None):
    """Seeded problem i of a r03 variant of the program family (variant_program)."""
    # (the r03 variants keep their seeds; the r04 ones follow)
    order = sorted(["sweep", "dynamics", "curve", "attract"]) + ["accel", "jerk"]
    rng = np.random.default_rng(9500 + 97 * order.index(variant) + i)
    prog = variant_program(variant, d)
    start = np.concatenate([[-1.0, rng.uniform(-0.2, 0.2)], rng.uniform(-0.3, 0.3, size=d - 2)])
    goal = np.concatenate([[1.0, rng.uniform(-0.2, 0.2)], rng.uniform(-0.3, 0.3, size=d - 2)])
    if variant == "dynamics":
        head = np.arctan2(goal[1] - start[1], goal[0] - start[0])
        start[2] = head; goal[2] = head
    if variant == "curve":
        amp, freq = rng.uniform(0.15, 0.3), rng.uniform(1.5, 3.0)
        start[2] = amp * np.sin(freq * start[0]); goal[2] = amp * np.sin(freq * goal[0])       # the pins sit on the curve
    s = np.linspace(0.0, 1.0, T)[:, None]
    x0 = (1 - s) * start[None, :] + s * goal[None, :] + noise * rng.standard_normal((T, d))

    def disc(lo=0.3, hi=0.7):
        a = rng.uniform(lo, hi)
        return (1 - a) * start[:2] + a * goal[:2] + 0.05 * rng.standard_normal(2)

    if variant == "sweep":
        par = np.zeros(9)
        for o in range(2):
            par[3 * o:3 * o + 2] = disc(0.25 + 0.3 * o, 0.45 + 0.3 * o); par[3 * o + 2] = rng.uniform(0.12, 0.2)
        par[6] = rng.uniform(1.0, 3.0)                               # clearance grows with the squared step
        step = np.linalg.norm(goal[:2] - start[:2]) / (T - 1)
        par[7] = rng.uniform(1.6, 2.2) * step; par[8] = rng.uniform(0.0, 0.4) * step
    elif variant == "accel":
        par = np.zeros(8)
        for o in range(2):
            par[3 * o:3 * o + 2] = disc(0.25 + 0.3 * o, 0.45 + 0.3 * o); par[3 * o + 2] = rng.uniform(0.12, 0.2)
        step = np.linalg.norm(goal[:2] - start[:2]) / (T - 1)
        par[6] = rng.uniform(0.25, 0.45) * step; par[7] = rng.uniform(-0.3, 0.3)
    elif variant == "jerk":
        par = np.zeros(8)
        par[0:2] = disc(0.3, 0.45); par[2] = rng.uniform(0.1, 0.16)
        par[3:5] = disc(0.55, 0.7); par[5] = rng.uniform(0.12, 0.2)
        step = np.linalg.norm(goal[:2] - start[:2]) / (T - 1)
        par[6] = rng.uniform(0.3, 0.6) * step; par[7] = rng.uniform(0.5, 1.5)
    elif variant == "dynamics":
        par = np.zeros(6)
        par[0] = rng.uniform(1.05, 1.2) * np.linalg.norm(goal[:2] - start[:2]) / (T - 1)        # a little faster than the chord
        par[1:3] = disc(0.4, 0.6); par[3] = rng.uniform(0.1, 0.18)
        par[5] = rng.uniform(0.5, 0.9)
    elif variant == "curve":
        par = np.zeros(10)
        for o in range(2):
            par[4 * o:4 * o + 2] = disc(0.25 + 0.3 * o, 0.45 + 0.3 * o)
            par[4 * o + 2] = rng.uniform(0.12, 0.2); par[4 * o + 3] = rng.uniform(-0.3, 0.3)
        par[8:10] = (amp, freq)
    else:                                                            # attract
        par = np.zeros(16)
        for o in range(2):
            a = rng.uniform(0.25, 0.75)
            par[4 * o:4 * o + 2] = (1 - a) * start[:2] + a * goal[:2] + 0.05 * rng.standard_normal(2)
            par[4 * o + 2] = rng.uniform(0.12, 0.22); par[4 * o + 3] = rng.uniform(-0.3, 0.3)
        par[8:11] = (rng.uniform(0.55, 0.75), rng.uniform(0.05, 0.15), rng.uniform(2.0, 4.0))
        par[11:13] = (-rng.uniform(0.6, 0.8), rng.uniform(-0.3, 0.3))
        par[13] = rng.uniform(0.2, 0.5); par[14:16] = (rng.uniform(-0.4, 0.4), rng.uniform(0.25, 0.5))
    out = dict(d=d, T=T, K=1, O=prog.n_rows, x0=x0.ravel(), start=start, goal=goal, link_len=np.ones(d),
               point_link=np.zeros(1, dtype=np.int32), point_frac=np.ones(1), obstacles=np.zeros((prog.n_rows, 3)), reach=False,
               row_program=prog, row_params=par)
    if groups is not None:
        out["groups"] = block_groups(T - prog.span + 1, False, groups)
    if vel_limit is not None:
        out["vmax"] = float(vel_limit)
    if joint_limit is not None:
        out["jlo"] = np.minimum(start, goal) - float(joint_limit)
        out["jhi"] = np.maximum(start, goal) + float(joint_limit)
    return out


def general_rows(i, d, T, start, goal):
    """r04: GENERAL affine rows of problem i (prob.add_cnt_expr(BoundExpr(LEqExpr / EqExpr(AffExpr(A, 0), rhs), traj)), prob.py:126-131,
    317-346): the pattern depends on (d, T) only, coefficients and right-hand sides on the problem.  Inequalities first:
    for every odd timestep t < T - 1 a coupling of two joints  theta[t][0] + c_t theta[t][1] <= b_t  and a two-step limit
    theta[t+1][0] - theta[t-1][0] <= g_t; then ONE equality at the middle timestep  theta[m][0] - a theta[m][1] = e.  Right-hand sides
    sit a margin away from the straight line between start and goal, so some rows are active and the problem stays feasible.
    Returns dict(A (n_rows, d T) dense, rhs, is_eq)."""
    rng = np.random.default_rng(8000 + i)
    s = np.linspace(0.0, 1.0, T)[:, None]
    line = (1 - s) * start[None, :] + s * goal[None, :]
    rows, rhs, eq = [], [], []
    for t in range(1, T - 1, 2):
        c = rng.uniform(0.5, 1.5)
        r = np.zeros(d * T); r[t * d] = 1.0; r[t * d + 1] = c
        rows.append(r); rhs.append(line[t, 0] + c * line[t, 1] + rng.uniform(0.02, 0.3)); eq.append(0)
        r = np.zeros(d * T); r[(t + 1) * d] = 1.0; r[(t - 1) * d] = -1.0
        rows.append(r); rhs.append(line[t + 1, 0] - line[t - 1, 0] + rng.uniform(0.02, 0.2)); eq.append(0)
    m_ = T // 2
    a = rng.uniform(0.5, 1.5)
    r = np.zeros(d * T); r[m_ * d] = 1.0; r[m_ * d + 1] = -a
    rows.append(r); rhs.append(line[m_, 0] - a * line[m_, 1] + rng.uniform(-0.05, 0.05)); eq.append(1)
    return dict(A=np.array(rows), rhs=np.array(rhs), is_eq=np.array(eq, dtype=np.int32))


def make_problem(i, obj_weights=False, per_step=False, lin_rows=False, circles=0, acc_weights=False, **kw):
    """Seeded problem i of the batch; see _make_problem for the families.  r04 (wider template):
    obj_weights=True adds per-joint weights w_j in [0.4, 3] of the smoothing objective (``obj_w``; a QuadExpr built from a
    weighted difference matrix, prob.py:88-104, 348-367); per_step=True (program family) gives every timestep its own
    parameter vector (``row_params`` of shape (T, n_params): obstacles that drift and pulse along the horizon -- each
    timestep's Expr closes over its own data, expr.py:22-41).  Both draw from their own generators: every other number of
    the problem is what it is without them.  lin_rows=True adds the general affine rows of ``general_rows`` (``lin_gen``)."""
    out = _make_problem(i, **kw)
    if circles:
        # r04: a SECOND kind of non-linear rows on every timestep: `circles` keep-out discs for the point (x[0], x[1]) IN FRONT of the
        # program's rows -- two BoundExprs on one timestep Variable (prob.py:112-144)
        prog = out.get("row_program")
        if prog is None or prog.span != 1:
            raise ValueError("circles: program family with blocks on one timestep")
        rng = np.random.default_rng(6000 + i)
        along = np.sort(rng.uniform(0.25, 0.75, size=circles))
        centre = (1 - along[:, None]) * out["start"][None, :2] + along[:, None] * out["goal"][None, :2] + 0.06 * rng.standard_normal((circles, 2))
        obs = np.zeros((circles + prog.n_rows, 3))
        obs[:circles, :2] = centre; obs[:circles, 2] = rng.uniform(0.06, 0.14, size=circles)
        out["obstacles"] = obs; out["O"] = circles + prog.n_rows; out["circle_rows"] = int(circles)
    if lin_rows:
        out["lin_gen"] = general_rows(i, out["d"], out["T"], out["start"], out["goal"])
    if obj_weights:
        out["obj_w"] = np.random.default_rng(7000 + i).uniform(0.4, 3.0, size=out["d"])
    if acc_weights:          # r04: acceleration term sum_t sum_j a_j (x[t+2][j] - 2 x[t+1][j] + x[t][j])^2 in the quadratic objective
        out["acc_w"] = np.random.default_rng(7500 + i).uniform(0.2, 2.0, size=out["d"])
    if per_step:
        if out.get("row_program") is None:
            raise ValueError("per_step needs the program family")
        par = np.asarray(out["row_params"], dtype=np.float64)
        t = np.arange(out["T"], dtype=np.float64)[:, None]
        out["row_params"] = par[None, :] + 0.04 * np.abs(par)[None, :] * np.sin(0.9 * t + np.arange(par.shape[0])[None, :])
    return out


def smooth_Q(d, T, w=None, a=None):
    """Q of the quadratic objective in QuadExpr's 0.5 x'Qx form: sum_t sum_j w_j (x[t+1][j] - x[t][j])^2 (w None = 1) plus, with a,
    sum_t sum_j a_j (x[t+2][j] - 2 x[t+1][j] + x[t][j])^2 -- added entry by entry in (t, j) order, first differences first."""
    n = d * T
    Q = np.zeros((n, n))
    w = np.ones(d) if w is None else np.asarray(w, dtype=np.float64)
    for t in range(T - 1):
        for j in range(d):
            i0, i1 = t * d + j, (t + 1) * d + j
            Q[i0, i0] += 2.0 * w[j]; Q[i1, i1] += 2.0 * w[j]; Q[i0, i1] -= 2.0 * w[j]; Q[i1, i0] -= 2.0 * w[j]
    if a is not None:
        c = np.array([1.0, -2.0, 1.0])
        for t in range(T - 2):
            for j in range(d):
                idx = np.array([t * d + j, (t + 1) * d + j, (t + 2) * d + j])
                Q[np.ix_(idx, idx)] += 2.0 * a[j] * np.outer(c, c)
    return Q


def step_params(pr, t):
    """Parameter vector of constraint block / objective term t of a program problem (shared or per timestep)."""
    par = pr["row_params"]
    return par[t] if np.ndim(par) == 2 else par


def _make_problem(i, d=7, T=20, K=5, O=2, noise=0.05, reach=False, groups=None, vel_limit=None, joint_limit=None,
                  ee_cost_weight=None, point=False, quadratic=False, program=False, variant=None, n_eq=0):
    """reach=True: the goal pin
    theta[T-1] = goal is replaced by the non-linear equality ee(theta[T-1]) = ee(goal)
    (EqExpr on an Expr: the abs-penalty path of prob.py:280-315); same random draws."""
    if program:
        return make_program_problem(i, d=d, T=T, groups=groups, vel_limit=vel_limit, joint_limit=joint_limit, variant=variant)
    if quadratic:
        return make_quadratic_problem(i, d=d, T=T, O=O, groups=groups, vel_limit=vel_limit, joint_limit=joint_limit, n_eq=n_eq)
    if point:
        return make_point_problem(i, d=d, T=T, O=O, groups=groups, vel_limit=vel_limit, joint_limit=joint_limit)
    is_reach = bool(reach)
    rng = np.random.default_rng(1000 + i)
    start = rng.uniform(-np.pi / 2, np.pi / 2, size=d)
    goal = rng.uniform(-np.pi / 2, np.pi / 2, size=d)
    s = np.linspace(0.0, 1.0, T)[:, None]
    x0 = (1 - s) * start[None, :] + s * goal[None, :] + noise * rng.standard_normal((T, d))
    link_len = np.full(d, 1.0 / d)
    reach = float(link_len.sum())
    ang = rng.uniform(0.0, 2 * np.pi, size=O)
    rad = reach * np.sqrt(rng.uniform(0.15 ** 2, 1.0, size=O))     # uniform on the annulus
    radius = rng.uniform(0.05, 0.15, size=O) * reach
    obstacles = np.stack([rad * np.cos(ang), rad * np.sin(ang), radius], axis=1)
    point_link, point_frac = default_points(d, K)
    out = dict(d=d, T=T, K=K, O=O, x0=x0.ravel(), start=start, goal=goal, link_len=link_len,
               point_link=point_link, point_frac=point_frac, obstacles=obstacles, reach=is_reach)
    if is_reach:
        out["target"] = ee_pos(goal, link_len)
    if groups is not None:
        out["groups"] = block_groups(T, is_reach, groups)      # prob.add_cnt_expr(..., group_ids=...)
    if vel_limit is not None:
        # joint-velocity limits |theta[t+1][j] - theta[t][j]| <= vmax: LINEAR inequalities, i.e.
        # LEqExpr(AffExpr) rows that go straight into every QP (prob.py:126-131, 317-346)
        out["vmax"] = float(vel_limit)
    if ee_cost_weight is not None:
        # a NON-QUADRATIC objective term per timestep, weight * ||ee(theta_t) - ee(goal)||^2: Prob.add_obj_expr routes
        # it to _nonquad_obj_exprs and every SQP iteration convexifies it to degree 2 (prob.py:88-104, 532-534)
        out["cost_weight"] = float(ee_cost_weight)
        out["cost_target"] = ee_pos(goal, link_len)
    if joint_limit is not None:
        # joint limits lo_j <= theta[t][j] <= hi_j for every timestep: two more LEqExpr(AffExpr) blocks.  The box
        # hugs the straight line between start and goal (margin `joint_limit`), so that avoiding the obstacles
        # runs into it.
        out["jlo"] = np.minimum(start, goal) - float(joint_limit)
        out["jhi"] = np.maximum(start, goal) + float(joint_limit)
    return out


def make_batch(B, first=0, **kw):
    """Stacked arrays of problems first .. first+B-1 in the layout sco_sqp_load takes."""
    probs = [make_problem(first + i, **kw) for i in range(B)]
    p0 = probs[0]
    extra = dict(reach=True, target=np.stack([p["target"] for p in probs])) if p0.get("reach") else {}
    if p0.get("point"):
        extra["point"] = True
    if p0.get("row_program") is not None:
        extra["row_program"] = p0["row_program"]; extra["row_params"] = np.stack([p["row_params"] for p in probs])
    if p0.get("quad_Q") is not None:
        extra["quad_Q"] = np.stack([p["quad_Q"] for p in probs]); extra["quad_a"] = np.stack([p["quad_a"] for p in probs])
        extra["quad_c"] = np.stack([p["quad_c"] for p in probs])
        if p0.get("quad_n_eq"):
            extra["quad_n_eq"] = p0["quad_n_eq"]
    if p0.get("groups") is not None:
        extra["groups"] = p0["groups"]
    if p0.get("vmax") is not None:
        extra["vmax"] = np.array([p["vmax"] for p in probs])
    if p0.get("cost_weight") is not None:
        extra["cost_weight"] = np.array([p["cost_weight"] for p in probs])
        extra["cost_target"] = np.stack([p["cost_target"] for p in probs])
    if p0.get("jlo") is not None:
        extra["jlo"] = np.stack([p["jlo"] for p in probs]); extra["jhi"] = np.stack([p["jhi"] for p in probs])
    if p0.get("obj_w") is not None:
        extra["obj_w"] = np.stack([p["obj_w"] for p in probs])
    if p0.get("acc_w") is not None:
        extra["acc_w"] = np.stack([p["acc_w"] for p in probs])
    if p0.get("circle_rows"):
        extra["circle_rows"] = p0["circle_rows"]
    if p0.get("lin_gen") is not None:
        # shared CSR pattern = the union of the problems' non-zeros (a problem's structural zero is a zero coefficient), values
        # and right-hand sides per problem (sco_sqp_create_rows / sco_sqp_load_linear_rows)
        mask = np.any(np.stack([p["lin_gen"]["A"] != 0 for p in probs]), axis=0)
        row_ptr = np.concatenate([[0], np.cumsum(mask.sum(axis=1))]).astype(np.int32)
        r, c = np.nonzero(mask)
        extra["lin_rows"] = (row_ptr, c.astype(np.int32), p0["lin_gen"]["is_eq"].astype(np.int32))
        extra["lin_vals"] = np.stack([p["lin_gen"]["A"][r, c] for p in probs])
        extra["lin_rhs"] = np.stack([p["lin_gen"]["rhs"] for p in probs])
    return dict(
        d=p0["d"], T=p0["T"], K=p0["K"], O=p0["O"], B=B, **extra,
        x0=np.stack([p["x0"] for p in probs]),
        start=np.stack([p["start"] for p in probs]),
        goal=np.stack([p["goal"] for p in probs]),
        link_len=np.stack([p["link_len"] for p in probs]),
        point_link=p0["point_link"].astype(np.int32), point_frac=p0["point_frac"].copy(),
        obstacles=np.stack([p["obstacles"] for p in probs]),
    ), probs
