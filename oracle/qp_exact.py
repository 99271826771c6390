"""ADMM-free reference solutions of the QPs the SCO loop hands to its solver -- TEST INFRASTRUCTURE.

    min 1/2 x'Px + q'x   s.t.   l <= Ax <= u        (the problem behind osqp_utils.py:195-216)

Everything else in oracle/ that solves a QP is OUR restatement of OSQP's ADMM (oracle/osqp_ref.c); the GPU kernels are
a second ADMM.  This module shares nothing with either: a dense Mehrotra predictor-corrector interior-point method finds
the active set, then the equality-constrained KKT system of that active set is solved with iterative refinement whose
residuals are accumulated in extended precision (numpy.longdouble, 64-bit mantissa on x86), and the optimality
conditions are checked in extended precision:

    stationarity   P x + q + A' y = 0
    primal         l <= A x <= u
    dual sign      y_i <= 0 where only the lower bound is active, >= 0 where only the upper one is, 0 on inactive rows

`check_kkt` needs no solver at all -- it is residual arithmetic on a claimed (x, y) -- so a stored solution certifies
itself.  Used by tests/test_qp_exact.py and tests/golden/make_qp_exact.py to pin the golden QPs of the 7-DOF x 20
workload: the converging ones against |x_ADMM - x*|, and the one the reference's compounded penalty (Q1) makes too stiff
for a fixed rho: its ADMM iterate after max_iter iterations is shown to violate OSQP's own termination test by
recomputing the residuals from the returned (x, y) here.
"""
import numpy as np

INF = 1e20            # bounds beyond this are "no bound" (OSQP uses 1e30 internally; the reference passes np.inf)


def _dedupe_rows(A, l, u):
    """Identical rows (same coefficients and bounds) are one constraint; quirk Q2 of the reference appends the
    penalty rows again on every update_obj (prob.py:508-509).  Returns (A, l, u, group) with group[i] = index of
    original row i in the reduced set."""
    keys = {}
    keep = []
    group = np.zeros(A.shape[0], dtype=np.int64)
    for i in range(A.shape[0]):
        k = (A[i].tobytes(), float(l[i]), float(u[i]))
        if k not in keys:
            keys[k] = len(keep)
            keep.append(i)
        group[i] = keys[k]
    keep = np.array(keep, dtype=np.int64)
    return A[keep], l[keep], u[keep], group


def interior_point(P, q, A, l, u, tol=1e-10, max_iter=200):
    """Mehrotra predictor-corrector on   min 1/2 x'Px + q'x,  E x = f,  G x <= h   (dense).  Returns x and the
    multipliers of the rows of A in OSQP's sign convention (y > 0 at an active upper bound, < 0 at a lower one)."""
    P = np.asarray(P, dtype=np.float64); q = np.asarray(q, dtype=np.float64).ravel()
    A = np.asarray(A, dtype=np.float64); l = np.asarray(l, dtype=np.float64).ravel(); u = np.asarray(u, dtype=np.float64).ravel()
    n = q.shape[0]
    eq = np.where((u - l) <= 0.0)[0]
    up = np.where(((u - l) > 0.0) & (u < INF))[0]
    lo = np.where(((u - l) > 0.0) & (l > -INF))[0]
    E, f = A[eq], u[eq]
    G = np.vstack([A[up], -A[lo]]) if len(up) + len(lo) else np.zeros((0, n))
    h = np.concatenate([u[up], -l[lo]])
    me, mi = E.shape[0], G.shape[0]
    x = np.zeros(n); s = np.ones(mi); z = np.ones(mi); nu = np.zeros(me)
    if mi:
        s = np.maximum(h - G @ x, 1.0)

    def solve(rd, rp_e, rp_i, rc):
        # [P G' E'; G -S/Z.. ] eliminated to (P + G' diag(z/s) G) dx + E' dnu = ...
        w = z / s
        H = P + (G.T * w) @ G + 1e-11 * np.eye(n)
        r1 = -rd + G.T @ ((rc - z * rp_i) / s)
        if me:
            K = np.block([[H, E.T], [E, -1e-11 * np.eye(me)]])
            rhs = np.concatenate([r1, -rp_e])
            sol = np.linalg.solve(K, rhs)
            sol += np.linalg.solve(K, rhs - K @ sol)
            dx, dnu = sol[:n], sol[n:]
        else:
            dx = np.linalg.solve(H, r1); dnu = np.zeros(0)
        ds = -rp_i - G @ dx
        dz = -(rc + z * ds) / s
        return dx, dnu, ds, dz

    for _ in range(max_iter):
        rd = P @ x + q + G.T @ z + E.T @ nu
        rp_e = E @ x - f
        rp_i = G @ x + s - h
        mu = float(s @ z) / mi if mi else 0.0
        scale = 1.0 + max(np.abs(q).max(), np.abs(h).max() if mi else 0.0)
        if max(np.abs(rd).max(), np.abs(rp_e).max() if me else 0.0, np.abs(rp_i).max() if mi else 0.0) < tol * scale and mu < tol:
            break
        if not mi:
            dx, dnu, _, _ = solve(rd, rp_e, rp_i, np.zeros(0))
            x += dx; nu += dnu
            continue
        dxa, dnua, dsa, dza = solve(rd, rp_e, rp_i, s * z)
        aa = min(1.0, _step(s, dsa), _step(z, dza))
        mu_aff = float((s + aa * dsa) @ (z + aa * dza)) / mi
        sigma = (mu_aff / mu) ** 3
        dx, dnu, ds, dz = solve(rd, rp_e, rp_i, s * z + dsa * dza - sigma * mu)
        al = min(1.0, 0.995 * _step(s, ds), 0.995 * _step(z, dz))
        x += al * dx; nu += al * dnu; s += al * ds; z += al * dz
    y = np.zeros(A.shape[0])
    y[eq] = nu
    y[up] += z[:len(up)]
    y[lo] -= z[len(up):]
    return x, y


def _step(v, dv):
    neg = dv < 0
    return float(np.min(-v[neg] / dv[neg])) if np.any(neg) else 1.0


def polish(P, q, A, l, u, x0, y0, act_tol=1e-7, refine=8):
    """Equality-constrained QP on the active set read off (x0, y0); iterative refinement with the residual of the KKT
    system accumulated in extended precision.  Returns (x, y) as longdouble arrays."""
    LD = np.longdouble
    Ax = A @ x0
    at_l = (l > -INF) & ((Ax - l) < act_tol * (1 + np.abs(l))) & ((y0 < 0) | ((u - l) <= 0))
    at_u = (u < INF) & ((u - Ax) < act_tol * (1 + np.abs(u))) & ((y0 > 0) | ((u - l) <= 0))
    act = np.where(at_l | at_u)[0]
    b = np.where(at_u[act], u[act], l[act])
    Aa = A[act]
    n, k = q.shape[0], len(act)
    K = np.block([[P, Aa.T], [Aa, np.zeros((k, k))]])
    rhs = np.concatenate([-q, b])
    # the active rows can be linearly dependent (a variable pinned by two rows): minimum-norm multipliers
    Kp = np.linalg.pinv(K, rcond=1e-13)
    sol = (Kp @ rhs).astype(LD)
    Kl, rl = K.astype(LD), rhs.astype(LD)
    for _ in range(refine):
        res = rl - Kl @ sol
        sol = sol + (Kp @ res.astype(np.float64)).astype(LD)
    x = sol[:n]
    y = np.zeros(A.shape[0], dtype=LD)
    y[act] = sol[n:]
    return x, y


def check_kkt(P, q, A, l, u, x, y):
    """Optimality conditions of (x, y) in extended precision; a dict of infinity-norm violations (all ~0 at the
    optimum).  No solver involved."""
    LD = np.longdouble
    P = np.asarray(P).astype(LD); A = np.asarray(A).astype(LD)
    q = np.asarray(q).astype(LD).ravel(); l = np.asarray(l).astype(LD).ravel(); u = np.asarray(u).astype(LD).ravel()
    x = np.asarray(x).astype(LD).ravel(); y = np.asarray(y).astype(LD).ravel()
    Ax = A @ x
    stat = P @ x + q + A.T @ y
    prim = np.maximum(np.maximum(l - Ax, Ax - u), 0)
    slack_l = np.where(l > -INF, Ax - l, np.inf)
    slack_u = np.where(u < INF, u - Ax, np.inf)
    ineq = (u - l) > 0
    # complementarity with the sign convention: y+ needs the upper bound active, y- the lower one
    comp = np.where(ineq, np.maximum(y, 0) * np.minimum(slack_u, 1e30) + np.maximum(-y, 0) * np.minimum(slack_l, 1e30), 0)
    bad_sign = np.where(ineq & ~(u < INF), np.maximum(y, 0), 0) + np.where(ineq & ~(l > -INF), np.maximum(-y, 0), 0)
    return dict(stationarity=float(np.abs(stat).max()), primal=float(prim.max()) if prim.size else 0.0,
                complementarity=float(np.abs(comp).max()) if comp.size else 0.0,
                dual_sign=float(bad_sign.max()) if bad_sign.size else 0.0,
                objective=float(0.5 * x @ (P @ x) + q @ x))


def solve_exact(P, q, A, l, u):
    """x*, y* (float64 copies of the extended-precision solution, y expanded back to the caller's rows with the
    multiplier of a duplicated constraint carried by its first copy) and the KKT report."""
    P = np.asarray(P, dtype=np.float64); A = np.asarray(A, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64).ravel()
    l = np.maximum(np.asarray(l, dtype=np.float64).ravel(), -1e30); u = np.minimum(np.asarray(u, dtype=np.float64).ravel(), 1e30)
    Ar, lr, ur, group = _dedupe_rows(A, l, u)
    x0, y0 = interior_point(P, q, Ar, lr, ur)
    x, yr = polish(P, q, Ar, lr, ur, x0, y0)
    rep = check_kkt(P, q, Ar, lr, ur, x, yr)
    y = np.zeros(A.shape[0])
    first = {}
    for i, g in enumerate(group):
        if g not in first:
            first[g] = i
            y[i] = float(yr[g])
    return np.asarray(x, dtype=np.float64), y, rep


def osqp_residuals(P, q, A, l, u, x, y):
    """OSQP's unscaled termination quantities recomputed from a returned (x, y) alone: the dual residual
    |Px + q + A'y|_inf exactly, and a LOWER bound of the primal residual |Ax - z|_inf (z in [l, u] is not returned;
    the projection of Ax is the closest admissible z), with the matching tolerances' scale terms."""
    P = np.asarray(P, dtype=np.float64); A = np.asarray(A, dtype=np.float64)
    q = np.asarray(q, dtype=np.float64).ravel(); x = np.asarray(x, dtype=np.float64).ravel(); y = np.asarray(y, dtype=np.float64).ravel()
    Ax = A @ x
    z = np.clip(Ax, l, u)
    Px, Aty = P @ x, A.T @ y
    return dict(dual=float(np.abs(Px + q + Aty).max()), primal_lower_bound=float(np.abs(Ax - z).max()),
                prim_scale=float(max(np.abs(Ax).max(), np.abs(z).max())),
                dual_scale=float(max(np.abs(Px).max(), np.abs(Aty).max(), np.abs(q).max())))
