"""Flat NumPy restatement of the reference penalty-SQP path -- TEST INFRASTRUCTURE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module.  It restates, on plain arrays instead of the reference's object
graph, what the following reference code does for a problem made of
  * one quadratic objective  1/2 x'Qx + a'x + b0          (QuadExpr)
  * non-quadratic objective terms  phi_k(x_I)              (Expr(f): numeric gradient + Hessian, degree-2 model)
  * affine equality / inequality rows                      (EqExpr/LEqExpr(AffExpr))
  * non-linear  g(x_I) <= val  /  h(x_I) = val  blocks     (LEqExpr/EqExpr(Expr(f)))
with every citation relative to /root/reference/sco_py:

  S1  Expr.eval / grad                          expr.py:34-41, 78-100 (memo on round(x, 6): Q3)
  S2  Expr.convexify(deg 1), Eq/LEqExpr.convexify   expr.py:139-142, 314-371
      Expr.convexify(deg 2) with the eigenvalue shift  expr.py:102-128, 143-153; prob.py:88-104, 532-534
  S3  Prob.update_obj and helpers               sco_osqp/prob.py:251-315, 414-512 (Q1, Q2)
  S4  Variable.add_trust_region                 sco_osqp/variable.py:37-45
  S5  osqp_utils.optimize assembly              sco_osqp/osqp_utils.py:136-193
  S6  OSQP                                      oracle/osqp_ref.c (third-party, restated)
  S7  Prob.get_value / get_approx_value / get_max_cnt_violation   prob.py:547-630
  S8  Solver._penalty_sqp / _min_merit_fn       sco_osqp/solver.py:62-283

It is pinned against the reference itself: tests/golden/make_golden.py runs the
reference's own modules (with the oracle ADMM at the ``osqp`` seam) on the same
problems and tests/test_golden.py checks that this file reproduces the recorded
QP sequence, decisions and solutions.
"""
import numpy as np
import scipy.sparse as sp

from . import osqp_ref

N_DIGS = 6
STEP_PROJECT, STEP_ACCEPT, STEP_SHRINK, STEP_YCONV, STEP_XCONV, STEP_BAD, STEP_GROUP = range(7)

# finite differences: same ladder as sco_py_amd/numdiff.py and fd_jacobian in csrc/sco_sqp.hip
FD_BASE, FD_LEVELS = 1.0 / 64.0, 4


def fd_jacobian(f, x):
    """Richardson-extrapolated central differences of f: R^n -> R^r at x (n,)."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    cols = []
    for j in range(n):
        h0 = FD_BASE * max(1.0, abs(x[j]))
        tab = []
        for k in range(FD_LEVELS):
            h = h0 / (2.0 ** k)
            xp = x.copy(); xm = x.copy()
            xp[j] += h; xm[j] -= h
            tab.append((f(xp) - f(xm)) / (2.0 * h))
        for i in range(1, FD_LEVELS):
            fac = 1.0 / (4.0 ** i - 1.0)
            tab = [tab[k] + (tab[k] - tab[k - 1]) * fac for k in range(1, len(tab))]
        cols.append(tab[0])
    return np.stack(cols, axis=1)


def fd_hessian(f, x):
    """Hessian of a scalar f at x (n,): second central differences on the same halving ladder, Richardson
    extrapolated -- the formula of sco_py_amd/numdiff.py:hessian (the stand-in for numdifftools.Hessian,
    expr.py:108) and of obj_hess in csrc/sco_sqp.hip."""
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    H = np.zeros((n, n))
    f0 = float(f(x))
    steps = [FD_BASE * max(1.0, abs(x[j])) for j in range(n)]
    for i in range(n):
        for j in range(i, n):
            tab = []
            for k in range(FD_LEVELS):
                hi, hj = steps[i] / (2.0 ** k), steps[j] / (2.0 ** k)
                if i == j:
                    xp = x.copy(); xm = x.copy()
                    xp[i] += hi; xm[i] -= hi
                    tab.append((float(f(xp)) - 2.0 * f0 + float(f(xm))) / (hi * hi))
                else:
                    xpp = x.copy(); xpm = x.copy(); xmp = x.copy(); xmm = x.copy()
                    xpp[i] += hi; xpp[j] += hj
                    xpm[i] += hi; xpm[j] -= hj
                    xmp[i] -= hi; xmp[j] += hj
                    xmm[i] -= hi; xmm[j] -= hj
                    tab.append((float(f(xpp)) - float(f(xpm)) - float(f(xmp)) + float(f(xmm))) / (4.0 * hi * hj))
            for lv in range(1, FD_LEVELS):
                fac = 1.0 / (4.0 ** lv - 1.0)
                tab = [tab[k] + (tab[k] - tab[k - 1]) * fac for k in range(1, len(tab))]
            H[i, j] = H[j, i] = tab[0]
    return H


def min_eig_jacobi(H, sweeps=12):
    """Smallest eigenvalue of a small symmetric matrix by cyclic Jacobi rotations (what the device does; the
    reference calls scipy.linalg.eigvalsh, expr.py:145 -- the two agree to rounding)."""
    A = np.array(H, dtype=np.float64)
    n = A.shape[0]
    for _ in range(sweeps):
        for p in range(n - 1):
            for q in range(p + 1, n):
                if A[p, q] == 0.0:
                    continue
                if abs(A[p, q]) < 1e-300:
                    continue
                theta = (A[q, q] - A[p, p]) / (2.0 * A[p, q])
                # |theta| beyond 1e150 would overflow theta^2: there sqrt(theta^2 + 1) = |theta| to the last bit
                t = (1.0 if theta >= 0 else -1.0) / (abs(theta) + (abs(theta) if abs(theta) > 1e150 else np.sqrt(theta * theta + 1.0)))
                c = 1.0 / np.sqrt(t * t + 1.0); s_ = t * c
                Rp, Rq = A[:, p].copy(), A[:, q].copy()
                A[:, p] = c * Rp - s_ * Rq; A[:, q] = s_ * Rp + c * Rq
                Rp, Rq = A[p, :].copy(), A[q, :].copy()
                A[p, :] = c * Rp - s_ * Rq; A[q, :] = s_ * Rp + c * Rq
    return float(np.min(np.diag(A)))


class ObjBlock(object):
    """One non-quadratic objective BoundExpr: scalar callable f on the block's own variables x[idx]
    (prob.py:88-104).  eval is memoised on the rounded point like every Expr (Q3, expr.py:34-41); gradient and
    Hessian are numeric and never cached (expr.py:61-69, 102-109)."""

    def __init__(self, f, idx):
        self.f = f
        self.idx = np.asarray(idx, dtype=np.int64)
        self._eval_cache = {}
        self.emulate_memo = True

    def eval(self, xb):
        if not self.emulate_memo:
            return float(self.f(xb))
        k = tuple(np.round(xb, N_DIGS))
        if k not in self._eval_cache:
            self._eval_cache[k] = float(self.f(xb))
        return self._eval_cache[k]

    def convexify(self, xb, exact_eig=True):              # expr.py:143-153 (eigvalsh, as the reference)
        H = fd_hessian(self.f, xb)
        lam = float(np.min(np.linalg.eigvalsh(H))) if exact_eig else min_eig_jacobi(H)
        if lam < 0:
            H = H - np.eye(H.shape[0]) * lam
        g = fd_jacobian(lambda v: np.array([float(self.f(v))]), xb)[0]
        A = g - xb.dot(H)
        b = 0.5 * xb.dot(H).dot(xb) - g.dot(xb) + self.eval(xb)
        return H, A, b


class Block(object):
    """One non-linear constraint BoundExpr: kind 'leq' | 'eq', callable f on the
    block's own variables x[idx], optional analytic jac, right-hand side val."""

    def __init__(self, kind, f, idx, val, jac=None, groups=None):
        assert kind in ("leq", "eq")
        self.kind, self.f, self.jac = kind, f, jac
        self.groups = list(groups) if groups is not None else ["all"]      # prob.py:135-142
        self.idx = np.asarray(idx, dtype=np.int64)
        self.val = np.asarray(val, dtype=np.float64).ravel()
        self.r = self.val.shape[0]
        self._eval_cache, self._grad_cache, self._cvx_cache = {}, {}, {}
        self.emulate_memo = True

    def _key(self, xb):
        return tuple(np.round(xb, N_DIGS))

    def eval(self, xb):                                   # expr.py:34-41
        if not self.emulate_memo:
            return np.asarray(self.f(xb), dtype=np.float64).ravel()
        k = self._key(xb)
        if k not in self._eval_cache:
            self._eval_cache[k] = np.asarray(self.f(xb), dtype=np.float64).ravel().copy()
        return self._eval_cache[k]

    def grad(self, xb):                                   # expr.py:78-100
        if self.jac is None:                              # numeric: never cached (Q4)
            return fd_jacobian(lambda v: np.asarray(self.f(v), dtype=np.float64).ravel(), xb)
        if not self.emulate_memo:
            return np.asarray(self.jac(xb), dtype=np.float64)
        k = self._key(xb)
        if k not in self._grad_cache:
            self._grad_cache[k] = np.asarray(self.jac(xb), dtype=np.float64).copy()
        return self._grad_cache[k].copy()

    def convexify(self, xb):                              # expr.py:139-142 + 323-332 / 362-371
        k = self._key(xb)
        if self.emulate_memo and k in self._cvx_cache:
            return self._cvx_cache[k]
        A = self.grad(xb)
        b = self.eval(xb) - A.dot(xb)
        b = b - self.val
        self._cvx_cache[k] = (A, b)
        return A, b

    def violation(self, xb):                              # prob.py:582-590
        v = self.eval(xb) - self.val
        return np.abs(v) if self.kind == "eq" else np.maximum(v, 0.0)


class FlatProblem(object):
    def __init__(self, x0, Q, a, b0=0.0, lin_A=None, lin_lo=None, lin_hi=None, blocks=(), prox_count=None,
                 obj_blocks=()):
        self.x0 = np.asarray(x0, dtype=np.float64).ravel()
        self.n_x = self.x0.shape[0]
        # how many Variables with a value contain each atom: find_closest_feasible_point
        # adds one (x_i - x0_i)^2 term per Variable (prob.py:381-404)
        self.prox_count = (np.ones(self.n_x) if prox_count is None
                           else np.asarray(prox_count, dtype=np.float64).ravel())
        self.Q = sp.csc_matrix(Q, dtype=np.float64)
        self.a = np.asarray(a, dtype=np.float64).ravel()
        self.b0 = float(b0)
        if lin_A is None:
            lin_A = sp.csc_matrix((0, self.n_x)); lin_lo = np.zeros(0); lin_hi = np.zeros(0)
        self.lin_A = sp.csr_matrix(lin_A, dtype=np.float64)
        self.lin_lo = np.asarray(lin_lo, dtype=np.float64).ravel()
        self.lin_hi = np.asarray(lin_hi, dtype=np.float64).ravel()
        self.blocks = list(blocks)
        self.obj_blocks = list(obj_blocks)


class SolverParams(object):
    """solver.py:17-28"""

    def __init__(self, **kw):
        self.improve_ratio_threshold = 0.25
        self.min_trust_region_size = 1e-4
        self.min_approx_improve = 1e-8
        self.trust_shrink_ratio = 0.1
        self.trust_expand_ratio = 1.5
        self.cnt_tolerance = 1e-4
        self.max_merit_coeff_increases = 1
        self.merit_coeff_increase_ratio = 1e1
        self.initial_trust_region_size = 1
        self.initial_penalty_coeff = 1e3
        self.compound_penalty = True      # Q1
        self.duplicate_rows = True        # Q2
        self.max_qp_solves = 10000        # safety cap (reference loops are unbounded, Q5)
        for k, v in kw.items():
            assert hasattr(self, k), k
            setattr(self, k, v)


class _State(object):
    pass


def _sym_triu(Q):
    """P handed to OSQP: triu of (Q + Q')/2 (osqp_utils.py:153-163 with prob.py:353-359)."""
    S = (Q + Q.T) * 0.5
    return sp.triu(S, format="csc")


def penalty_sqp(p, params=None, qp_settings=None, record_qps=False, emulate_memo=True, qp_solver=None):
    """Solver.solve(prob, method='penalty_sqp') on a FlatProblem.

    Returns a namespace: x, success, trace (one row per QP solve:
    kind, merit, model_merit, new_merit, trust, penalty, qp_status, qp_iters),
    sqp_iters, qp_solves, admm_iters and optionally the list of QPs."""
    P_ = params or SolverParams()
    qs = dict(qp_settings or {})
    solve_qp = qp_solver or (lambda P, q, A, l, u, w, kw: osqp_ref.solve(P, q, A, l, u, w=w, **kw))
    for blk in list(p.blocks) + list(p.obj_blocks):
        blk.emulate_memo = emulate_memo
    st = _State()
    st.x = p.x0.copy(); st.x_saved = None
    st.trace = []; st.qps = []; st.sqp_iters = 0; st.qp_solves = 0; st.admm_iters = 0
    n_x = p.n_x
    Ptri = _sym_triu(p.Q)
    m_lin = p.lin_A.shape[0]

    def quad_obj(x):
        # prob.py:571-573: quadratic objective expressions, then the non-quadratic ones at their true value
        v = 0.5 * x.dot(p.Q.dot(x)) + p.a.dot(x) + p.b0
        for ob in p.obj_blocks:
            v += ob.eval(x[ob.idx])
        return v

    def model_obj(x, omodels):
        # prob.py:625-626: quadratic objective expressions, then the degree-2 models of the non-quadratic ones
        v = 0.5 * x.dot(p.Q.dot(x)) + p.a.dot(x) + p.b0
        for ob, (H, A, b) in zip(p.obj_blocks, omodels):
            xb = x[ob.idx]
            v += 0.5 * xb.dot(H.dot(xb)) + A.dot(xb) + b
        return v

    def record(kind, merit, model, new, trust, pen, res):
        st.trace.append((kind, merit, model, new, trust, pen, res.info.status_val, res.info.iter))

    def run_qp(P, q, A, l, u, w, settings):
        res = solve_qp(P, q, A, l, u, w, settings)
        st.qp_solves += 1; st.admm_iters += res.info.iter
        if record_qps:
            st.qps.append(dict(P=sp.csc_matrix(P).toarray(), q=np.array(q), A=sp.csc_matrix(A).toarray(),
                               l=np.array(l), u=np.array(u), w=None if w is None else np.array(w),
                               x=res.x.copy(), status=res.info.status_val, iters=res.info.iter))
        return res

    # ---- find_closest_feasible_point (prob.py:369-412), default QP settings (Q7)
    known = ~np.isnan(st.x)
    Pp = sp.diags(np.where(known, 2.0 * p.prox_count, 0.0)).tocsc()
    qp = np.where(known, -2.0 * np.where(known, st.x, 0.0) * p.prox_count, 0.0)
    A0 = sp.vstack([p.lin_A, sp.identity(n_x, format="csr")]).tocsc()
    l0 = np.concatenate([p.lin_lo, np.full(n_x, -np.inf)])
    u0 = np.concatenate([p.lin_hi, np.full(n_x, np.inf)])
    res = run_qp(Pp, qp, A0, l0, u0, None, {})
    out = _State()
    if res.info.status_val not in (1, 2):                    # solver.py:81-82: give up, variables untouched
        record(STEP_PROJECT, 0.0, 0.0, 0.0, P_.initial_trust_region_size, P_.initial_penalty_coeff, res)
        out.x, out.success = st.x, False
        out.trace, out.qps = np.array(st.trace, dtype=np.float64).reshape(-1, 8), st.qps
        out.sqp_iters, out.qp_solves, out.admm_iters = 0, st.qp_solves, st.admm_iters
        out.max_violation = max([float(np.max(b.violation(st.x[b.idx]))) for b in p.blocks] + [0.0])
        out.merit = 0.0
        out.group_ids, out.nonconverged_groups = sorted(set(g for b in p.blocks for g in b.groups)), []
        return out
    st.x = res.x.copy()
    record(STEP_PROJECT, 0.0, 0.0, 0.0, P_.initial_trust_region_size, P_.initial_penalty_coeff, res)

    # ---- penalty-QP layout (fixed once the slacks exist, prob.py:434-458)
    n_slack = sum(b.r * (1 if b.kind == "leq" else 2) for b in p.blocks)
    n = n_x + n_slack
    slack_of = []
    off = n_x
    for b in p.blocks:
        if b.kind == "leq":
            slack_of.append((np.arange(off, off + b.r), None)); off += b.r
        else:
            # p_i, n_i side by side (the canonical column order of tests/trajopt_build.py)
            slack_of.append((off + 2 * np.arange(b.r), off + 2 * np.arange(b.r) + 1)); off += 2 * b.r
    m_nl = sum(b.r for b in p.blocks)
    Pfull = sp.block_diag([Ptri, sp.csc_matrix((n_slack, n_slack))], format="csc") if n_slack else Ptri
    state = dict(slack_cost=1.0, k=0, masks=None)

    # constraint groups (prob.py:81-86, 135-142): sorted ids, member blocks, overlap graph
    gids = sorted(set(g for b in p.blocks for g in b.groups))
    gind = {g: i for i, g in enumerate(gids)}
    members = [[k for k, b in enumerate(p.blocks) if g in b.groups] for g in gids]
    overlap = [sorted(set(gind[h] for b in p.blocks if g in b.groups for h in b.groups if h != g)) for g in gids]
    st.nonconverged = []

    def group_vec(per_block):
        return np.array([sum(per_block[k] for k in mem) for mem in members])

    def max_violation():
        worst = 0.0
        for b in p.blocks:
            worst = max(worst, float(np.max(b.violation(st.x[b.idx]))))
        return worst

    def min_merit_fn(penalty, trust):
        """solver.py:108-253; returns (success, trust)"""
        while True:
            if st.qp_solves >= P_.max_qp_solves:
                return False
            st.sqp_iters += 1
            # convexify (prob.py:522-544)
            omodels = [ob.convexify(st.x[ob.idx]) for ob in p.obj_blocks]        # degree 2 (prob.py:532-534)
            models = [b.convexify(st.x[b.idx]) for b in p.blocks]
            # update_obj (prob.py:414-426): spawn pattern on first use, refresh rows, scale costs
            if state["masks"] is None:
                state["masks"] = [(A != 0.0) for A, _ in models]
            state["k"] = state["k"] + 1 if P_.duplicate_rows else 1
            state["slack_cost"] = state["slack_cost"] * penalty if P_.compound_penalty else penalty
            rows, lo, hi = [], [], []
            for b, (A, bb), mask, (s1, s2) in zip(p.blocks, models, state["masks"], slack_of):
                Am = np.where(mask, A, 0.0)
                R = sp.lil_matrix((b.r, n))
                R[:, b.idx] = Am
                for i in range(b.r):
                    R[i, s1[i]] = -1.0
                    if s2 is not None:
                        R[i, s2[i]] = 1.0
                rows.append(R.tocsr())
                hi.append(-bb)
                lo.append(-bb if b.kind == "eq" else np.full(b.r, -np.inf))
            qx = p.a.copy()
            Pit = Pfull
            if omodels:
                # QuadExpr lowering (prob.py:348-367, osqp_utils.py:153-163): Q into P, A into q
                Hs = sp.lil_matrix((n, n))
                for ob, (H, A, b) in zip(p.obj_blocks, omodels):
                    Hs[np.ix_(ob.idx, ob.idx)] = Hs[np.ix_(ob.idx, ob.idx)].toarray() + H
                    qx[ob.idx] += A
                Pit = Pfull + sp.triu(Hs.tocsc(), format="csc")
            q = np.concatenate([qx, np.full(n_slack, state["slack_cost"])])
            lin_ext = sp.hstack([p.lin_A, sp.csr_matrix((m_lin, n_slack))]).tocsr() if n_slack else p.lin_A
            A_top = sp.vstack([lin_ext] + rows).tocsr() if rows else lin_ext
            w_top = np.concatenate([np.ones(m_lin, dtype=np.int64), np.full(m_nl, state["k"], dtype=np.int64)])
            lo_top = np.concatenate([p.lin_lo] + lo) if lo else p.lin_lo
            hi_top = np.concatenate([p.lin_hi] + hi) if hi else p.lin_hi
            # merit at the convexification point (prob.py:571-579)
            bviol = [float(np.sum(b.violation(st.x[b.idx]))) for b in p.blocks]
            viol = sum(bviol)
            merit = quad_obj(st.x) + penalty * viol
            merit_vec = group_vec(bviol)                      # get_value(vectorize=True), prob.py:558-570
            st.x_saved = st.x.copy()
            while True:
                # trust region on x, slacks in [0, inf) (variable.py:43-45, prob.py:454)
                lb = np.concatenate([st.x_saved - trust, np.zeros(n_slack)])
                ub = np.concatenate([st.x_saved + trust, np.full(n_slack, np.inf)])
                A = sp.vstack([A_top, sp.identity(n, format="csr")]).tocsc()
                l = np.concatenate([lo_top, lb]); u = np.concatenate([hi_top, ub])
                w = np.concatenate([w_top, np.ones(n, dtype=np.int64)])
                res = run_qp(Pit, q, A, l, u, w, qs)
                if res.info.status_val in (1, 2):                 # prob.py:197-203
                    st.x = res.x[:n_x].copy()
                # model merit (prob.py:605-630): full Jacobian, not the masked rows
                bmviol = []
                for b, (Am, bm) in zip(p.blocks, models):
                    v = Am.dot(st.x[b.idx]) + bm
                    bmviol.append(float(np.sum(np.abs(v) if b.kind == "eq" else np.maximum(v, 0.0))))
                mviol = sum(bmviol)
                model_merit = model_obj(st.x, omodels) + penalty * mviol
                model_vec = group_vec(bmviol)                 # get_approx_value(vectorize=True), prob.py:617-622
                nviol = sum(float(np.sum(b.violation(st.x[b.idx]))) for b in p.blocks)
                new_merit = quad_obj(st.x) + penalty * nviol

                approx = merit - model_merit
                if not approx:
                    approx += 1e-12
                exact = merit - new_merit
                ratio = exact / approx
                approx_vec, violated = merit_vec - model_vec, merit_vec > P_.cnt_tolerance
                rec = (merit, model_merit, new_merit, trust, penalty, res)
                if approx < -1e-5:                                # _bad_model
                    st.x = st.x_saved.copy(); record(STEP_BAD, *rec); return False, trust
                if approx < P_.min_approx_improve:                # _y_converged
                    st.x = st.x_saved.copy(); record(STEP_YCONV, *rec); return True, trust
                # a violated group that no longer improves, and none of whose overlapping groups
                # does either, ends the merit minimisation (solver.py:209-235)
                stalled = []
                for g in range(len(gids)):
                    if violated[g] and approx_vec[g] < P_.min_approx_improve:
                        if not any(approx_vec[h] > P_.min_approx_improve for h in overlap[g]):
                            stalled.append(g)
                if stalled:
                    st.x = st.x_saved.copy(); record(STEP_GROUP, *rec)
                    st.nonconverged = sorted(set(stalled) | set(
                        g for g in range(len(gids)) if violated[g] and approx_vec[g] < P_.min_approx_improve))
                    return True, trust
                st.nonconverged = []
                if exact < 0 or ratio < P_.improve_ratio_threshold:
                    st.x = st.x_saved.copy(); record(STEP_SHRINK, *rec)
                    trust = trust * P_.trust_shrink_ratio
                else:
                    record(STEP_ACCEPT, *rec)
                    trust = trust * P_.trust_expand_ratio
                    break
                if trust < P_.min_trust_region_size:              # _x_converged
                    st.trace[-1] = (STEP_XCONV,) + st.trace[-1][1:]
                    return True, trust
                if st.qp_solves >= P_.max_qp_solves:
                    return False, trust

    penalty = P_.initial_penalty_coeff
    trust = P_.initial_trust_region_size
    success = False
    for _ in range(P_.max_merit_coeff_increases):
        r = min_merit_fn(penalty, trust)
        ok = r[0] if isinstance(r, tuple) else r
        if p.blocks and max_violation() > P_.cnt_tolerance:
            penalty = penalty * P_.merit_coeff_increase_ratio
            trust = P_.initial_trust_region_size
        else:
            success = ok
            break
    else:
        success = False
    out.x, out.success = st.x.copy(), bool(success)
    out.trace = np.array(st.trace, dtype=np.float64).reshape(-1, 8)
    out.qps = st.qps
    out.sqp_iters, out.qp_solves, out.admm_iters = st.sqp_iters, st.qp_solves, st.admm_iters
    out.max_violation = max_violation() if p.blocks else 0.0
    out.group_ids = gids
    out.nonconverged_groups = [gids[g] for g in st.nonconverged]
    out.merit = quad_obj(st.x) + penalty * sum(float(np.sum(b.violation(st.x[b.idx]))) for b in p.blocks)
    return out


# ---------------------------------------------------------------------------
# the SURVEY 8(d) trajectory problem as a FlatProblem
# ---------------------------------------------------------------------------
def trajopt_flat(prob, analytic_jac=False):
    """FlatProblem of one oracle.arm_family.make_problem(...) instance."""
    from . import arm_family as af
    d, T = prob["d"], prob["T"]
    n_x = d * T
    # r04: per-joint weights of the smoothing objective and an acceleration term (workloads.smooth_Q: the matrix the object-API
    # builder hands to QuadExpr)
    Q = sp.lil_matrix(af.smooth_Q(d, T, prob.get("obj_w"), prob.get("acc_w")))
    reach = bool(prob.get("reach"))
    lin = sp.lil_matrix((d if reach else 2 * d, n_x))
    for j in range(d):
        lin[j, j] = 1.0
        if not reach:
            lin[d + j, (T - 1) * d + j] = 1.0
    rhs = prob["start"].copy() if reach else np.concatenate([prob["start"], prob["goal"]])
    lin_lo, lin_hi = rhs, rhs
    if prob.get("vmax") is not None:               # linear inequality rows (prob.py:329-338: lb = -inf, ub = val - b)
        V = af.velocity_rows(d, T)
        lin = sp.vstack([lin.tocsr(), sp.csr_matrix(V)]).tolil()
        lin_lo = np.concatenate([rhs, np.full(V.shape[0], -np.inf)])
        lin_hi = np.concatenate([rhs, np.full(V.shape[0], prob["vmax"])])
    if prob.get("jlo") is not None:                # joint limits: theta <= hi, then -theta <= -lo, every timestep
        lin = sp.vstack([lin.tocsr(), sp.csr_matrix(af.joint_limit_rows(d, T))]).tolil()
        lin_lo = np.concatenate([lin_lo, np.full(2 * n_x, -np.inf)])
        lin_hi = np.concatenate([lin_hi, np.tile(prob["jhi"], T), -np.tile(prob["jlo"], T)])
    if prob.get("lin_gen") is not None:            # r04: general affine rows (LEqExpr / EqExpr on an AffExpr, prob.py:317-346), behind the others
        g = prob["lin_gen"]
        lin = sp.vstack([sp.csr_matrix(lin), sp.csr_matrix(g["A"])]).tolil()
        lin_lo = np.concatenate([lin_lo, np.where(g["is_eq"] != 0, g["rhs"], -np.inf)])
        lin_hi = np.concatenate([lin_hi, g["rhs"]])
    blocks = []
    R = prob["K"] * prob["O"]
    prog = prob.get("row_program")
    span = prog.span if prog is not None else 1
    if prog is not None and (span > 1 or prog.n_eq > 0):
        # r03 program blocks: block t binds its rows to (theta[t], .., theta[t+span-1]); its inequality rows are one
        # LEqExpr, its equality rows one EqExpr (val 0) on the same Variable (tests/trajopt_build.py)
        for t in range(T - span + 1):
            idx = np.arange(t * d, (t + span) * d)
            gids = prob["groups"][t] if prob.get("groups") is not None else None
            for kind, rows in (("leq", prog.ineq_rows), ("eq", prog.eq_rows)):
                if not rows:
                    continue
                f = prog.numpy_fn(af.step_params(prob, t), rows)
                jac = prog.numpy_jac(af.step_params(prob, t), rows) if analytic_jac else None
                blocks.append(Block(kind, f, idx, np.zeros(len(rows)), jac=jac, groups=gids))
    if prob.get("quad_n_eq"):
        # quadratic rows with equality rows (r03): the block's inequality rows are one LEqExpr, its last quad_n_eq rows one
        # EqExpr (val 0) on the same timestep Variable
        ne = int(prob["quad_n_eq"])
        for t in range(T):
            idx = np.arange(t * d, (t + 1) * d)
            gids = prob["groups"][t] if prob.get("groups") is not None else None
            for kind, sl in (("leq", slice(0, R - ne)), ("eq", slice(R - ne, R))):
                f = (lambda th, pr=prob, sl=sl: af.quad_rows(th, pr["quad_Q"][sl], pr["quad_a"][sl], pr["quad_c"][sl]))
                jac = (lambda th, pr=prob, sl=sl: af.quad_rows_jac(th, pr["quad_Q"][sl], pr["quad_a"][sl], pr["quad_c"][sl])) if analytic_jac else None
                blocks.append(Block(kind, f, idx, np.zeros(len(range(R)[sl])), jac=jac, groups=gids))
    for t in range(T if not ((prog is not None and (span > 1 or prog.n_eq > 0)) or prob.get("quad_n_eq")) else 0):
        if prob.get("circle_rows"):              # r04: a second kind of rows on the timestep, added first: the point's keep-out discs
            nc = prob["circle_rows"]
            fcir = (lambda th, pr=prob, nc=nc: af.point_dist(th, pr["obstacles"][:nc]))
            jcir = (lambda th, pr=prob, nc=nc: af.point_dist_jac(th, pr["obstacles"][:nc])) if analytic_jac else None
            blocks.append(Block("leq", fcir, np.arange(t * d, (t + 1) * d), np.zeros(nc), jac=jcir,
                                groups=prob["groups"][t] if prob.get("groups") is not None else None))
        if prob.get("row_program") is not None:  # SCO_FAM_STATE_PROGRAM: closed-form rows
            f = prob["row_program"].numpy_fn(af.step_params(prob, t))
            jac = prob["row_program"].numpy_jac(af.step_params(prob, t)) if analytic_jac else None
        elif prob.get("quad_Q") is not None:    # SCO_FAM_STATE_QUADRATIC: general quadratic rows on the state
            f = (lambda th, pr=prob: af.quad_rows(th, pr["quad_Q"], pr["quad_a"], pr["quad_c"]))
            jac = (lambda th, pr=prob: af.quad_rows_jac(th, pr["quad_Q"], pr["quad_a"], pr["quad_c"])) if analytic_jac else None
        elif prob.get("point"):                 # SCO_FAM_POINT_CIRCLES: a point robot, K = 1
            f = (lambda th, pr=prob: af.point_dist(th, pr["obstacles"]))
            jac = (lambda th, pr=prob: af.point_dist_jac(th, pr["obstacles"])) if analytic_jac else None
        else:
            f = (lambda th, pr=prob: af.arm_dist(th, pr["link_len"], pr["point_link"], pr["point_frac"], pr["obstacles"]))
            jac = None
        if analytic_jac and not prob.get("point") and prob.get("quad_Q") is None and prob.get("row_program") is None:
            jac = (lambda th, pr=prob: af.arm_dist_jac(th, pr["link_len"], pr["point_link"], pr["point_frac"], pr["obstacles"]))
        blocks.append(Block("leq", f, np.arange(t * d, (t + 1) * d), np.zeros(R - int(prob.get("circle_rows") or 0)), jac=jac,
                            groups=prob["groups"][t] if prob.get("groups") is not None else None))
    if reach:
        f = (lambda th, pr=prob: af.ee_pos(th, pr["link_len"]))
        jac = (lambda th, pr=prob: af.ee_jac(th, pr["link_len"])) if analytic_jac else None
        blocks.append(Block("eq", f, np.arange((T - 1) * d, T * d), prob["target"], jac=jac,
                            groups=prob["groups"][T] if prob.get("groups") is not None else None))
    obj_blocks = []
    if prog is not None and prog.objective:      # SCO_FAM_FLAG_OBJ_PROGRAM: a non-quadratic objective term per timestep
        for t in range(T):
            obj_blocks.append(ObjBlock(prog.objective_fn(af.step_params(prob, t)), np.arange(t * d, (t + 1) * d)))
    if prob.get("cost_weight") is not None:
        for t in range(T):
            fc = (lambda th, pr=prob: af.ee_cost(th, pr["link_len"], pr["cost_target"], pr["cost_weight"]))
            obj_blocks.append(ObjBlock(fc, np.arange(t * d, (t + 1) * d)))
    # the object-API construction (tests/trajopt_build.py) binds every atom to two
    # Variables: the whole trajectory and its timestep block; with blocks of `span` timesteps an atom of timestep t sits
    # in the trajectory and in every block Variable that covers t
    nblk = np.array([min(t, T - span) - max(0, t - span + 1) + 1 for t in range(T)], dtype=np.float64)
    return FlatProblem(prob["x0"], Q.tocsc(), np.zeros(n_x), 0.0, lin.tocsr(), lin_lo, lin_hi, blocks,
                       prox_count=np.repeat(1.0 + nblk, d), obj_blocks=obj_blocks)
