"""Compile a ``Prob`` of the object API onto the device-resident penalty-SQP loop.

The reference's entry point is ``Solver.solve(prob)`` on a ``Prob`` built from ``BoundExpr`` objects
(/root/reference/sco_py/sco_osqp/solver.py:30-59, 107-253; prob.py:88-144; expr.py:413-437): Python drives every SQP
iteration.  ``compile_prob`` reads such a ``Prob`` -- Variables, objective expressions, affine rows, non-linear
``BoundExpr`` blocks, constraint groups -- and, when everything non-linear in it is a ``devexpr.DeviceExpr`` and the
rest matches the device template (include/sco_hip.h: sco_trajopt_desc), turns it into the flat per-problem record the
``sco_sqp_*`` layer takes.  ``Solver.solve`` then runs the whole solve on the GPU and writes the result back into the
``Variable`` objects; ``solve_many`` buckets many compiled ``Prob`` objects by structure into one batch each.  A
``Prob`` that is not recognised (arbitrary callables, another objective, rows outside the template, a callback, a
``Solver`` subclass with its own predicates) keeps the host loop with one device QP per ``optimize`` -- ``compile_prob``
returns ``None`` and says why in ``last_reason()``.

What is read and what it must look like (each test names the reference behaviour it protects):

* trajectory atoms: every ``OSQPVar`` of the ``Prob``, distinct names; QP columns are the name-sorted atoms
  (osqp_utils.py:136-143), so timestep t, coordinate j is sorted position t * d + j;
* ``Variable`` objects: all hold values, agree on them, and every atom is held by the same number of Variables apart
  from the constraint blocks covering it -- the projection QP adds one (x_i - x0_i)^2 per holder (prob.py:381-404);
* objective: one ``QuadExpr`` sum_t |x_{t+1} - x_t|^2 on the whole trajectory (prob.py:348-367), optionally one
  non-quadratic ``DeviceExpr`` objective term per timestep (prob.py:88-104);
* affine rows in the order they were added (``_osqp_lin_cnt_exprs``, prob.py:317-346): start pin, goal pin, velocity
  limits, joint limits; r04: whatever follows them is taken as general affine rows (equalities and upper bounds over the
  trajectory atoms) -- a shared sparsity pattern per device batch, coefficients and right-hand sides per problem;
* non-linear blocks in timestep order (``_nonlin_cnt_exprs``, prob.py:132-142): ``LEqExpr`` (val 0) and ``EqExpr``
  bodies of one device family, same structure at every timestep, per-problem parameters.
"""
import numpy as np

from .. import devexpr as dx
from .. import workloads as wl
from .. import expr as ex

_reason = [""]


def last_reason():
    """Why the last ``compile_prob`` call returned None ("" if it did not)."""
    return _reason[0]


class _No(Exception):
    pass


def _no(why):
    raise _No(why)


class CompiledProb(object):
    """``key``: hashable structure (problems with equal keys share a device batch); ``pr``: per-problem record in the
    layout of ``workloads.make_problem``; ``holders``: [(Variable, index array into x)] for the write-back."""

    def __init__(self, key, pr, holders, atoms, group_ids):
        self.key, self.pr, self.holders, self.atoms, self.group_ids = key, pr, holders, atoms, group_ids


def _same(a, b):
    if a is b:                                  # the usual case: one parameter array shared by the expressions of every timestep
        return True
    a, b = np.asarray(a), np.asarray(b)
    return a.shape == b.shape and bool(np.all((a == b) | (np.isnan(a) & np.isnan(b))))


def compile_prob(prob):
    """``CompiledProb`` or ``None`` (see the module docstring)."""
    try:
        out = _compile(prob)
        _reason[0] = ""
        return out
    except _No as why:
        _reason[0] = str(why)
        return None


def _compile(prob):
    from . import prob as prob_mod
    if prob._callback is not prob_mod._noop:
        _no("the Prob has a callback (prob.py:204 calls it after every QP; the resident loop has no host round trip)")
    if prob.hinge_created or prob._osqp_quad_objs or prob._osqp_lin_objs:
        _no("the Prob has already been lowered or solved on the host")
    nl = list(prob._nonlin_cnt_exprs)
    if not nl:
        _no("no non-linear constraint")
    atoms = sorted(prob._osqp_vars, key=lambda a: a.var_name)
    names = [a.var_name for a in atoms]
    if len(set(names)) != len(names):
        _no("atoms with equal names (column order would be set-iteration order, SURVEY Q10)")
    n_x = len(atoms)
    col = {id(a): i for i, a in enumerate(atoms)}
    for a in atoms:
        if a.get_lower_bound() != -np.inf or a.get_upper_bound() != np.inf:
            _no("an atom carries its own bounds (the projection QP would see them)")

    def cols_of(var):
        try:
            return np.array([col[id(a)] for a in var.get_osqp_vars().ravel()], dtype=np.int64)
        except KeyError:
            _no("a Variable holds an atom the Prob does not know")

    # ---- non-linear blocks -------------------------------------------------------------------------------------
    first = nl[0].expr.expr
    if not isinstance(first, dx.DeviceExpr) or first.role != "rows":
        _no("non-linear constraint bodies are not device expressions")
    analytic = bool(first.analytic)
    span = first.program.span if first.kind == "program" else 1
    c0 = cols_of(nl[0].var)
    if c0.shape[0] % span or c0[0] != 0:
        _no("the first block does not start at timestep 0")
    d = c0.shape[0] // span
    if n_x % d:
        _no("atom count is not a multiple of the state dimension")
    T = n_x // d
    n_blocks = T - span + 1
    blocks = [[] for _ in range(n_blocks)]        # per block: the BoundExprs on it, in order
    reach_be = None
    last_t = 0
    for be in nl:
        comp, e = be.expr, be.expr.expr
        if not isinstance(e, dx.DeviceExpr) or e.role != "rows" or bool(e.analytic) != analytic:
            _no("mixed device / host expressions or mixed Jacobian modes")
        c = cols_of(be.var)
        if e.kind == "arm_reach":
            if reach_be is not None or be is not nl[-1] or not isinstance(comp, ex.EqExpr) or \
                    not np.array_equal(c, np.arange((T - 1) * d, T * d)):
                _no("the end-effector equality must be the last constraint, on the last timestep")
            reach_be = be
            continue
        if c.shape[0] != span * d or c[0] % d or not np.array_equal(c, np.arange(c[0], c[0] + span * d)):
            _no("a block's Variable is not %d consecutive timesteps" % span)
        t = int(c[0]) // d
        if t < last_t or t >= n_blocks:
            _no("blocks are not in timestep order")
        last_t = t
        if np.any(comp.val != 0.0) or comp.val.shape != (e.n_rows(), 1):
            _no("device rows compare with val = 0")
        blocks[t].append(be)

    fam = first.kind
    blocks_all = [list(bl) for bl in blocks]
    circ = None
    if fam == "point_circles" and all(len(bl) >= 2 and bl[1].expr.expr.kind == "program" for bl in blocks):
        # r04: two kinds of non-linear rows on every timestep Variable -- the point's keep-out discs first, then program rows
        # (sco_sqp_set_circle_rows)
        circ = [bl[0] for bl in blocks]
        if any(type(be.expr) is not ex.LEqExpr or be.expr.expr.kind != "point_circles" or
               not _same(be.expr.expr.obstacles, circ[0].expr.expr.obstacles) for be in circ):
            _no("the circle rows in front of the program rows are one LEqExpr per timestep with per-problem obstacles")
        blocks = [bl[1:] for bl in blocks]
        fam = "program"
        if blocks[0][0].expr.expr.program.span != 1:
            _no("circle rows go with program rows on single timesteps")
    pr = dict(d=d, T=T, reach=reach_be is not None, link_len=np.ones(d), point_link=np.zeros(1, dtype=np.int32),
              point_frac=np.ones(1))
    n_eq = 0
    key_fam = None
    for t, bl in enumerate(blocks):
        kinds = [(type(be.expr), be.expr.expr.kind) for be in bl]
        if fam in ("arm_circles", "point_circles"):
            if kinds != [(ex.LEqExpr, fam)]:
                _no("every timestep needs exactly one LEqExpr block of the %s family" % fam)
        elif kinds not in ([(ex.LEqExpr, fam)], [(ex.LEqExpr, fam), (ex.EqExpr, fam)], [(ex.EqExpr, fam)]):
            _no("a block is one LEqExpr and / or one EqExpr of the %s family, inequalities first" % fam)
    e0 = blocks[0][0].expr.expr
    if fam == "arm_circles":
        if reach_be is not None and not _same(reach_be.expr.expr.link_len, e0.link_len):
            _no("arm parameters differ between expressions")
        for bl in blocks:
            e = bl[0].expr.expr
            if not (_same(e.link_len, e0.link_len) and _same(e.point_link, e0.point_link) and
                    _same(e.point_frac, e0.point_frac) and _same(e.obstacles, e0.obstacles)):
                _no("per-timestep arm / obstacle parameters (the device template has per-problem ones)")
        if e0.link_len.shape[0] != d:
            _no("link count differs from the state dimension")
        pr.update(K=int(e0.point_link.shape[0]), O=int(e0.obstacles.shape[0]), link_len=e0.link_len.copy(),
                  point_link=e0.point_link.copy(), point_frac=e0.point_frac.copy(), obstacles=e0.obstacles.copy())
        if reach_be is not None:
            pr["target"] = np.array(reach_be.expr.val, dtype=np.float64).ravel()
            if pr["target"].shape != (2,):
                _no("the reach target has two coordinates")
        key_fam = ("arm", pr["K"], pr["O"], tuple(e0.point_link.tolist()), tuple(e0.point_frac.tolist()))
    elif reach_be is not None:
        _no("the end-effector equality belongs to the arm family")
    elif fam == "point_circles":
        for bl in blocks:
            if not _same(bl[0].expr.expr.obstacles, e0.obstacles):
                _no("per-timestep obstacles")
        pr.update(K=1, O=int(e0.obstacles.shape[0]), obstacles=e0.obstacles.copy(), point=True)
        key_fam = ("point", pr["O"])
    elif fam == "quad_rows":
        parts0 = [be.expr.expr for be in blocks[0]]
        for bl in blocks:
            if len(bl) != len(parts0) or any(not (_same(a.Q, b.Q) and _same(a.a, b.a) and _same(a.c, b.c))
                                             for a, b in zip((be.expr.expr for be in bl), parts0)):
                _no("per-timestep quadratic coefficients")
        if any(p.Q.shape[1] != d for p in parts0):
            _no("quadratic rows are not on one timestep's state")
        n_eq = parts0[-1].n_rows() if isinstance(blocks[0][-1].expr, ex.EqExpr) else 0
        Q = np.concatenate([p.Q for p in parts0]); a = np.concatenate([p.a for p in parts0]); c = np.concatenate([p.c for p in parts0])
        pr.update(K=1, O=int(c.shape[0]), obstacles=np.zeros((c.shape[0], 3)), quad_Q=Q, quad_a=a, quad_c=c)
        if n_eq:
            pr["quad_n_eq"] = n_eq
        key_fam = ("quad", pr["O"], n_eq)
    elif fam == "program":
        prog = e0.program
        want = []
        if prog.ineq_rows:
            want.append((ex.LEqExpr, prog.ineq_rows))
        if prog.eq_rows:
            want.append((ex.EqExpr, prog.eq_rows))
        step_par = []                      # the parameter vector of every block (r04: they may differ per timestep)
        for bl in blocks:
            if len(bl) != len(want):
                _no("a block does not hold the program's inequality and equality rows")
            for be, (cls, rows) in zip(bl, want):
                e = be.expr.expr
                if e.program is not prog or type(be.expr) is not cls or e.rows != list(rows) or \
                        np.shape(e.params) != np.shape(e0.params) or not _same(e.params, bl[0].expr.expr.params):
                    _no("blocks differ in program or rows, or the rows of one block in their parameters")
            step_par.append(np.asarray(bl[0].expr.expr.params, dtype=np.float64))
        n_eq = prog.n_eq
        per_step = any(not _same(q, step_par[0]) for q in step_par)
        if per_step:                       # (T, n_params): block t reads row t; rows beyond the last block belong to objective terms
            par = np.stack(step_par + [step_par[-1]] * (T - len(step_par)))
        else:
            par = step_par[0].copy()
        pr.update(K=1, O=prog.n_rows, obstacles=np.zeros((prog.n_rows, 3)), row_program=prog, row_params=par)
        nc = 0
        if circ is not None:
            cobs = np.asarray(circ[0].expr.expr.obstacles, dtype=np.float64)
            nc = int(cobs.shape[0])
            pr.update(O=nc + prog.n_rows, obstacles=np.concatenate([cobs, np.zeros((prog.n_rows, 3))]), circle_rows=nc)
        key_fam = ("program", id(prog), per_step, nc)
    else:
        _no("no device family for %r" % fam)

    # ---- holders and the projection count ---------------------------------------------------------------------
    holders = []
    count = np.zeros(n_x, dtype=np.int64)
    x0 = np.full(n_x, np.nan)
    seen = np.zeros(n_x, dtype=bool)
    for var in prob._vars:
        val = var.get_value()
        if val is None:
            _no("a Variable without a value")
        c = cols_of(var)
        v = np.asarray(val, dtype=np.float64).ravel()
        if v.shape != c.shape:
            _no("value / atom shape mismatch")
        if not _same(np.where(seen[c], x0[c], v), v):
            _no("Variables disagree on the value of an atom")
        x0[c] = v; seen[c] = True
        np.add.at(count, c, 1)
        holders.append((var, c))
    if not np.all(seen):
        _no("an atom is held by no Variable")
    cover = np.zeros(T, dtype=np.int64)
    for t in range(n_blocks):
        cover[t:t + span] += 1
    base = count - np.repeat(cover, d)
    if np.any(base != base[0]) or base[0] < 0:
        _no("atoms are held by different numbers of Variables (projection weights, prob.py:381-404)")
    pr["x0"] = x0
    prox_count = int(base[0]) + 1

    # ---- objective ---------------------------------------------------------------------------------------------
    if len(prob._quad_obj_exprs) != 1:
        _no("the template has one quadratic objective")
    qb = prob._quad_obj_exprs[0]
    qe = qb.expr
    if not isinstance(qe, ex.QuadExpr) or not np.array_equal(cols_of(qb.var), np.arange(n_x)):
        _no("the quadratic objective is not a QuadExpr on the whole trajectory")
    # sum_t sum_j w_j (x[t+1][j] - x[t][j])^2 with weights w_j >= 0 read off the first super-diagonal (r04; all 1 = the
    # reference examples' smoothing term)
    # plus (r04) an acceleration term sum_t sum_j a_j (x[t+2][j] - 2 x[t+1][j] + x[t][j])^2: a_j off the second super-diagonal
    # block, then w_j off the first one (entry ((0, j), (1, j)) is -2 w_j - 4 a_j)
    Qm = np.asarray(qe.Q, dtype=np.float64)
    ok_shape = Qm.shape == (n_x, n_x)
    aw = 0.5 * Qm[np.arange(d), 2 * d + np.arange(d)] if (ok_shape and T > 2) else np.zeros(d)
    ow = -0.5 * (Qm[np.arange(d), d + np.arange(d)] + 4.0 * aw) if (ok_shape and T > 1) else np.ones(d)
    has_acc = bool(np.any(aw != 0.0))
    if not (ok_shape and np.all(ow >= 0) and np.all(aw >= 0) and not np.any(qe.A) and not np.any(qe.b) and
            np.allclose(Qm, wl.smooth_Q(d, T, ow, aw) if has_acc else _smooth_Q(d, T, ow), rtol=1e-12, atol=1e-14)):
        _no("the quadratic objective is not sum_t sum_j w_j (x[t+1][j] - x[t][j])^2 (+ a_j (second differences)^2)")
    if not np.all(ow == 1.0):
        pr["obj_w"] = ow
    if has_acc:
        pr["acc_w"] = aw
    nq = list(prob._nonquad_obj_exprs)
    if nq:
        if len(nq) != T:
            _no("objective terms: one per timestep")
        o0 = nq[0].expr
        for t, be in enumerate(nq):
            e = be.expr
            if not isinstance(e, dx.DeviceExpr) or e.role != "objective" or e.kind != o0.kind or \
                    not np.array_equal(cols_of(be.var), np.arange(t * d, (t + 1) * d)):
                _no("objective terms are not device expressions on consecutive timesteps")
        if o0.kind == "arm_ee_cost" and fam == "arm_circles":
            if any(not (_same(be.expr.link_len, pr["link_len"]) and _same(be.expr.target, o0.target) and
                        be.expr.weight == o0.weight) for be in nq):
                _no("per-timestep objective parameters")
            pr["cost_weight"] = o0.weight; pr["cost_target"] = o0.target.copy()
        elif o0.kind == "program_obj" and fam == "program" and span == 1:
            rp = pr["row_params"]
            if any(be.expr.program is not pr["row_program"] or not _same(be.expr.params, rp[t] if np.ndim(rp) == 2 else rp)
                   for t, be in enumerate(nq)):
                _no("the objective program is not the constraint rows' program (or its parameters are not its timestep's)")
        else:
            _no("this objective term does not go with the %s family" % fam)
    elif fam == "program" and pr["row_program"].objective:
        _no("the program carries an objective term the Prob does not use")

    # ---- affine rows ---------------------------------------------------------------------------------------------
    rows = list(prob._osqp_lin_cnt_exprs)
    pos = [0]

    def take(n, pattern):
        """The next n rows as (lb, ub) arrays if row k has atoms/coefficients pattern(k), else None."""
        if pos[0] + n > len(rows):
            return None
        lb, ub = np.zeros(n), np.zeros(n)
        for k in range(n):
            r = rows[pos[0] + k]
            cs, vs = pattern(k)
            ra = r.osqp_vars.ravel()
            if len(ra) != len(cs) or any(col.get(id(a)) != c for a, c in zip(ra, cs)) or \
                    not np.array_equal(np.asarray(r.coeffs, dtype=np.float64).ravel(), vs):
                return None
            lb[k], ub[k] = float(np.ravel(r.lb)[0]), float(np.ravel(r.ub)[0])
        pos[0] += n
        return lb, ub

    one = np.array([1.0])
    got = take(d, lambda k: ([k], one))
    if got is None or not np.array_equal(got[0], got[1]):
        _no("the first affine rows are not the start pin")
    pr["start"] = got[1].copy()
    if reach_be is None:
        got = take(d, lambda k: ([(T - 1) * d + k], one))
        if got is None or not np.array_equal(got[0], got[1]):
            _no("no goal pin after the start pin")
        pr["goal"] = got[1].copy()
    else:
        pr["goal"] = pr["start"].copy()
    nv = d * (T - 1)
    save = pos[0]
    up = take(nv, lambda k: ([k, k + d], np.array([-1.0, 1.0])))
    dn = take(nv, lambda k: ([k, k + d], np.array([1.0, -1.0]))) if up is not None else None
    if dn is not None:
        ub = np.concatenate([up[1], dn[1]])
        if np.any(np.concatenate([up[0], dn[0]]) != -np.inf) or np.any(ub != ub[0]) or not ub[0] > 0:
            _no("velocity rows are not |x[t+1] - x[t]| <= vmax with one positive vmax")
        pr["vmax"] = float(ub[0])
    else:
        pos[0] = save
    save = pos[0]
    hi = take(n_x, lambda k: ([k], one))
    lo = take(n_x, lambda k: ([k], -one)) if hi is not None else None
    if lo is not None:
        jhi, jlo = hi[1].reshape(T, d), -lo[1].reshape(T, d)
        if np.any(hi[0] != -np.inf) or np.any(lo[0] != -np.inf) or np.any(jhi != jhi[0]) or np.any(jlo != jlo[0]) or \
                not np.all(jlo[0] < jhi[0]):
            _no("joint-limit rows are not lo <= x[t] <= hi with per-coordinate limits")
        pr["jlo"], pr["jhi"] = jlo[0].copy(), jhi[0].copy()
    else:
        pos[0] = save
    lin_key = None
    if pos[0] != len(rows):
        # r04: whatever follows is taken as GENERAL affine rows (sco_sqp_create_rows): equalities (lb = ub) and inequalities
        # (lb = -inf) over the trajectory atoms, pattern = the atoms each row names (prob.py:317-346)
        rest = rows[pos[0]:]
        A = np.zeros((len(rest), n_x)); rhs = np.zeros(len(rest)); is_eq = np.zeros(len(rest), dtype=np.int32)
        for k, r in enumerate(rest):
            ra = r.osqp_vars.ravel()
            cs = [col.get(id(a)) for a in ra]
            if any(c is None for c in cs) or len(set(cs)) != len(cs):
                _no("an affine row names an atom outside the trajectory (or one twice)")
            lb_, ub_ = float(np.ravel(r.lb)[0]), float(np.ravel(r.ub)[0])
            if not np.isfinite(ub_) or not (lb_ == ub_ or lb_ == -np.inf):
                _no("affine rows are equalities (lb = ub) or upper bounds (lb = -inf)")
            A[k, cs] = np.asarray(r.coeffs, dtype=np.float64).ravel()
            rhs[k] = ub_; is_eq[k] = 1 if lb_ == ub_ else 0
        if np.any(np.sum(A != 0, axis=1) == 0):
            _no("an affine row without coefficients")
        pr["lin_gen"] = dict(A=A, rhs=rhs, is_eq=is_eq)
        lin_key = ((A != 0).tobytes(), is_eq.tobytes())          # problems share a batch when the PATTERN agrees

    # ---- groups ---------------------------------------------------------------------------------------------------
    groups = prob._cnt_groups
    block_bes = [bl for bl in blocks_all] + ([[reach_be]] if reach_be is not None else [])
    gids = sorted(groups.keys())
    block_groups = []
    for bl in block_bes:
        g = [[gid for gid in gids if be in groups[gid]] for be in bl]
        if any(x != g[0] for x in g) or not g[0]:
            _no("the expressions of one block belong to different constraint groups")
        block_groups.append(g[0])
    trivial = gids == ["all"]
    if not trivial:
        if len(gids) > 32:
            _no("more than 32 constraint groups")
        pr["groups"] = block_groups
    gkey = None if trivial else tuple(tuple(g) for g in block_groups)

    key = (d, T, span, n_eq, key_fam, analytic, prox_count, reach_be is not None, "vmax" in pr, "jlo" in pr,
           "cost_weight" in pr, bool(nq), gkey, lin_key, "acc_w" in pr)
    return CompiledProb(key, pr, holders, atoms, gids)


_Q_CACHE = {}


def _smooth_Q(d, T, w=None):
    """Q of sum_t sum_j w_j (x[t+1][j] - x[t][j])^2 in the 0.5 x'Qx form of QuadExpr, entry by entry in the order a caller's
    loop over (t, j) adds them (tests/trajopt_build.py); w None = all 1 (cached)."""
    if w is not None and not np.all(np.asarray(w) == 1.0):
        n = d * T
        Q = np.zeros((n, n))
        w2 = 2.0 * np.tile(np.asarray(w, dtype=np.float64), T - 1)
        i = np.arange(n - d)
        Q[i, i] += w2; Q[i + d, i + d] += w2; Q[i, i + d] -= w2; Q[i + d, i] -= w2
        return Q
    if (d, T) not in _Q_CACHE:
        n = d * T
        Q = np.zeros((n, n))
        i = np.arange(n - d)
        Q[i, i] += 2.0; Q[i + d, i + d] += 2.0; Q[i, i + d] -= 2.0; Q[i + d, i] -= 2.0
        _Q_CACHE[(d, T)] = Q
    return _Q_CACHE[(d, T)]


# ------------------------------------------------------------------------------------------------------------------
# running compiled problems
# ------------------------------------------------------------------------------------------------------------------
_HANDLES = {}        # (key, B, device) -> TrajOptBatch; a handle costs tens of ms to create, a caller solves many batches
_HANDLE_CAP = 4


def _stack(cps):
    """Batch arrays (layout of ``workloads.make_batch``) of compiled problems sharing one key."""
    p0 = cps[0].pr
    B = len(cps)
    st = lambda k: np.stack([np.asarray(c.pr[k], dtype=np.float64) for c in cps])
    a = dict(d=p0["d"], T=p0["T"], K=p0["K"], O=p0["O"], B=B, x0=st("x0"), start=st("start"), goal=st("goal"),
             link_len=st("link_len"), point_link=p0["point_link"], point_frac=p0["point_frac"], obstacles=st("obstacles"))
    if p0.get("reach"):
        a["reach"] = True; a["target"] = st("target")
    if p0.get("point"):
        a["point"] = True
    if p0.get("row_program") is not None:
        a["row_program"] = p0["row_program"]; a["row_params"] = st("row_params")
    if p0.get("quad_Q") is not None:
        a["quad_Q"], a["quad_a"], a["quad_c"] = st("quad_Q"), st("quad_a"), st("quad_c")
        a["quad_n_eq"] = p0.get("quad_n_eq", 0)
    for k in ("vmax", "cost_weight"):
        if p0.get(k) is not None:
            a[k] = np.array([c.pr[k] for c in cps], dtype=np.float64)
    for k in ("cost_target", "jlo", "jhi"):
        if p0.get(k) is not None:
            a[k] = st(k)
    if p0.get("groups") is not None:
        a["groups"] = p0["groups"]
    if p0.get("acc_w") is not None:
        a["acc_w"] = st("acc_w")
    if p0.get("circle_rows"):
        a["circle_rows"] = p0["circle_rows"]
    if p0.get("lin_gen") is not None:
        from ..batch import rows_pattern
        row_ptr, col_idx, (r, c) = rows_pattern(p0["lin_gen"]["A"] != 0)
        a["lin_rows"] = (row_ptr, col_idx, p0["lin_gen"]["is_eq"])
        a["lin_vals"] = np.stack([cp.pr["lin_gen"]["A"][r, c] for cp in cps]); a["lin_rhs"] = np.stack([cp.pr["lin_gen"]["rhs"] for cp in cps])
    if any(c.pr.get("obj_w") is not None for c in cps):          # weights are values: weighted and plain problems share a batch
        a["obj_w"] = np.stack([np.asarray(c.pr["obj_w"], dtype=np.float64) if c.pr.get("obj_w") is not None else np.ones(p0["d"])
                               for c in cps])
    return a


def run_compiled(cps, params, qp_settings, device=0):
    """Solve compiled problems of ONE key as one device batch; returns the ``TrajOptBatch.fetch`` namespace with
    ``trace``, ``timing`` and ``stalled_groups`` added."""
    from .. import batch as sb
    key = cps[0].key
    assert all(c.key == key for c in cps)
    a = _stack(cps)
    hk = (key, len(cps), device)
    tb = _HANDLES.pop(hk, None)
    if tb is None:
        tb = sb.TrajOptBatch(a["B"], a["d"], a["T"], a["K"], a["O"], device=device, analytic_jac=bool(key[5]),
                             prox_count=int(key[6]), reach=bool(a.get("reach")), vel_limits=a.get("vmax") is not None,
                             joint_limits=a.get("jlo") is not None, ee_cost=a.get("cost_weight") is not None,
                             point=bool(a.get("point")), quadratic=a.get("quad_Q") is not None,
                             program=a.get("row_program") if a.get("row_program") is not None else False,
                             n_eq_rows=a.get("quad_n_eq", 0), lin_rows=a.get("lin_rows"), circle_rows=a.get("circle_rows", 0),
                             acc_cost=a.get("acc_w") is not None)
    _HANDLES[hk] = tb                       # most recently used last
    while len(_HANDLES) > _HANDLE_CAP:
        _HANDLES.pop(next(iter(_HANDLES))).close()
    import time
    t0 = time.perf_counter()
    tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
            target=a.get("target"), vmax=a.get("vmax"), jlo=a.get("jlo"), jhi=a.get("jhi"),
            cost_weight=a.get("cost_weight"), cost_target=a.get("cost_target"),
            quad_Q=a.get("quad_Q"), quad_a=a.get("quad_a"), quad_c=a.get("quad_c"),
            row_program=a.get("row_program"), row_params=a.get("row_params"), obj_weights=a.get("obj_w"),
            lin_vals=a.get("lin_vals"), lin_rhs=a.get("lin_rhs"), acc_weights=a.get("acc_w"))
    if a.get("groups") is not None:
        tb.set_groups(a["groups"])
    t1 = time.perf_counter()
    tb.solve(params, qp_settings)
    res = tb.fetch()
    t2 = time.perf_counter()
    res.trace = tb.trace()
    res.timing = tb.last_timing()
    # wall clock of the three host-visible parts: upload, solve + fetch (what bench.py times as a step of the array
    # API), decision traces
    res.timing.update(load_s=t1 - t0, solve_fetch_s=t2 - t1, trace_s=time.perf_counter() - t2)
    return res


def release_handles():
    while _HANDLES:
        _HANDLES.popitem()[1].close()


def write_back(prob, cp, res, b):
    """Leave ``prob`` as ``Solver.solve`` leaves it: the returned point in every Variable (value and saved value: all of
    the reference's exits restore or have just saved, solver.py:197-251), the atoms' last values, and
    ``prob.nonconverged_groups`` (solver.py:209, 232-234).  A failed projection QP changes nothing (solver.py:81-82)."""
    tr = res.trace[b]
    if tr.shape[0] >= 1 and int(tr[0, 0]) == 0 and int(tr[0, 6]) not in (1, 2):
        return False
    x = res.x[b]
    for var, c in cp.holders:
        shape = var.get_osqp_vars().shape
        var._value = x[c].reshape(shape).copy()
        var._saved_value = var._value.copy()
    for a, v in zip(cp.atoms, x):
        a.val = float(v)
    prob.nonconverged_groups = list(res.stalled_groups[b]) + list(res.nonconverged_groups[b])
    return bool(res.success[b])
