"""Batched, device-resident SCO solves (``sco_sqp_*`` in include/sco_hip.h).

The reference has no batch API: a caller such as OpenTAMP builds one ``Prob`` per
candidate motion plan and calls ``Solver.solve`` on each in a Python loop
(/root/reference/sco_py/sco_osqp/solver.py:30).  ``TrajOptBatch`` runs that same
per-problem algorithm for B structurally identical trajectory problems at once,
entirely on the GPU, for the constraint families the device can evaluate
(include/sco_hip.h: SCO_FAM_*).  Arbitrary Python callables stay on the
per-problem path (``sco_py_amd.sco_osqp``), where only the QP solve is on the GPU.
"""
import ctypes as C
from types import SimpleNamespace

import numpy as np

from . import _lib

SCO_FAM_ARM_CIRCLES = 1
SCO_FAM_ARM_REACH = 2
SCO_FAM_POINT_CIRCLES = 3
SCO_FAM_STATE_QUADRATIC = 4
SCO_FAM_STATE_PROGRAM = 5
SCO_FAM_FLAG_VEL_LIMITS = 16
SCO_FAM_FLAG_JOINT_LIMITS = 32
SCO_FAM_FLAG_EE_COST = 64
SCO_FAM_FLAG_OBJ_PROGRAM = 128
SCO_FAM_FLAG_ACC_COST = 256
TRACE_W = 8


class TrajOptBatch(object):
    """B planar-arm trajectory problems:

        min  sum_t ||theta[t+1] - theta[t]||^2
        s.t. theta[0] = start, theta[T-1] = goal,
             r_o - ||p_k(theta[t]) - c_o|| <= 0   for all t, link points k, obstacles o

    (``reach=True``: the goal pin is replaced by the non-linear equality
    ee(theta[T-1]) = target, SCO_FAM_ARM_REACH; ``vel_limits=True``: linear rows
    |theta[t+1] - theta[t]| <= vmax in every QP, SCO_FAM_FLAG_VEL_LIMITS; ``joint_limits=True``:
    linear rows lo <= theta[t] <= hi, SCO_FAM_FLAG_JOINT_LIMITS; ``ee_cost=True``: a non-quadratic objective term
    weight * ||ee(theta[t]) - target||^2 per timestep, convexified to degree 2 like ``Prob.add_obj_expr`` on a plain
    ``Expr`` (numeric Hessian + eigenvalue shift), SCO_FAM_FLAG_EE_COST)
    ``point=True``: SCO_FAM_POINT_CIRCLES, a point robot in the plane (state of a timestep: dof >= 2 numbers, the first
    two its position; n_points = 1; link data are not read) instead of the arm; ``quadratic=True``:
    SCO_FAM_STATE_QUADRATIC, n_obstacles rows 1/2 x' Q_r x + a_r' x + c_r <= 0 on the state of every timestep with
    per-problem coefficients (``load(..., quad_Q=, quad_a=, quad_c=)``); ``program=True``: SCO_FAM_STATE_PROGRAM, rows
    given as closed-form expressions over the state and a per-problem parameter vector, compiled by
    ``sco_py_amd.rowexpr.compile_rows`` (``load(..., row_program=, row_params=)``); the program says how many timesteps
    a constraint block spans (``span`` 1 .. 4: block t binds its rows to (theta[t], .., theta[t+span-1])), how many of a
    block's rows are equalities (``n_eq``) and whether it carries a non-quadratic objective term per timestep -- pass it
    to the constructor as ``program=prog``; ``analytic_jac=True`` differentiates program rows in forward mode)
    solved per problem exactly like ``Solver().solve(prob, method="penalty_sqp")``.
    ``prox_count`` says how many Variables hold each atom in the equivalent object-API
    construction (it scales the projection QP of find_closest_feasible_point,
    prob.py:381-404); building the problem with one trajectory Variable plus one
    Variable per timestep, as tests/trajopt_build.py does, gives 2.
    """

    def __init__(self, batch, dof, horizon, n_points, n_obstacles, device=0, analytic_jac=False,
                 prox_count=2, reach=False, vel_limits=False, joint_limits=False, ee_cost=False, point=False,
                 quadratic=False, program=False, n_eq_rows=0, lin_rows=None, circle_rows=0, acc_cost=False):
        """acc_cost (r04): the quadratic objective carries an acceleration term sum_t sum_j a_j (x[t+2][j] - 2 x[t+1][j] + x[t][j])^2
        (``load(acc_weights=)``, (B, dof)); fixed at creation because P gets a second super-diagonal block.
        circle_rows (r04, program family, span 1): the first ``circle_rows`` of the ``n_obstacles`` rows of a block are keep-out
        rows of the point (x[0], x[1]) against ``obstacles[b, :circle_rows]`` -- a second kind of non-linear rows next to the
        program's (two BoundExprs on one timestep Variable in the reference).
        lin_rows (r04): pattern of GENERAL affine rows over the trajectory variables, shared by the batch:
        (row_ptr, col_idx, is_eq) in CSR form, column t * dof + j (``rows_pattern`` builds it from a dense 0 / 1 mask) -- the rows
        a caller of the reference adds with prob.add_cnt_expr(BoundExpr(EqExpr / LEqExpr(AffExpr(A, b), val), traj)); values and
        right-hand sides per problem go to ``load(lin_vals=, lin_rhs=)``."""
        self.B, self.d, self.T, self.K, self.O = int(batch), int(dof), int(horizon), int(n_points), int(n_obstacles)
        self.n_x = self.d * self.T
        self.device = int(device)
        self._h = C.c_void_p()
        self.reach = bool(reach)
        self.vel_limits = bool(vel_limits)
        self.joint_limits = bool(joint_limits)
        self.ee_cost = bool(ee_cost)
        self.point = bool(point)          # SCO_FAM_POINT_CIRCLES: a point robot in the plane instead of the arm
        self.quadratic = bool(quadratic)   # SCO_FAM_STATE_QUADRATIC: n_obstacles general quadratic rows per timestep
        if (self.point or self.quadratic) and (self.reach or self.ee_cost):
            raise ValueError("the point-robot and quadratic-row families have neither the reach equality nor the objective term")
        self.program = bool(program)       # SCO_FAM_STATE_PROGRAM: rows as closed-form programs (sco_py_amd.rowexpr)
        if sum((self.point, self.quadratic, self.program)) > 1 or (self.program and (self.reach or self.ee_cost)):
            raise ValueError("one family per batch; the program family has neither the reach equality nor the arm's objective term")
        # the structure of the program family's blocks comes with the compiled program (program=prog); program=True keeps
        # the r02 form: one timestep per block, inequality rows only, no objective term
        prog = program if hasattr(program, "row_ptr") else None
        self.span = prog.span if prog is not None else 1
        # equality rows: the program says so itself; the quadratic-row family takes n_eq_rows (the last rows of a timestep)
        self.n_eq = prog.n_eq if prog is not None else (int(n_eq_rows) if self.quadratic else 0)
        if n_eq_rows and not (self.quadratic or prog is not None):
            raise ValueError("equality rows inside a block exist for the quadratic-row and program families")
        if prog is not None and n_eq_rows and int(n_eq_rows) != prog.n_eq:
            raise ValueError("n_eq_rows = %d, but the compiled program has %d equality rows (the program says how many of a "
                             "block's rows are equalities; leave n_eq_rows out)" % (int(n_eq_rows), prog.n_eq))
        self.obj_program = bool(prog is not None and prog.objective)
        self.n_blocks = self.T - self.span + 1
        desc = _lib.TrajoptDesc(self.B, self.d, self.T, self.K, self.O,
                                (SCO_FAM_STATE_PROGRAM if self.program else SCO_FAM_STATE_QUADRATIC if self.quadratic else
                                 SCO_FAM_POINT_CIRCLES if self.point else
                                 SCO_FAM_ARM_REACH if self.reach else SCO_FAM_ARM_CIRCLES) |
                                (SCO_FAM_FLAG_VEL_LIMITS if self.vel_limits else 0) |
                                (SCO_FAM_FLAG_JOINT_LIMITS if self.joint_limits else 0) |
                                (SCO_FAM_FLAG_EE_COST if self.ee_cost else 0) |
                                (SCO_FAM_FLAG_OBJ_PROGRAM if self.obj_program else 0) |
                                (SCO_FAM_FLAG_ACC_COST if bool(acc_cost) else 0),
                                1 if analytic_jac else 0, int(prox_count), self.span, self.n_eq)
        self.acc_cost = bool(acc_cost)
        self.circle_rows = int(circle_rows)
        if self.circle_rows and not (self.program and self.span == 1):
            raise ValueError("circle rows next to program rows: program family, blocks on one timestep")
        self.lin_rows = None
        if lin_rows is not None and len(lin_rows[2]):
            rp = np.ascontiguousarray(lin_rows[0], dtype=np.int32); ci = np.ascontiguousarray(lin_rows[1], dtype=np.int32)
            eq = np.ascontiguousarray(lin_rows[2], dtype=np.int32)
            if rp.shape != (eq.shape[0] + 1,) or ci.shape != (int(rp[-1]),):
                raise ValueError("lin_rows = (row_ptr [n_rows + 1], col_idx [nnz], is_eq [n_rows])")
            self.lin_rows = (rp, ci, eq)
            _lib.check(_lib.load().sco_sqp_create_rows(self.device, C.byref(desc), eq.shape[0], _lib.iptr(rp), _lib.iptr(ci), _lib.iptr(eq),
                                                       C.byref(self._h)))
        else:
            _lib.check(_lib.load().sco_sqp_create(self.device, C.byref(desc), C.byref(self._h)))
        if self.circle_rows:
            _lib.check(_lib.load().sco_sqp_set_circle_rows(self._h, self.circle_rows))

    def close(self):
        if self._h:
            _lib.load().sco_sqp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def load(self, x0, start, goal, link_len, point_link, point_frac, obstacles, target=None, vmax=None,
             jlo=None, jhi=None, cost_weight=None, cost_target=None, quad_Q=None, quad_a=None, quad_c=None,
             row_program=None, row_params=None, obj_weights=None, lin_vals=None, lin_rhs=None, acc_weights=None):
        """Upload per-problem data (host arrays, copied).  ``target`` (B, 2): end-effector
        position of the reach variant (``goal`` is then ignored by the device).  r04: ``row_params`` may be (B, T, n_params) --
        one parameter vector per timestep (block t and the objective term of timestep t read row_params[b, t]);
        ``obj_weights`` (B, dof): weights w_j of the smoothing objective sum_t sum_j w_j (x[t+1][j] - x[t][j])^2."""
        B, d, K, O = self.B, self.d, self.K, self.O

        def arr(a, shape, dt=np.float64):
            a = np.ascontiguousarray(a, dtype=dt)
            if a.shape != shape:
                raise ValueError("expected shape %r, got %r" % (shape, a.shape))
            return a

        x0 = arr(x0, (B, self.n_x)); start = arr(start, (B, d)); goal = arr(goal, (B, d))
        link_len = arr(link_len, (B, d)); obstacles = arr(obstacles, (B, O, 3))
        point_link = arr(point_link, (K,), np.int32); point_frac = arr(point_frac, (K,))
        _lib.check(_lib.load().sco_sqp_load(self._h, _lib.dptr(x0), _lib.dptr(start), _lib.dptr(goal),
                                            _lib.dptr(link_len), _lib.iptr(point_link), _lib.dptr(point_frac),
                                            _lib.dptr(obstacles)))
        if self.quadratic:
            if quad_Q is None or quad_a is None or quad_c is None:
                raise ValueError("the quadratic-row family needs quad_Q (B, O, d, d), quad_a (B, O, d), quad_c (B, O)")
            qQ = arr(quad_Q, (B, O, d, d)); qa = arr(quad_a, (B, O, d)); qc = arr(quad_c, (B, O))
            _lib.check(_lib.load().sco_sqp_load_quadratic(self._h, _lib.dptr(qQ), _lib.dptr(qa), _lib.dptr(qc)))
        if self.program:
            if row_program is None:
                raise ValueError("the program family needs row_program (sco_py_amd.rowexpr.compile_rows) and row_params (B, n_params)")
            if row_program.n_rows != O - self.circle_rows or row_program.n_state > d * self.span or row_program.span != self.span or \
                    row_program.n_eq != self.n_eq or row_program.objective != self.obj_program:
                raise ValueError("the program has %d rows (%d equalities) over %d state coordinates, span %d, objective term %s; the "
                                 "batch was created for %d rows per block (%d equalities), dof %d, span %d, objective term %s"
                                 % (row_program.n_rows, row_program.n_eq, row_program.n_state, row_program.span, row_program.objective,
                                    O - self.circle_rows, self.n_eq, d, self.span, self.obj_program))
            npar = row_program.n_params
            per_step = npar > 0 and row_params is not None and np.ndim(row_params) == 3
            if per_step:
                par = arr(row_params, (B, self.T, npar))
            else:
                par = arr(row_params if row_params is not None else np.zeros((B, 0)), (B, npar)) if npar else np.zeros((B, 0))
            _lib.check((_lib.load().sco_sqp_load_program_steps if per_step else _lib.load().sco_sqp_load_program)(
                self._h, len(row_program.words), _lib.iptr(np.ascontiguousarray(row_program.words.ravel())), _lib.iptr(row_program.row_ptr),
                len(row_program.consts), _lib.dptr(row_program.consts) if len(row_program.consts) else None,
                npar, _lib.dptr(par) if npar else None))
        if self.reach:
            if target is None:
                raise ValueError("the reach variant needs target (B, 2)")
            target = arr(target, (B, 2))
            _lib.check(_lib.load().sco_sqp_load_target(self._h, _lib.dptr(target)))
        if self.vel_limits:
            if vmax is None:
                raise ValueError("velocity limits need vmax (B,)")
            vmax = arr(np.broadcast_to(np.asarray(vmax, dtype=np.float64), (B,)), (B,))
            _lib.check(_lib.load().sco_sqp_load_vel_limit(self._h, _lib.dptr(vmax)))
        if self.ee_cost:
            if cost_weight is None or cost_target is None:
                raise ValueError("the objective term needs cost_weight (B,) and cost_target (B, 2)")
            cw = arr(np.broadcast_to(np.asarray(cost_weight, dtype=np.float64), (B,)), (B,))
            ct = arr(cost_target, (B, 2))
            _lib.check(_lib.load().sco_sqp_load_ee_cost(self._h, _lib.dptr(cw), _lib.dptr(ct)))
        if self.joint_limits:
            if jlo is None or jhi is None:
                raise ValueError("joint limits need jlo, jhi (B, dof)")
            jlo = arr(np.broadcast_to(np.asarray(jlo, dtype=np.float64), (B, d)), (B, d))
            jhi = arr(np.broadcast_to(np.asarray(jhi, dtype=np.float64), (B, d)), (B, d))
            _lib.check(_lib.load().sco_sqp_load_joint_limits(self._h, _lib.dptr(jlo), _lib.dptr(jhi)))
        if self.lin_rows is not None:
            if lin_vals is None or lin_rhs is None:
                raise ValueError("general affine rows need lin_vals (B, nnz) and lin_rhs (B, n_rows)")
            lv = arr(lin_vals, (B, self.lin_rows[1].shape[0])); lr = arr(lin_rhs, (B, self.lin_rows[2].shape[0]))
            _lib.check(_lib.load().sco_sqp_load_linear_rows(self._h, _lib.dptr(lv), _lib.dptr(lr)))
        elif lin_vals is not None or lin_rhs is not None:
            raise ValueError("the batch was created without general affine rows (lin_rows=)")
        if self.acc_cost:
            aw = arr(np.broadcast_to(np.asarray(acc_weights if acc_weights is not None else 0.0, dtype=np.float64), (B, d)), (B, d))
            _lib.check(_lib.load().sco_sqp_load_acc_weights(self._h, _lib.dptr(aw)))
        elif acc_weights is not None:
            raise ValueError("the batch was created without the acceleration term (acc_cost=True)")
        if obj_weights is not None:
            ow = arr(np.broadcast_to(np.asarray(obj_weights, dtype=np.float64), (B, d)), (B, d))
            _lib.check(_lib.load().sco_sqp_load_obj_weights(self._h, _lib.dptr(ow)))
        else:
            _lib.check(_lib.load().sco_sqp_load_obj_weights(self._h, None))

    def set_groups(self, block_groups):
        """Constraint groups (``prob.add_cnt_expr(bound_expr, group_ids)``): one list of group ids
        per constraint block (T timestep blocks, then the reach block).  Ids are sorted like the
        reference sorts them; ``fetch().nonconverged_groups`` reports them by name."""
        nb = self.n_blocks + (1 if self.reach else 0)
        if len(block_groups) != nb:
            raise ValueError("expected %d blocks, got %d" % (nb, len(block_groups)))
        gids = sorted(set(g for blk in block_groups for g in blk))
        if not 1 <= len(gids) <= 32 or any(len(blk) == 0 for blk in block_groups):
            raise ValueError("1..32 group ids, every block in at least one group")
        mask = np.array([sum(1 << gids.index(g) for g in set(blk)) for blk in block_groups], dtype=np.uint32)
        _lib.check(_lib.load().sco_sqp_set_groups(self._h, len(gids), mask.ctypes.data_as(C.POINTER(C.c_uint))))
        self.group_ids = gids

    def solve(self, params=None, qp_settings=None):
        """Run the penalty SQP for every problem; blocks until all are done.
        May be called repeatedly: every call restarts from the loaded state."""
        p = params if params is not None else _lib.default_sqp_params()
        q = qp_settings if qp_settings is not None else _lib.default_qp_settings()
        _lib.check(_lib.load().sco_sqp_solve(self._h, C.byref(p), C.byref(q)))

    def fetch(self, with_merit=True):
        B = self.B
        x = np.zeros((B, self.n_x)); success = np.zeros(B, dtype=np.int32)
        sqp_iters = np.zeros(B, dtype=np.int32); qp_solves = np.zeros(B, dtype=np.int32)
        admm = np.zeros(B, dtype=np.int64)
        merit = np.zeros(B) if with_merit else None
        viol = np.zeros(B) if with_merit else None
        _lib.check(_lib.load().sco_sqp_fetch(
            self._h, _lib.dptr(x), _lib.iptr(success), _lib.iptr(sqp_iters), _lib.iptr(qp_solves),
            admm.ctypes.data_as(C.POINTER(C.c_longlong)), _lib.dptr(merit), _lib.dptr(viol)))
        nc = np.zeros(B, dtype=np.uint32)
        _lib.check(_lib.load().sco_sqp_fetch_groups(self._h, nc.ctypes.data_as(C.POINTER(C.c_uint))))
        st = np.zeros(B, dtype=np.uint32)
        _lib.check(_lib.load().sco_sqp_fetch_stalled_groups(self._h, st.ctypes.data_as(C.POINTER(C.c_uint))))
        flags = np.zeros(B, dtype=np.int32)
        _lib.check(_lib.load().sco_sqp_fetch_flags(self._h, _lib.iptr(flags)))
        gids = getattr(self, "group_ids", ["all"])
        groups = [[g for k, g in enumerate(gids) if (int(m) >> k) & 1] for m in nc]
        stalled = [[g for k, g in enumerate(gids) if (int(m) >> k) & 1] for m in st]
        return SimpleNamespace(x=x, success=success.astype(bool), sqp_iters=sqp_iters, qp_solves=qp_solves,
                               admm_iters=admm, merit=merit, max_violation=viol, nonconverged_groups=groups,
                               stalled_groups=stalled,   # the groups that ended the minimisation (listed first by the reference)
                               flags=flags)      # bit flags SCO_SQP_FLAG_* (memo history full, capped, trace full)

    def trace(self, cap=64):
        """Per-problem decision trace: list of (n_rows, 8) arrays with columns
        kind, merit, model_merit, new_merit, trust, penalty, qp_status, qp_iters."""
        tr = np.zeros((self.B, cap, TRACE_W)); n = np.zeros(self.B, dtype=np.int32)
        _lib.check(_lib.load().sco_sqp_trace(self._h, cap, _lib.dptr(tr), _lib.iptr(n)))
        return [tr[b, : min(int(n[b]), cap)].copy() for b in range(self.B)]

    def last_timing(self):
        ms = np.zeros(5)
        _lib.check(_lib.load().sco_sqp_last_timing(self._h, _lib.dptr(ms)))
        rounds = C.c_int(0)
        _lib.check(_lib.load().sco_sqp_last_rounds(self._h, C.byref(rounds)))
        launches, groups = C.c_int(0), C.c_int(0)
        _lib.check(_lib.load().sco_sqp_last_launches(self._h, C.byref(launches), C.byref(groups)))
        tms = np.zeros(2); tit = np.zeros(2, dtype=np.int64); tl = np.zeros(2, dtype=np.int32)
        _lib.check(_lib.load().sco_sqp_last_tiers(self._h, _lib.dptr(tms), tit.ctypes.data_as(C.POINTER(C.c_longlong)), _lib.iptr(tl)))
        return dict(convexify_ms=float(ms[0]), qp_setup_ms=float(ms[1]), admm_ms=float(ms[2]),
                    decide_ms=float(ms[3]), total_ms=float(ms[4]), rounds=int(rounds.value),
                    launches=int(launches.value), groups=int(groups.value),
                    # ADMM launches by kernel tier: wavefront tier (rounds with >= ~3 live problems per CU) / the rest
                    wv_ms=float(tms[0]), wv_iters=int(tit[0]), wv_launches=int(tl[0]), other_admm_ms=float(tms[1]),
                    other_launches=int(tl[1]))


def rows_pattern(mask):
    """CSR pattern (row_ptr, col_idx) of a dense (n_rows, dof * horizon) 0 / 1 mask, and the gather index of its entries."""
    mask = np.asarray(mask) != 0
    row_ptr = np.concatenate([[0], np.cumsum(mask.sum(axis=1))]).astype(np.int32)
    r, c = np.nonzero(mask)                      # row-major: CSR order
    return row_ptr, c.astype(np.int32), (r, c)


def solve_batch(batch_arrays, params=None, qp_settings=None, device=0, analytic_jac=False, prox_count=2):
    """One-shot helper: dict with keys d, T, K, O, B, x0, start, goal, link_len,
    point_link, point_frac, obstacles (as produced by the synthetic workload
    generator) -> result namespace of :meth:`TrajOptBatch.fetch`."""
    a = batch_arrays
    with TrajOptBatch(a["B"], a["d"], a["T"], a["K"], a["O"], device=device, analytic_jac=analytic_jac,
                      prox_count=prox_count, reach=bool(a.get("reach")), vel_limits=a.get("vmax") is not None,
                      joint_limits=a.get("jlo") is not None, ee_cost=a.get("cost_weight") is not None,
                      point=bool(a.get("point")), quadratic=a.get("quad_Q") is not None,
                      program=a.get("row_program") if a.get("row_program") is not None else False,
                      n_eq_rows=a.get("quad_n_eq", 0), lin_rows=a.get("lin_rows"), circle_rows=a.get("circle_rows", 0),
                      acc_cost=a.get("acc_w") is not None) as tb:
        tb.load(a["x0"], a["start"], a["goal"], a["link_len"], a["point_link"], a["point_frac"], a["obstacles"],
                target=a.get("target"), vmax=a.get("vmax"), jlo=a.get("jlo"), jhi=a.get("jhi"),
                cost_weight=a.get("cost_weight"), cost_target=a.get("cost_target"),
                quad_Q=a.get("quad_Q"), quad_a=a.get("quad_a"), quad_c=a.get("quad_c"),
                row_program=a.get("row_program"), row_params=a.get("row_params"), obj_weights=a.get("obj_w"),
                lin_vals=a.get("lin_vals"), lin_rhs=a.get("lin_rhs"), acc_weights=a.get("acc_w"))
        if a.get("groups") is not None:
            tb.set_groups(a["groups"])
        tb.solve(params, qp_settings)
        res = tb.fetch()
        res.trace = tb.trace()
        res.timing = tb.last_timing()
    return res
