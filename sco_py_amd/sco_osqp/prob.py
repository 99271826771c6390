"""Sequential-convex problem container and its lowering to a QP.

Mirror of ``sco_py.sco_osqp.prob`` (/root/reference/sco_py/sco_osqp/prob.py):
same public and semi-public surface (tests and OpenTAMP reach into
``_osqp_lin_cnt_exprs``, ``_penalty_exprs``, ``_cnt_groups`` ...), same
behaviour including the reference's load-bearing quirks, which SURVEY.md 2.4
lists as Q1 (penalty costs compound), Q2 (penalty rows re-appended), Q14
(affine objective terms scaled by the penalty coefficient).  The per-nonzero
Python loops of the reference's coefficient refresh (prob.py:488-504) are
replaced by index maps built once when the slack variables are spawned.
"""
from collections import defaultdict

import numpy as np

from .. import expr as sco_osqp_expr
from . import osqp_utils
from .osqp_utils import OSQPLinearConstraint, OSQPLinearObj, OSQPQuadraticObj, OSQPVar


def _noop():
    pass


class _RowMap(object):
    """Where the coefficients of one penalty block live inside its constraint
    rows: for row i, ``cols[i]`` are the columns of the affine model that were
    non-zero when the row was created and ``pos[i]`` their positions inside the
    row's coefficient array.  Columns outside this pattern are never written
    (prob.py:493, 502: the ``in cnts[i].osqp_vars`` tests)."""

    def __init__(self, A):
        self.cols = []
        for i in range(A.shape[0]):
            (nz,) = np.nonzero(A[i, :])
            self.cols.append(nz)
        # the x-part of a row is written first, in np.nonzero order (prob.py:273, 308)
        self.pos = [np.arange(c.shape[0]) for c in self.cols]


class Prob(object):
    """SCO problem with a scalar objective, solved by the l1 penalty method
    (prob.py:14-86)."""

    def __init__(self, callback=None):
        self._vars = set()
        self._osqp_vars = set()
        self._callback = callback if callback is not None else _noop

        self._quad_obj_exprs = []
        self._nonquad_obj_exprs = []
        self._approx_obj_exprs = []
        self._nonlin_cnt_exprs = []

        # atoms of the QP that is handed to the solver
        self._osqp_quad_objs = []
        self._osqp_lin_objs = []
        self._osqp_lin_cnt_exprs = []

        self.hinge_created = False
        self._penalty_exprs = []
        self._osqp_penalty_cnts = []
        self._osqp_penalty_exprs = []
        self._row_maps = []

        # constraint groups (prob.py:81-86)
        self._cnt_groups = defaultdict(set)
        self._cnt_groups_overlap = defaultdict(set)
        self._penalty_groups = []
        self.nonconverged_groups = []
        self.gid2ind = {}

    # ------------------------------------------------------------------ build
    def add_obj_expr(self, bound_expr):
        """Affine/quadratic objectives go straight to the QP, anything else is
        convexified every iteration (prob.py:88-104)."""
        e = bound_expr.expr
        if isinstance(e, (sco_osqp_expr.AffExpr, sco_osqp_expr.QuadExpr)):
            self._quad_obj_exprs.append(bound_expr)
        else:
            self._nonquad_obj_exprs.append(bound_expr)
        self.add_var(bound_expr.var)

    def add_var(self, var):
        self._vars.add(var)

    def add_osqp_var(self, osqp_var):
        self._osqp_vars.add(osqp_var)

    def add_cnt_expr(self, bound_expr, group_ids=None):
        """Affine Eq/LEq constraints become rows immediately; everything else is
        a non-linear constraint handled by the penalty method (prob.py:112-144).
        An affine LExpr is silently ignored, as in the reference (Q18)."""
        comp = bound_expr.expr
        inner = comp.expr
        var = bound_expr.var
        assert isinstance(comp, sco_osqp_expr.CompExpr)
        if isinstance(inner, sco_osqp_expr.AffExpr):
            if isinstance(comp, sco_osqp_expr.EqExpr):
                self._add_osqp_cnt_from_aff_expr(inner, var, "eq", comp.val)
            elif isinstance(comp, sco_osqp_expr.LEqExpr):
                self._add_osqp_cnt_from_aff_expr(inner, var, "leq", comp.val)
        else:
            self._nonlin_cnt_exprs.append(bound_expr)
            self._reset_hinge_cnts()
            gids = ["all"] if group_ids is None else group_ids
            for gid in gids:
                self._cnt_groups[gid].add(bound_expr)
                for other in gids:
                    if other != gid:
                        self._cnt_groups_overlap[gid].add(other)
        self.add_var(var)

    def batch_add_lin_cnts(self, list_of_lin_cnts):
        self._osqp_lin_cnt_exprs.extend(list_of_lin_cnts)

    # --------------------------------------------------------------- QP solve
    def optimize(self,
                 add_convexified_terms=False,
                 osqp_eps_abs=osqp_utils.DEFAULT_EPS_ABS,
                 osqp_eps_rel=osqp_utils.DEFAULT_EPS_REL,
                 osqp_max_iter=osqp_utils.DEFAULT_MAX_ITER,
                 rho: float = osqp_utils.DEFAULT_RHO,
                 adaptive_rho: bool = osqp_utils.DEFAULT_ADAPTIVE_RHO,
                 sigma: float = osqp_utils.DEFAULT_SIGMA,
                 verbose=False):
        """Solve the current QP; on success scatter the solution into the
        variables and fire the callback (prob.py:146-205)."""
        lin_objs = self._osqp_lin_objs
        cnts = self._osqp_lin_cnt_exprs
        if add_convexified_terms:
            lin_objs = self._osqp_lin_objs + self._osqp_penalty_exprs
            cnts = self._osqp_lin_cnt_exprs[:]
            for block in self._osqp_penalty_cnts:
                cnts.extend(block)
        solve_res, var_to_index_dict = osqp_utils.optimize(
            self._osqp_vars, self._vars, self._osqp_quad_objs, lin_objs, cnts,
            osqp_eps_abs, osqp_eps_rel, osqp_max_iter,
            rho=rho, adaptive_rho=adaptive_rho, sigma=sigma, verbose=verbose,
        )
        if solve_res.info.status_val not in [1, 2]:      # prob.py:197
            return False
        osqp_utils.update_osqp_vars(var_to_index_dict, solve_res.x)
        self._update_vars()
        self._callback()
        return True

    def _reset_hinge_cnts(self):
        self.hinge_created = False

    # ----------------------------------------------------- lowering of exprs
    def _add_osqp_objs_and_cnts_from_expr(self, bound_expr):
        """Dispatch on the expression class (prob.py:211-238)."""
        e, var = bound_expr.expr, bound_expr.var
        if isinstance(e, sco_osqp_expr.AffExpr):
            self._add_to_lin_objs_and_cnts_from_aff_expr(e, var)
        elif isinstance(e, sco_osqp_expr.QuadExpr):
            self._add_to_quad_and_lin_objs_from_quad_expr(e, var)
        elif isinstance(e, sco_osqp_expr.HingeExpr):
            self._add_to_lin_objs_and_cnts_from_hinge_expr(e, var)
        elif isinstance(e, sco_osqp_expr.AbsExpr):
            self._add_to_lin_objs_and_cnts_from_abs_expr(e, var)
        elif isinstance(e, sco_osqp_expr.CompExpr):
            raise Exception(
                "Comparison Expressions cannot be converted to OSQP problem objectives; "
                "use _add_osqp_cnt_from_aff_expr instead"
            )
        else:
            raise Exception("This type of Expression cannot be converted to an OSQP objective.")

    def _add_to_lin_objs_and_cnts_from_aff_expr(self, aff_expr, var):
        """Affine OBJECTIVE term: one linear cost per non-zero coefficient, filed
        with the penalty terms so update_obj scales it by the penalty
        coefficient (prob.py:240-249; Q14)."""
        atoms = var.get_osqp_vars()
        A = aff_expr.A
        for i in range(A.shape[0]):
            (nz,) = np.nonzero(A[i, :])
            for coeff, atom in zip(A[i, nz].tolist(), atoms[nz, 0].tolist()):
                self._osqp_penalty_exprs.append(OSQPLinearObj(atom, coeff))

    def _add_to_lin_objs_and_cnts_from_hinge_expr(self, hinge_expr, var):
        """max(a_i x + b_i, 0): slack t_i >= 0 with unit cost and the row
        -inf <= a_i x - t_i <= -b_i (prob.py:251-278)."""
        aff = hinge_expr.expr
        assert isinstance(aff, sco_osqp_expr.AffExpr)
        atoms = var.get_osqp_vars()
        A, b = aff.A, aff.b
        slack = self.create_pos_osqp_var_arr((A.shape[0], 1))
        for _, t in np.ndenumerate(slack):
            self._osqp_penalty_exprs.append(OSQPLinearObj(t, 1.0))
        rows = []
        for i in range(A.shape[0]):
            (nz,) = np.nonzero(A[i, :])
            rows.append(OSQPLinearConstraint(
                np.concatenate((atoms[nz, 0], slack[i])),
                np.concatenate((A[i, nz], np.array([-1.0]))),
                -np.inf, -b[i]))
        self._osqp_penalty_cnts.append(rows)

    def _add_to_lin_objs_and_cnts_from_abs_expr(self, abs_expr, var):
        """|a_i x + b_i|: slacks p_i, n_i >= 0 with unit cost and the row
        a_i x - p_i + n_i = -b_i (prob.py:280-315)."""
        aff = abs_expr.expr
        assert isinstance(aff, sco_osqp_expr.AffExpr)
        A, b = aff.A, aff.b
        pos = self.create_pos_osqp_var_arr((A.shape[0], 1))
        neg = self.create_pos_osqp_var_arr((A.shape[0], 1))
        for t in pos.flat:
            self._osqp_penalty_exprs.append(OSQPLinearObj(t, 1.0))
        for t in neg.flat:
            self._osqp_penalty_exprs.append(OSQPLinearObj(t, 1.0))
        atoms = var.get_osqp_vars()
        rows = []
        for i in range(A.shape[0]):
            (nz,) = np.nonzero(A[i, :])
            rows.append(OSQPLinearConstraint(
                np.concatenate((atoms[nz, 0], pos[i], neg[i])),
                np.concatenate((A[i, nz], np.array([-1.0]), np.array([1.0]))),
                -b[i], -b[i]))
        self._osqp_penalty_cnts.append(rows)

    def _add_osqp_cnt_from_aff_expr(self, aff_expr, var, cnt_type, cnt_val):
        """Affine constraint rows: eq -> lb = ub = val - b, leq -> (-inf, val - b]
        (prob.py:317-346)."""
        atoms = var.get_osqp_vars()
        A, b = aff_expr.A, aff_expr.b
        if cnt_type not in ("eq", "leq"):
            raise NotImplementedError
        for i in range(A.shape[0]):
            (nz,) = np.nonzero(A[i, :])
            hi = cnt_val[i] - b[i]
            lo = hi if cnt_type == "eq" else -np.inf
            self._osqp_lin_cnt_exprs.append(OSQPLinearConstraint(atoms[nz, 0], A[i, nz], lo, hi))

    def _add_to_quad_and_lin_objs_from_quad_expr(self, quad_expr, var):
        """0.5 x'Qx + a x: Q is passed un-halved because the solver minimises
        0.5 x'Px (prob.py:348-367); the constant b is dropped (Q13)."""
        atoms = var.get_osqp_vars()
        assert atoms.shape[1] == 1
        Q = quad_expr.Q
        ri, ci = np.nonzero(Q)
        self._osqp_quad_objs.append(OSQPQuadraticObj(atoms[ri, 0], atoms[ci, 0], Q[ri, ci]))
        _, lin_cols = np.nonzero(quad_expr.A)
        lin_vals = quad_expr.A[0, lin_cols] if quad_expr.A.shape[0] == 1 else quad_expr.A[np.nonzero(quad_expr.A)]
        lin_atoms = atoms[lin_cols, 0]
        assert lin_vals.shape == lin_atoms.shape
        for atom, coeff in zip(lin_atoms.tolist(), lin_vals.tolist()):
            self._osqp_lin_objs.append(OSQPLinearObj(atom, coeff))

    def find_closest_feasible_point(self,
                                    osqp_eps_abs=osqp_utils.DEFAULT_EPS_ABS,
                                    osqp_eps_rel=osqp_utils.DEFAULT_EPS_REL,
                                    osqp_max_iter=osqp_utils.DEFAULT_MAX_ITER,
                                    rho: float = osqp_utils.DEFAULT_RHO,
                                    adaptive_rho: bool = osqp_utils.DEFAULT_ADAPTIVE_RHO,
                                    sigma: float = osqp_utils.DEFAULT_SIGMA,
                                    ):
        """Project the initial guess onto the linear constraints in the l2 norm:
        adds (x_i - x0_i)^2 for every non-NaN initial entry (prob.py:369-412)."""
        for var in self._vars:
            val = var.get_value()
            if val is None:
                continue
            atoms = var.get_osqp_vars()
            assert atoms.shape == val.shape
            known = np.where(~np.isnan(val))
            for atom, x0 in zip(atoms[known].flatten().tolist(), val[known].flatten()):
                self._osqp_lin_objs.append(OSQPLinearObj(atom, -2.0 * x0))
                one = np.array([atom])
                self._osqp_quad_objs.append(OSQPQuadraticObj(one, one, np.array([2.0])))
        return self.optimize(osqp_eps_abs=osqp_eps_abs, osqp_eps_rel=osqp_eps_rel,
                             osqp_max_iter=osqp_max_iter, rho=rho,
                             adaptive_rho=adaptive_rho, sigma=sigma)

    # ------------------------------------------------- per-iteration lowering
    def update_obj(self, penalty_coeff=0.0):
        """Rebuild the QP objective for the current convexification and refresh /
        re-append the penalty rows (prob.py:414-426)."""
        self._reset_osqp_objs()
        self._lazy_spawn_osqp_cnts()
        for bexpr in self._quad_obj_exprs + self._approx_obj_exprs:
            self._add_osqp_objs_and_cnts_from_expr(bexpr)
        for k, bexpr in enumerate(self._penalty_exprs):
            self._update_nonlin_cnt_and_add_to_qp(bexpr, k)
        # Q1/Q14: the SAME objects are scaled in place on every call, so their
        # coefficients compound across iterations (prob.py:424-426)
        for term in self._osqp_penalty_exprs:
            term.coeff = term.coeff * penalty_coeff
            self._osqp_lin_objs.append(term)

    def _reset_osqp_objs(self):
        self._osqp_quad_objs = []
        self._osqp_lin_objs = []

    def _lazy_spawn_osqp_cnts(self):
        """First update_obj after the constraint set changed: create slack
        variables, their costs and the penalty rows (prob.py:434-444)."""
        if self.hinge_created:
            return
        self._osqp_penalty_cnts = []
        self._osqp_penalty_exprs = []
        self._osqp_nz = []
        self._row_maps = []
        for bexpr in self._penalty_exprs:
            A = bexpr.expr.expr.A
            self._osqp_nz.append(np.nonzero(A))
            self._row_maps.append(_RowMap(A))
            self._add_osqp_objs_and_cnts_from_expr(bexpr)
        self.hinge_created = True

    def create_pos_osqp_var_arr(self, shape):
        """Fresh slack atoms in [0, inf), named so that they sort after every
        user variable (prob.py:446-458)."""
        arr = np.empty(shape, dtype=object)
        for pos in np.ndindex(*shape):
            t = OSQPVar("z+_pos_osqp_var", 0.0, np.inf, 0.0)
            self._osqp_vars.add(t)
            arr[pos] = t
        return arr
    def _update_nonlin_cnt_and_add_to_qp(self, bexpr, ind):
        """Write the new affine model (A, b) of penalty block ``ind`` into its
        existing rows, then append those rows to the QP once more
        (prob.py:461-512; the append is Q2)."""
        e = bexpr.expr
        if not isinstance(e, (sco_osqp_expr.HingeExpr, sco_osqp_expr.AbsExpr)):
            raise NotImplementedError
        aff = e.expr
        assert isinstance(aff, sco_osqp_expr.AffExpr)
        A, b = aff.A, aff.b
        rows = self._osqp_penalty_cnts[ind]
        rmap = self._row_maps[ind]
        for i in range(A.shape[0]):
            row = rows[i]
            if row.lb == row.ub:           # equality (abs) row: both sides move
                row.lb = -b[i, 0]
                row.ub = -b[i, 0]
            else:                          # hinge row: only the upper side
                row.ub = -b[i, 0]
            # columns of the creation-time pattern take the new coefficient (zero
            # included); all other columns of A are ignored
            row.coeffs[rmap.pos[i]] = A[i, rmap.cols[i]]
        self._osqp_nz[ind] = np.nonzero(A)
        self._osqp_lin_cnt_exprs.extend(rows)

    def add_trust_region(self, trust_region_size):
        for var in self._vars:
            var.add_trust_region(trust_region_size)
    def convexify(self):
        """Quadratic models of the non-quadratic objectives and l1-penalty models
        of the non-linear constraints at the current point, plus the per-group
        lists used for group-wise convergence (prob.py:522-544)."""
        self._approx_obj_exprs = [be.convexify(degree=2) for be in self._nonquad_obj_exprs]
        self._penalty_exprs = [be.convexify(degree=1) for be in self._nonlin_cnt_exprs]
        self._penalty_groups = []
        self.gid2ind = {}
        for k, gid in enumerate(sorted(self._cnt_groups.keys())):
            self.gid2ind[gid] = k
            self._penalty_groups.append([be.convexify(degree=1) for be in self._cnt_groups[gid]])

    # ------------------------------------------------------------------ merit
    def get_value(self, penalty_coeff, vectorize=False):
        """Exact penalty objective at the current point, or (vectorize) the
        per-group sums of constraint violation (prob.py:547-579)."""
        if vectorize:
            gids = sorted(self._cnt_groups.keys())
            out = np.zeros(len(gids))
            for k, gid in enumerate(gids):
                out[k] = np.sum(np.sum(
                    [np.sum(self._compute_cnt_violation(be)) for be in self._cnt_groups[gid]]))
            return out
        total = 0.0
        for be in self._quad_obj_exprs + self._nonquad_obj_exprs:
            total += np.sum(np.sum(be.eval()))
        for be in self._nonlin_cnt_exprs:
            total += penalty_coeff * np.sum(self._compute_cnt_violation(be))
        return total
    def _compute_cnt_violation(self, bexpr):
        comp = bexpr.expr
        at = bexpr.var.get_value()
        if isinstance(comp, sco_osqp_expr.EqExpr):
            return np.absolute(comp.expr.eval(at) - comp.val)
        elif isinstance(comp, sco_osqp_expr.LEqExpr):
            v = comp.expr.eval(at) - comp.val
            return np.maximum(v, np.zeros(v.shape))
        # other comparison kinds fall through and yield None, as in the reference

    def get_max_cnt_violation(self):
        """Largest violation over all non-linear constraint rows (prob.py:592-603)."""
        worst = 0.0
        for be in self._nonlin_cnt_exprs:
            worst = np.maximum(worst, np.amax(self._compute_cnt_violation(be)))
        return worst

    def get_approx_value(self, penalty_coeff, vectorize=False):
        """Value of the convex model built at the last convexify call, evaluated
        at the current point (prob.py:605-630)."""
        if vectorize:
            out = np.zeros(len(self._penalty_groups))
            for k, bexprs in enumerate(self._penalty_groups):
                out[k] = np.sum(np.array([np.sum(be.eval()) for be in bexprs]).flatten())
            return out
        total = 0.0
        for be in self._quad_obj_exprs + self._approx_obj_exprs:
            total += np.sum(np.sum(be.eval()))
        for be in self._penalty_exprs:
            total += penalty_coeff * np.sum(be.eval())
        return total

    # ------------------------------------------------------------------ state
    def _update_vars(self):
        for var in self._vars:
            var.update()

    def save(self):
        for var in self._vars:
            var.save()

    def restore(self):
        for var in self._vars:
            var.restore()
