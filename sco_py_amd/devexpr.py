"""Device-evaluable expressions: ``Expr`` subclasses the GPU can evaluate and differentiate itself.

The reference's ``Expr(f, grad)`` wraps an arbitrary Python callable (/root/reference/sco_py/expr.py:22-41); a GPU
cannot call Python, so a ``Prob`` built from plain ``Expr`` objects keeps Python in the SQP loop (one device QP per
``Prob.optimize``).  The classes below ARE ``Expr`` objects -- ``f`` / ``grad`` are ordinary NumPy callables, so every
reference semantic (eval memo, numeric or analytic Jacobian, ``convexify``, use inside ``EqExpr`` / ``LEqExpr`` /
``BoundExpr``, the host SQP loop) works on them unchanged -- and they additionally carry the parameters of one of the
device constraint families of include/sco_hip.h (SCO_FAM_*).  ``sco_osqp.compile.compile_prob`` recognises a ``Prob``
whose non-linear expressions are all of these classes and hands it to the device-resident penalty-SQP loop
(``sco_sqp_*``), which is what ``Solver.solve(prob)`` and ``solve_many(probs)`` then run.

``host_evals`` counts calls of the Python ``f`` / ``grad`` of an instance: the GPU parity tests assert it stays zero
when the resident loop ran.
"""
import numpy as np

from . import workloads as wl
from .expr import Expr


class DeviceExpr(Expr):
    """Marker base: an ``Expr`` with a device twin.  ``kind`` names the family, ``role`` is "rows" (constraint body)
    or "objective" (scalar term for ``Prob.add_obj_expr``)."""
    kind = None
    role = "rows"

    def __init__(self, f, grad=None):
        self.host_evals = 0

        def counted_f(x):
            self.host_evals += 1
            return f(x)
        counted_grad = None
        if grad is not None:
            def counted_grad(x):
                self.host_evals += 1
                return grad(x)
        super(DeviceExpr, self).__init__(counted_f, counted_grad)
        self.analytic = grad is not None

    def n_rows(self):
        raise NotImplementedError


def _f64(a):
    return np.array(a, dtype=np.float64)


class ArmCirclesExpr(DeviceExpr):
    """SCO_FAM_ARM_CIRCLES: g[k * O + o](theta) = r_o - || p_k(theta) - c_o || for the K link points of a planar
    serial arm and O circular obstacles (cx, cy, radius); theta = the joint angles of ONE timestep (d, 1)."""
    kind = "arm_circles"

    def __init__(self, link_len, point_link, point_frac, obstacles, analytic=False):
        self.link_len = _f64(link_len).ravel()
        self.point_link = np.array(point_link, dtype=np.int32).ravel()
        self.point_frac = _f64(point_frac).ravel()
        self.obstacles = _f64(obstacles).reshape(-1, 3)
        f = lambda x: wl.arm_dist(x.ravel(), self.link_len, self.point_link, self.point_frac, self.obstacles).reshape(-1, 1)
        g = (lambda x: wl.arm_dist_jac(x.ravel(), self.link_len, self.point_link, self.point_frac, self.obstacles)) if analytic else None
        super(ArmCirclesExpr, self).__init__(f, g)

    def n_rows(self):
        return self.point_link.shape[0] * self.obstacles.shape[0]


class ArmReachExpr(DeviceExpr):
    """SCO_FAM_ARM_REACH: end-effector position (2 rows) of the planar arm; use inside ``EqExpr(expr, target)`` on the
    last timestep."""
    kind = "arm_reach"

    def __init__(self, link_len, analytic=False):
        self.link_len = _f64(link_len).ravel()
        f = lambda x: wl.ee_pos(x.ravel(), self.link_len).reshape(-1, 1)
        g = (lambda x: wl.ee_jac(x.ravel(), self.link_len)) if analytic else None
        super(ArmReachExpr, self).__init__(f, g)

    def n_rows(self):
        return 2


class ArmEECostExpr(DeviceExpr):
    """SCO_FAM_FLAG_EE_COST: the non-quadratic objective term weight * || ee(theta) - target ||^2 of one timestep
    (``Prob.add_obj_expr`` convexifies it to degree 2 with numeric derivatives, prob.py:88-104, expr.py:143-153)."""
    kind = "arm_ee_cost"
    role = "objective"

    def __init__(self, link_len, target, weight):
        self.link_len = _f64(link_len).ravel()
        self.target = _f64(target).ravel()
        self.weight = float(weight)
        super(ArmEECostExpr, self).__init__(
            lambda x: np.array([[wl.ee_cost(x.ravel(), self.link_len, self.target, self.weight)]]))


class PointCirclesExpr(DeviceExpr):
    """SCO_FAM_POINT_CIRCLES: g[o](x) = r_o - || x[0:2] - c_o || for a point robot (state (d, 1), d >= 2)."""
    kind = "point_circles"

    def __init__(self, obstacles, analytic=False):
        self.obstacles = _f64(obstacles).reshape(-1, 3)
        f = lambda x: wl.point_dist(x.ravel(), self.obstacles).reshape(-1, 1)
        g = (lambda x: wl.point_dist_jac(x.ravel(), self.obstacles)) if analytic else None
        super(PointCirclesExpr, self).__init__(f, g)

    def n_rows(self):
        return self.obstacles.shape[0]


class QuadRowsExpr(DeviceExpr):
    """SCO_FAM_STATE_QUADRATIC: g[r](x) = 1/2 x' Q_r x + a_r' x + c_r (Q_r symmetric, either sign)."""
    kind = "quad_rows"

    def __init__(self, Q, a, c, analytic=False):
        self.Q, self.a, self.c = _f64(Q), _f64(a), _f64(c).ravel()
        assert self.Q.ndim == 3 and self.Q.shape[1] == self.Q.shape[2] == self.a.shape[1] and self.Q.shape[0] == self.a.shape[0] == self.c.shape[0]
        f = lambda x: wl.quad_rows(x.ravel(), self.Q, self.a, self.c).reshape(-1, 1)
        g = (lambda x: wl.quad_rows_jac(x.ravel(), self.Q, self.a, self.c)) if analytic else None
        super(QuadRowsExpr, self).__init__(f, g)

    def n_rows(self):
        return self.c.shape[0]


class ProgramExpr(DeviceExpr):
    """SCO_FAM_STATE_PROGRAM: rows ``rows`` (default: all constraint rows) of a ``rowexpr.Program`` with the parameter
    vector ``params``; the state is the concatenation of the program's ``span`` timesteps."""
    kind = "program"

    def __init__(self, program, params=(), rows=None, analytic=False):
        self.program = program
        self.params = _f64(params).ravel()
        self.rows = list(range(program.n_rows)) if rows is None else [int(r) for r in rows]
        f = lambda x: program.evaluate(x.ravel(), self.params, self.rows).reshape(-1, 1)
        g = (lambda x: program.jacobian(x.ravel(), self.params, self.rows)) if analytic else None
        super(ProgramExpr, self).__init__(f, g)

    def n_rows(self):
        return len(self.rows)


class ProgramObjExpr(DeviceExpr):
    """SCO_FAM_FLAG_OBJ_PROGRAM: the objective program of a ``rowexpr.Program`` (``compile_rows(..., objective=)``) as
    the non-quadratic objective term of one timestep."""
    kind = "program_obj"
    role = "objective"

    def __init__(self, program, params=()):
        assert program.objective
        self.program = program
        self.params = _f64(params).ravel()
        super(ProgramObjExpr, self).__init__(
            lambda x: np.array([[program.evaluate(x.ravel(), self.params, rows=[program.n_rows])[0]]]))
