"""Expression algebra of the SCO front-end (host side).

Drop-in mirror of the reference module ``sco_py.expr``
(/root/reference/sco_py/expr.py).  Class names, constructor signatures, method
names, return conventions and error behaviour follow the reference so that
caller code (OpenTAMP, the reference's tests) runs unchanged; the file:line
citations below say which reference behaviour each piece preserves.  Numeric
differentiation is done by :mod:`sco_py_amd.numdiff` (the reference uses the
third-party numdifftools, expr.py:67, 108).

Values are 2-D ``numpy`` arrays; a variable value is an (n, 1) column.
"""
import numpy as np
from scipy.linalg import eigvalsh

from . import numdiff

DEFAULT_TOL = 1e-4   # expr.py:5
N_DIGS = 6           # expr.py:13 -- memo keys are x rounded to 6 decimals


def _memo_key(x):
    """Cache key of a point: its entries rounded to N_DIGS (expr.py:31-32)."""
    return tuple(np.round(x, N_DIGS).ravel())


class Expr(object):
    """Black-box expression ``f`` with optional analytic ``grad`` / ``hess``
    (expr.py:16-156)."""

    def __init__(self, f, grad=None, hess=None, **kwargs):
        self.f = f
        self._grad = grad
        self._hess = hess
        self._eval_cache = {}
        self._grad_cache = {}
        self._convexify_cache = {}

    def _get_key(self, x):
        return _memo_key(x)

    # -- evaluation ---------------------------------------------------------
    def eval(self, x):
        """f(x), memoised on the rounded point.  The first call hands back the
        fresh array and stores a copy; later calls return the stored array
        (expr.py:34-41)."""
        k = self._get_key(x)
        hit = self._eval_cache.get(k)
        if hit is not None:
            return hit
        fresh = self.f(x)
        self._eval_cache[k] = fresh.copy()
        return fresh

    def _get_flat_f(self, x):
        """Callable on a flat vector for the differentiator (expr.py:43-59)."""
        nd = np.ndim(x)
        if nd == 1:
            return self.f
        if nd == 2:
            shape = np.shape(x)
            assert shape[1] == 1
            return lambda v: np.ravel(self.f(np.reshape(v, shape)))
        raise Exception("Input shape not supported")

    def _num_grad(self, x):
        """Finite-difference Jacobian, (r, n) (expr.py:61-69)."""
        return numdiff.jacobian(self._get_flat_f(x), np.ravel(x))

    def _debug_grad(self, g1, g2, atol=DEFAULT_TOL):
        bad = np.argwhere(~np.isclose(np.asarray(g1), np.asarray(g2), atol=atol))
        for i, j in bad:
            print("{}, {}".format(i, j))
            print(g1[i, j], g2[i, j])

    def grad(self, x, num_check=False, atol=DEFAULT_TOL):
        """Jacobian at x.  Analytic results are cached per rounded point, numeric
        ones are not (expr.py:78-100)."""
        k = self._get_key(x)
        if k in self._grad_cache:
            return self._grad_cache[k].copy()
        assert not num_check or self._grad is not None
        if self._grad is None:
            return self._num_grad(x)
        g = self._grad(x)
        if num_check:
            g_fd = self._num_grad(x)
            if not np.allclose(g_fd, g, atol=atol):
                self._debug_grad(g, g_fd, atol=atol)
                raise Exception(
                    "Numerical and analytical gradients aren't close. "
                    "\nnum_grad: {0}\nana_grad: {1}\n".format(g_fd, g)
                )
        self._grad_cache[k] = g.copy()
        return g

    def _num_hess(self, x):
        """Finite-difference Hessian of a scalar expression (expr.py:102-109)."""
        return numdiff.hessian(self._get_flat_f(x), np.ravel(x))

    def hess(self, x, num_check=False, atol=DEFAULT_TOL):
        """Hessian at x (expr.py:111-128)."""
        assert not num_check or self._hess is not None
        if self._hess is None:
            return self._num_hess(x)
        h = self._hess(x)
        if num_check:
            h_fd = self._num_hess(x)
            if not np.allclose(h_fd, h, atol=atol):
                raise Exception(
                    "Numerical and analytical hessians aren't close. "
                    "\nnum_hess: {0}\nana_hess: {1}\n".format(h_fd, h)
                )
        return h

    # -- convexification ----------------------------------------------------
    def convexify(self, x, degree=1):
        """Model of self around x: degree 1 -> AffExpr(J, f - J x); degree 2 ->
        QuadExpr with the Hessian shifted by its most negative eigenvalue so it
        is PSD (expr.py:130-156)."""
        if degree == 1:
            J = self.grad(x)
            return AffExpr(J, self.eval(x) - J.dot(x))
        if degree == 2:
            H = self.hess(x)
            lam_min = min(eigvalsh(H))
            if lam_min < 0:
                H = H - lam_min * np.eye(H.shape[0])
            g = self.grad(x)
            xtH = np.transpose(x).dot(H)
            lin = g - xtH
            const = 0.5 * xtH.dot(x) - g.dot(x) + self.eval(x)
            return QuadExpr(H, lin, const)
        raise NotImplementedError


class AffExpr(Expr):
    """A x + b (expr.py:159-181)."""

    def __init__(self, A, b):
        assert b.shape[0] == A.shape[0]
        self.A = A
        self.b = b
        self.x_shape = (A.shape[1], 1)

    def eval(self, x):
        return self.A.dot(x) + self.b

    def grad(self, x):
        # NB the reference returns the TRANSPOSED Jacobian here (expr.py:177-178)
        return self.A.T

    def hess(self, x):
        n = self.x_shape[0]
        return np.zeros((n, n))


class QuadExpr(Expr):
    """0.5 x'Qx + A x + b, scalar valued (expr.py:184-213)."""

    def __init__(self, Q, A, b):
        assert A.shape[0] == 1, "Can only define scalar quadrative expressions"
        assert Q.shape[0] == Q.shape[1]
        assert Q.shape[0] == A.shape[1]
        assert b.shape[0] == 1
        self.Q = Q
        self.A = A
        self.b = b
        self.x_shape = (A.shape[1], 1)

    def eval(self, x):
        return 0.5 * x.T.dot(self.Q.dot(x)) + self.A.dot(x) + self.b

    def grad(self, x):
        # (n, 1) column, unlike Expr.grad (expr.py:208-210)
        assert x.shape == self.x_shape
        return 0.5 * (self.Q.dot(x) + self.Q.T.dot(x)) + self.A.T

    def hess(self, x):
        return self.Q.copy()


class AbsExpr(Expr):
    """|expr| (expr.py:216-235); non-smooth, no derivatives."""

    def __init__(self, expr):
        self.expr = expr

    def eval(self, x):
        return np.absolute(self.expr.eval(x))

    def grad(self, x):
        raise NotImplementedError

    def hess(self, x):
        raise NotImplementedError


class HingeExpr(Expr):
    """max(expr, 0) (expr.py:238-259); non-smooth, no derivatives."""

    def __init__(self, expr):
        self.expr = expr

    def eval(self, x):
        v = self.expr.eval(x)
        return np.maximum(v, np.zeros(v.shape))

    def grad(self, x):
        raise NotImplementedError

    def hess(self, x):
        raise NotImplementedError


class CompExpr(Expr):
    """expr compared with a constant ``val`` (expr.py:262-296)."""

    def __init__(self, expr, val):
        self.expr = expr
        self.val = val.copy()
        self._convexify_cache = {}

    def eval(self, x, tol=DEFAULT_TOL):
        raise NotImplementedError

    def grad(self, x):
        raise Exception("The gradient is not well defined for comparison expressions")

    def hess(self, x):
        raise Exception("The hessian is not well defined for comparison expressions")

    def convexify(self, x, degree=1):
        raise NotImplementedError

    def _penalty_model(self, x, wrap):
        """Shared body of the Eq/LEq/L convexify methods: linearise the inner
        expression at x, move ``val`` into the offset and wrap the affine model
        in an l1 penalty shape; memoised per rounded x (expr.py:323-332,
        362-371, 401-410)."""
        k = self._get_key(x)
        if k in self._convexify_cache:
            return self._convexify_cache[k]
        model = self.expr.convexify(x, degree=1)
        model.b = model.b - self.val
        out = wrap(model)
        self._convexify_cache[k] = out
        return out


class EqExpr(CompExpr):
    """expr == val (expr.py:299-332)."""

    def eval(self, x, tol=DEFAULT_TOL, negated=False):
        assert tol >= 0.0
        close = np.allclose(self.expr.eval(x), self.val, atol=tol)
        return (not close) if negated else close

    def convexify(self, x, degree=1):
        """h(x) = val  ->  |h(x) - val| (expr.py:314-332)."""
        assert degree == 1
        return self._penalty_model(x, AbsExpr)


class LEqExpr(CompExpr):
    """expr <= val (expr.py:335-371)."""

    def eval(self, x, tol=DEFAULT_TOL, negated=False):
        assert tol >= 0.0
        v = self.expr.eval(x)
        if negated:
            # the tolerance flips side for the negated test (expr.py:347-349)
            return not np.all(v <= self.val - tol * np.ones(v.shape))
        return np.all(v <= self.val + tol * np.ones(v.shape))

    def convexify(self, x, degree=1):
        """g(x) <= val  ->  max(g(x) - val, 0) (expr.py:353-371)."""
        assert degree == 1
        return self._penalty_model(x, HingeExpr)


class LExpr(CompExpr):
    """expr < val (expr.py:374-410)."""

    def eval(self, x, tol=DEFAULT_TOL, negated=False):
        assert tol >= 0.0
        v = self.expr.eval(x)
        if negated:
            return not np.all(v < self.val - tol * np.ones(v.shape))
        return np.all(v < self.val + tol * np.ones(v.shape))

    def convexify(self, x, degree=1):
        assert degree == 1
        return self._penalty_model(x, HingeExpr)


class BoundExpr(object):
    """An expression tied to the Variable it is evaluated on (expr.py:413-437)."""

    def __init__(self, expr, var):
        self.expr = expr
        self.var = var

    def eval(self):
        return self.expr.eval(self.var.get_value())

    def convexify(self, degree=1):
        at = self.var.get_value()
        assert at is not None
        return BoundExpr(self.expr.convexify(at, degree), self.var)


class TFExpr(Expr):
    """Placeholder kept for import compatibility (expr.py:440-451)."""

    def __init__(self, f, grad=None, hess=None, sess=None):
        self.sess = sess
        super(TFExpr, self).__init__(f, grad, hess)
