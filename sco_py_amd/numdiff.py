"""Finite-difference derivatives used wherever the reference calls numdifftools.

The reference differentiates black-box callables with ``numdifftools.Jacobian`` /
``numdifftools.Hessian`` (/root/reference/sco_py/expr.py:67, 108).  numdifftools
is a third-party package that is neither vendored in the reference tree nor
installed here; its exact step sequence is *parity unpinned*.  What the
reference's tests pin is the derivative itself (rtol 1e-5 / atol 1e-8 against the
analytic value, tests/sco_osqp/test_expr.py:71-78, 151-211).

The scheme below is the one ``sqp_pre_kernel`` in csrc/sco_sqp.hip implements
(its FD_BASE / FD_LEVELS ladder and ``richardson``), and ``oracle/sco_ref.py``'s
``fd_jacobian`` / ``fd_hessian``, so host, oracle and device agree to rounding:

  central differences  D(h) = (f(x + h e_j) - f(x - h e_j)) / (2h)
  on the step ladder   h_k = h0 / 2^k,  k = 0..LEVELS-1,  h0 = BASE * max(1, |x_j|)
  Richardson table     T[k][0] = D(h_k)
                       T[k][i] = T[k][i-1] + (T[k][i-1] - T[k-1][i-1]) / (4^i - 1)
  result               T[LEVELS-1][LEVELS-1]

The error expansion of a central difference contains only even powers of h, so
every Richardson column removes one more power of h^2.
"""
import numpy as np

BASE_STEP = 1.0 / 64.0   # exactly representable: x + h - x == h for |x| < 2^46
LEVELS = 4


def _richardson(cols):
    """cols[k] = D(h_k) for the halving ladder; returns the extrapolated value."""
    tab = [np.asarray(c, dtype=np.float64) for c in cols]
    for i in range(1, len(tab)):
        fac = 1.0 / (4.0 ** i - 1.0)
        tab = [tab[k] + (tab[k] - tab[k - 1]) * fac for k in range(1, len(tab))]
    return tab[0]


def jacobian(f, x):
    """d f / d x for f: R^n -> R^r given as a callable on a FLAT x; returns (r, n)."""
    x = np.asarray(x, dtype=np.float64).ravel()
    n = x.shape[0]
    out_cols = []
    for j in range(n):
        h0 = BASE_STEP * max(1.0, abs(x[j]))
        ladder = []
        for k in range(LEVELS):
            h = h0 / (2.0 ** k)
            xp = x.copy(); xm = x.copy()
            xp[j] += h; xm[j] -= h
            fp = np.asarray(f(xp), dtype=np.float64).ravel()
            fm = np.asarray(f(xm), dtype=np.float64).ravel()
            ladder.append((fp - fm) / (2.0 * h))
        out_cols.append(_richardson(ladder))
    if not out_cols:
        return np.zeros((0, 0))
    return np.stack(out_cols, axis=1)


def hessian(f, x):
    """Hessian of a scalar f given on a FLAT x; returns (n, n), symmetric.

    Second central differences on the same halving ladder, Richardson
    extrapolated (the stencil error again has only even powers of h)."""
    x = np.asarray(x, dtype=np.float64).ravel()
    n = x.shape[0]
    H = np.zeros((n, n))

    def fv(v):
        return float(np.asarray(f(v), dtype=np.float64).ravel()[0])

    f0 = fv(x)
    steps = [BASE_STEP * max(1.0, abs(x[j])) for j in range(n)]
    for i in range(n):
        for j in range(i, n):
            ladder = []
            for k in range(LEVELS):
                hi = steps[i] / (2.0 ** k)
                hj = steps[j] / (2.0 ** k)
                if i == j:
                    xp = x.copy(); xm = x.copy()
                    xp[i] += hi; xm[i] -= hi
                    ladder.append((fv(xp) - 2.0 * f0 + fv(xm)) / (hi * hi))
                else:
                    xpp = x.copy(); xpm = x.copy(); xmp = x.copy(); xmm = x.copy()
                    xpp[i] += hi; xpp[j] += hj
                    xpm[i] += hi; xpm[j] -= hj
                    xmp[i] -= hi; xmp[j] += hj
                    xmm[i] -= hi; xmm[j] -= hj
                    ladder.append((fv(xpp) - fv(xpm) - fv(xmp) + fv(xmm)) / (4.0 * hi * hj))
            val = float(_richardson(ladder))
            H[i, j] = val
            H[j, i] = val
    return H
