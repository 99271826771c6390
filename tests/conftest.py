"""Shared test plumbing.

`-m "not gpu"`: oracle vs golden vectors, host logic of the mirror API (with the
oracle ADMM patched in at the QP seam -- test infrastructure only), ABI export
checks.  `-m gpu`: parity tests proper, all through the C ABI of libsco_hip.so.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def built_artifacts():
    """Make sure the checker (oracle C restatement) and, when hipcc is around,
    the product library exist.  On the GPU box both arrive prebuilt."""
    from oracle import osqp_ref
    osqp_ref.build()
    from sco_py_amd import _build
    if _build.needs_build():
        _build.build()
    return True


def _gpu_count():
    try:
        from sco_py_amd import _lib
        return _lib.device_count()
    except Exception:
        return 0


@pytest.fixture(scope="session")
def gpu():
    n = _gpu_count()
    if n <= 0:
        pytest.fail("test marked gpu but libsco_hip.so sees no device (no CPU fallback exists)")
    return n


@pytest.fixture
def oracle_qp_backend(monkeypatch):
    """Route the mirror API's single QP seam to the oracle ADMM so the HOST logic
    (assembly, penalty lowering, SQP control flow) can be tested without a GPU.
    The product never does this by itself."""
    from oracle import osqp_ref
    from sco_py_amd.sco_osqp import osqp_utils
    log = []

    def cpu_batch(requests):
        out = []
        for rq in requests:
            eps_abs, eps_rel, max_iter, rho, sigma, adaptive = rq["settings"]
            r = osqp_ref.solve(rq["P"], rq["q"], rq["A"], rq["l"], rq["u"], w=rq["w"], eps_abs=eps_abs,
                               eps_rel=eps_rel, max_iter=int(max_iter), rho=rho, sigma=sigma,
                               adaptive_rho=1 if adaptive else 0)
            log.append(dict(P=rq["P"].toarray(), q=rq["q"].copy(), A=rq["A"].toarray(), l=rq["l"].copy(),
                            u=rq["u"].copy(), w=None if rq["w"] is None else rq["w"].copy(), x=r.x.copy(),
                            status=r.info.status_val, iters=r.info.iter))
            out.append((r.x, r.info.status_val, r.info.iter))
        return out

    monkeypatch.setattr(osqp_utils, "_solve_qp_batch", cpu_batch)
    return log


@pytest.fixture
def hip_qp_log(monkeypatch):
    """Record the QPs the mirror API sends through the real HIP seam."""
    from sco_py_amd.sco_osqp import osqp_utils
    real = osqp_utils._solve_qp_batch
    log = []

    def logged(requests):
        res = real(requests)
        for rq, (x, st, it) in zip(requests, res):
            log.append(dict(P=rq["P"].copy(), q=rq["q"].copy(), A=rq["A"].copy(), l=rq["l"].copy(), u=rq["u"].copy(),
                            w=rq["w"], x=x.copy(), status=st, iters=it))
        return res

    monkeypatch.setattr(osqp_utils, "_solve_qp_batch", logged)
    return log


def mirror_mods():
    """Namespace of the mirror API classes, shaped like the one make_golden.py
    builds from the reference modules."""
    from types import SimpleNamespace
    from sco_py_amd import expr
    from sco_py_amd.sco_osqp import osqp_utils, prob, solver, variable
    return SimpleNamespace(
        Expr=expr.Expr, AffExpr=expr.AffExpr, QuadExpr=expr.QuadExpr, EqExpr=expr.EqExpr,
        LEqExpr=expr.LEqExpr, LExpr=expr.LExpr, BoundExpr=expr.BoundExpr, AbsExpr=expr.AbsExpr,
        HingeExpr=expr.HingeExpr, CompExpr=expr.CompExpr, Variable=variable.Variable,
        OSQPVar=osqp_utils.OSQPVar, OSQPLinearObj=osqp_utils.OSQPLinearObj,
        OSQPQuadraticObj=osqp_utils.OSQPQuadraticObj, OSQPLinearConstraint=osqp_utils.OSQPLinearConstraint,
        Prob=prob.Prob, Solver=solver.Solver, osqp_utils=osqp_utils)


def load_golden_qps(g, prefix, sparse=False):
    import scipy.sparse as sp
    out = []
    for k in range(int(g[prefix + "n_qp"])):
        b = "%sqp%d_" % (prefix, k)
        d = {nm: g[b + nm] for nm in ("q", "l", "u", "x")}
        d["status"] = int(g[b + "status"]); d["iters"] = int(g[b + "iters"])
        for nm in ("P", "A"):
            if sparse:
                d[nm] = sp.coo_matrix((g[b + nm + "_val"], (g[b + nm + "_row"], g[b + nm + "_col"])),
                                      shape=tuple(g[b + nm + "_shape"])).toarray()
            else:
                d[nm] = g[b + nm]
        out.append(d)
    return out


def expand_weighted_qp(qp):
    """Oracle QP with row multiplicities -> physical rows in the reference's order
    [linear rows, k copies of the penalty rows, bound rows] (prob.py:508-509)."""
    A, l, u, w = qp["A"], qp["l"], qp["u"], qp.get("w")
    if w is None or int(np.max(w)) == 1:
        return qp["P"], qp["q"], A, l, u
    n = qp["q"].shape[0]; m = A.shape[0]; k = int(np.max(w))
    nl = np.where(w == k)[0]
    ones = np.where(w == 1)[0]
    top, bnd = ones[ones < m - n], ones[ones >= m - n]
    order = list(top) + list(nl) * k + list(bnd)
    return qp["P"], qp["q"], A[order], l[order], u[order]


def assert_qp_close(a, P, q, A, l, u, tag, tol=1e-9):
    for nm, x, y in (("P", a["P"], np.triu(P)), ("q", a["q"], q), ("A", a["A"], A), ("l", a["l"], l), ("u", a["u"], u)):
        assert x.shape == y.shape, (tag, nm, x.shape, y.shape)
        fin = np.isfinite(x)
        assert np.array_equal(fin, np.isfinite(y)), (tag, nm)
        assert np.array_equal(x[~fin], y[~fin]), (tag, nm)
        err = np.max(np.abs(x[fin] - y[fin]) / (1 + np.abs(x[fin])), initial=0.0)
        assert err < tol, (tag, nm, err)
