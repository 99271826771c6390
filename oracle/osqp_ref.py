"""ctypes front-end of oracle/osqp_ref.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module (as the checker / the timed CPU baseline).  The product package
``sco_py_amd`` never imports anything from ``oracle/``.

The C file restates the third-party OSQP solve the reference performs at
/root/reference/sco_py/sco_osqp/osqp_utils.py:195-216 (see the header of
osqp_ref.c for what is and is not pinned).
"""
import ctypes as C
import os
import subprocess
from types import SimpleNamespace

import numpy as np
import scipy.sparse as sp

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_osqp_ref.so")
_SRC = os.path.join(_HERE, "osqp_ref.c")
_SO_LD = os.path.join(_HERE, "_osqp_ref_ld.so")          # the same file compiled in x87 extended precision (osqp_ref_ld.c)
_SRC_LD = os.path.join(_HERE, "osqp_ref_ld.c")


class Settings(C.Structure):
    _fields_ = [
        ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
        ("eps_abs", C.c_double), ("eps_rel", C.c_double),
        ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
        ("max_iter", C.c_int), ("check_termination", C.c_int), ("scaling", C.c_int),
        ("expand_dups", C.c_int), ("linsys", C.c_int),
        ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int), ("adaptive_rho_tolerance", C.c_double),
    ]


class Info(C.Structure):
    _fields_ = [("status", C.c_int), ("iters", C.c_int), ("obj", C.c_double),
                ("pri_res", C.c_double), ("dua_res", C.c_double), ("rho", C.c_double), ("rho_updates", C.c_int)]


def build(force=False):
    """Compile the C restatement (gcc, host only).  SCO_ORACLE_SANITIZE=1 (scripts/cpu_sanitize.sh, CPU only) builds
    and loads an AddressSanitizer + UndefinedBehaviorSanitizer variant next to it instead."""
    global _SO
    if os.environ.get("SCO_ORACLE_SANITIZE", "0") == "1":
        _SO = os.path.join(_HERE, "_osqp_ref_asan.so")
        if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
            subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                                   "-fno-sanitize-recover=undefined", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"])
        return _SO
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO, _SRC, "-lm"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.osqp_ref_default_settings.argtypes = [C.POINTER(Settings)]
        _lib.osqp_ref_solve.restype = C.c_int
    return _lib


_lib_ld = None


def build_extended(force=False):
    """Compile osqp_ref_ld.c: osqp_ref.c with `double` -> `long double` behind the same double ABI (gcc, host only)."""
    global _SO_LD
    newest = max(os.path.getmtime(_SRC), os.path.getmtime(_SRC_LD))
    if os.environ.get("SCO_ORACLE_SANITIZE", "0") == "1":       # scripts/cpu_sanitize.sh: ASan + UBSan variant next to it
        _SO_LD = os.path.join(_HERE, "_osqp_ref_ld_asan.so")
        if force or not os.path.exists(_SO_LD) or os.path.getmtime(_SO_LD) < newest:
            subprocess.check_call(["gcc", "-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined",
                                   "-fno-sanitize-recover=undefined", "-shared", "-fPIC", "-o", _SO_LD, _SRC_LD, "-lm"], cwd=_HERE)
        return _SO_LD
    if force or not os.path.exists(_SO_LD) or os.path.getmtime(_SO_LD) < newest:
        subprocess.check_call(["gcc", "-O2", "-shared", "-fPIC", "-o", _SO_LD, _SRC_LD, "-lm"], cwd=_HERE)
    return _SO_LD


def lib_extended():
    global _lib_ld
    if _lib_ld is None:
        build_extended()
        _lib_ld = C.CDLL(_SO_LD)
        _lib_ld.osqp_ref_default_settings.argtypes = [C.POINTER(Settings)]
        _lib_ld.osqp_ref_solve.restype = C.c_int
    return _lib_ld


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


def default_settings(**kw):
    s = Settings()
    lib().osqp_ref_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise TypeError("unknown setting %r" % k)
        setattr(s, k, v)
    return s


def solve(P, q, A, l, u, w=None, trace_cap=0, extended=False, **settings):
    """Solve  min 1/2 x'Px + q'x  s.t.  l <= Ax <= u  the way OSQP 0.6 would.

    P: (n, n) dense or sparse; only its upper triangle is read (the reference
       hands OSQP an upper-triangular P, osqp_utils.py:153-163, 192).
    A: (m, n) dense or sparse.  w: optional integer row multiplicities.
    Returns a namespace with x, y, info.status_val, info.iter, ... mirroring the
    fields the reference reads (prob.py:197, 202; osqp_utils.py:218).
    extended=True runs the x87 long double build of the same code (osqp_ref_ld.c): where the algorithm stops once rounding
    noise is 2048 x smaller -- the referee between two float64 routes that disagree on an iteration count.
    """
    q = np.ascontiguousarray(q, dtype=np.float64).ravel()
    n = q.shape[0]
    Pu = sp.triu(sp.csc_matrix(P, dtype=np.float64), format="csc")
    Pu.sort_indices()
    Ac = sp.csc_matrix(A, dtype=np.float64) if A is not None else sp.csc_matrix((0, n))
    Ac.sort_indices()
    m = Ac.shape[0]
    l = np.ascontiguousarray(l, dtype=np.float64).ravel()
    u = np.ascontiguousarray(u, dtype=np.float64).ravel()
    assert Pu.shape == (n, n) and Ac.shape[1] == n and l.shape == (m,) and u.shape == (m,)
    st = default_settings(**settings)
    Pp, Pi, Px = Pu.indptr.astype(np.int32), Pu.indices.astype(np.int32), np.ascontiguousarray(Pu.data)
    Ap, Ai, Ax = Ac.indptr.astype(np.int32), Ac.indices.astype(np.int32), np.ascontiguousarray(Ac.data)
    x = np.zeros(max(n, 1)); y = np.zeros(max(m, 1))
    info = Info()
    wv = None
    if w is not None:
        wv = np.ascontiguousarray(w, dtype=np.int32).ravel()
        assert wv.shape == (m,)
    trace = np.zeros((max(trace_cap, 1), 4)); tl = C.c_int(0)
    rc = (lib_extended() if extended else lib()).osqp_ref_solve(
        C.c_int(n), C.c_int(m), _ip(Pp), _ip(Pi), _dp(Px), _dp(q), _ip(Ap), _ip(Ai), _dp(Ax),
        _dp(l), _dp(u), _ip(wv) if wv is not None else None, C.byref(st),
        _dp(x), _dp(y), C.byref(info), _dp(trace), C.c_int(trace_cap), C.byref(tl))
    res = SimpleNamespace()
    res.rc = rc
    res.x = x[:n].copy()
    res.y = y[:m].copy()
    res.info = SimpleNamespace(status_val=info.status, iter=info.iters, obj_val=info.obj,
                               pri_res=info.pri_res, dua_res=info.dua_res, rho_estimate=info.rho,
                               rho_updates=info.rho_updates)
    res.trace = trace[: tl.value].copy()
    return res


def kkt_violation(P, q, A, l, u, x, y):
    """Independent optimality check of a QP answer (no ADMM involved).

    Returns (primal infeasibility, stationarity residual, complementarity gap)
    in the inf-norm; all three vanish at an exact solution."""
    P = sp.csc_matrix(P, dtype=np.float64)
    Pfull = sp.triu(P) + sp.triu(P, 1).T
    A = sp.csc_matrix(A, dtype=np.float64)
    Axv = A @ x
    prim = max(0.0, float(np.max(np.maximum(l - Axv, 0.0), initial=0.0)),
               float(np.max(np.maximum(Axv - u, 0.0), initial=0.0)))
    stat = float(np.max(np.abs(Pfull @ x + q + A.T @ y), initial=0.0))
    yp, ym = np.maximum(y, 0.0), np.minimum(y, 0.0)
    fin_u, fin_l = np.isfinite(u) & (np.abs(u) < 1e29), np.isfinite(l) & (np.abs(l) < 1e29)
    comp = 0.0
    if m_any(fin_u):
        comp = max(comp, float(np.max(np.abs(yp[fin_u] * (u[fin_u] - Axv[fin_u])))))
    if m_any(fin_l):
        comp = max(comp, float(np.max(np.abs(ym[fin_l] * (Axv[fin_l] - l[fin_l])))))
    if m_any(~fin_u):
        comp = max(comp, float(np.max(yp[~fin_u], initial=0.0)))
    if m_any(~fin_l):
        comp = max(comp, float(np.max(-ym[~fin_l], initial=0.0)))
    return prim, stat, comp


def m_any(mask):
    return bool(np.any(mask))
