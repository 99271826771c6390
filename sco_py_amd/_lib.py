"""ctypes binding of libsco_hip.so (the C ABI declared in include/sco_hip.h).

There is deliberately NO CPU fallback: if the shared library is missing, or no
gfx950 device is visible when a solve is requested, the call raises.
"""
import ctypes as C
import os

import numpy as np

from . import _build

SCO_OK = 0
ERR_NAMES = {-1: "SCO_ERR_ARG", -2: "SCO_ERR_DEVICE", -3: "SCO_ERR_NO_GPU",
             -4: "SCO_ERR_STATE", -5: "SCO_ERR_CAPACITY"}


class ScoHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (ERR_NAMES.get(code, "SCO_ERR"), code, msg))
        self.code = code


class QpSettings(C.Structure):
    """struct sco_qp_settings (include/sco_hip.h)."""
    _fields_ = [
        ("rho", C.c_double), ("sigma", C.c_double), ("alpha", C.c_double),
        ("eps_abs", C.c_double), ("eps_rel", C.c_double),
        ("eps_prim_inf", C.c_double), ("eps_dual_inf", C.c_double),
        ("max_iter", C.c_int), ("check_termination", C.c_int), ("scaling", C.c_int),
        ("warm_start", C.c_int), ("adaptive_rho", C.c_int), ("adaptive_rho_interval", C.c_int),
        ("adaptive_rho_tolerance", C.c_double),
    ]


class SqpParams(C.Structure):
    """struct sco_sqp_params (include/sco_hip.h); mirrors Solver attributes."""
    _fields_ = [
        ("improve_ratio_threshold", C.c_double), ("min_trust_region_size", C.c_double),
        ("min_approx_improve", C.c_double), ("trust_shrink_ratio", C.c_double),
        ("trust_expand_ratio", C.c_double), ("cnt_tolerance", C.c_double),
        ("merit_coeff_increase_ratio", C.c_double), ("initial_trust_region_size", C.c_double),
        ("initial_penalty_coeff", C.c_double),
        ("max_merit_coeff_increases", C.c_int), ("compound_penalty", C.c_int),
        ("duplicate_rows", C.c_int), ("max_sqp_iters", C.c_int),
        ("memoize_rounded", C.c_int), ("warm_start_qps", C.c_int), ("admm_slice", C.c_int),
    ]


class TrajoptDesc(C.Structure):
    """struct sco_trajopt_desc (include/sco_hip.h)."""
    _fields_ = [
        ("batch", C.c_int), ("dof", C.c_int), ("horizon", C.c_int), ("n_points", C.c_int),
        ("n_obstacles", C.c_int), ("family", C.c_int), ("analytic_jac", C.c_int),
        ("prox_count", C.c_int), ("span", C.c_int), ("n_eq_rows", C.c_int),
    ]


_DP = C.POINTER(C.c_double)
_IP = C.POINTER(C.c_int)
_LP = C.POINTER(C.c_longlong)

# every symbol include/sco_hip.h declares: name -> (restype, argtypes)
ABI = {
    "sco_last_error": (C.c_char_p, []),
    "sco_version": (C.c_int, []),
    "sco_device_count": (C.c_int, [_IP]),
    "sco_qp_default_settings": (None, [C.POINTER(QpSettings)]),
    "sco_qp_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, _IP, _IP, _IP, _IP, C.POINTER(C.c_void_p)]),
    "sco_qp_destroy": (C.c_int, [C.c_void_p]),
    "sco_qp_load": (C.c_int, [C.c_void_p, _DP, _DP, _DP, _DP, _DP, _IP]),
    "sco_qp_set_bounds": (C.c_int, [C.c_void_p, _DP, _DP]),
    "sco_qp_solve": (C.c_int, [C.c_void_p, C.POINTER(QpSettings), _DP, _DP, _IP, _IP, _DP]),
    "sco_qp_info": (C.c_int, [C.c_void_p, _IP]),
    "sco_qp_adaptive_info": (C.c_int, [C.c_void_p, _DP, _IP]),
    "sco_qp_last_timing": (C.c_int, [C.c_void_p, _DP]),
    "sco_sqp_default_params": (None, [C.POINTER(SqpParams)]),
    "sco_sqp_create": (C.c_int, [C.c_int, C.POINTER(TrajoptDesc), C.POINTER(C.c_void_p)]),
    "sco_sqp_create_rows": (C.c_int, [C.c_int, C.POINTER(TrajoptDesc), C.c_int, _IP, _IP, _IP, C.POINTER(C.c_void_p)]),
    "sco_sqp_destroy": (C.c_int, [C.c_void_p]),
    "sco_sqp_load": (C.c_int, [C.c_void_p, _DP, _DP, _DP, _DP, _IP, _DP, _DP]),
    "sco_sqp_load_target": (C.c_int, [C.c_void_p, _DP]),
    "sco_sqp_load_quadratic": (C.c_int, [C.c_void_p, _DP, _DP, _DP]),
    "sco_sqp_load_program": (C.c_int, [C.c_void_p, C.c_int, _IP, _IP, C.c_int, _DP, C.c_int, _DP]),
    "sco_sqp_load_program_steps": (C.c_int, [C.c_void_p, C.c_int, _IP, _IP, C.c_int, _DP, C.c_int, _DP]),
    "sco_sqp_load_obj_weights": (C.c_int, [C.c_void_p, _DP]),
    "sco_sqp_load_acc_weights": (C.c_int, [C.c_void_p, _DP]),
    "sco_sqp_load_linear_rows": (C.c_int, [C.c_void_p, _DP, _DP]),
    "sco_sqp_set_circle_rows": (C.c_int, [C.c_void_p, C.c_int]),
    "sco_sqp_load_vel_limit": (C.c_int, [C.c_void_p, _DP]),
    "sco_sqp_load_joint_limits": (C.c_int, [C.c_void_p, _DP, _DP]),
    "sco_sqp_load_ee_cost": (C.c_int, [C.c_void_p, _DP, _DP]),
    "sco_sqp_set_groups": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_uint)]),
    "sco_sqp_fetch_groups": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint)]),
    "sco_sqp_fetch_stalled_groups": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint)]),
    "sco_sqp_fetch_flags": (C.c_int, [C.c_void_p, _IP]),
    "sco_sqp_last_rounds": (C.c_int, [C.c_void_p, _IP]),
    "sco_sqp_last_launches": (C.c_int, [C.c_void_p, _IP, _IP]),
    "sco_sqp_last_tiers": (C.c_int, [C.c_void_p, _DP, _LP, _IP]),
    "sco_sqp_solve": (C.c_int, [C.c_void_p, C.POINTER(SqpParams), C.POINTER(QpSettings)]),
    "sco_sqp_fetch": (C.c_int, [C.c_void_p, _DP, _IP, _IP, _IP, _LP, _DP, _DP]),
    "sco_sqp_trace": (C.c_int, [C.c_void_p, C.c_int, _DP, _IP]),
    "sco_sqp_last_timing": (C.c_int, [C.c_void_p, _DP]),
}

_lib = None


def lib_path():
    return _build.LIB


def load():
    """dlopen the in-tree shared library and bind the ABI; raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise ScoHipError(-3, "libsco_hip.so not built (%s); run `python __graft_entry__.py` "
                              "or sco_py_amd._build.build() -- there is no CPU fallback" % path)
    lib = C.CDLL(path)
    for name, (res, args) in ABI.items():
        fn = getattr(lib, name)   # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(code):
    if code != SCO_OK:
        msg = load().sco_last_error()
        raise ScoHipError(code, msg.decode() if msg else "")


def dptr(a):
    return None if a is None else a.ctypes.data_as(_DP)


def iptr(a):
    return None if a is None else a.ctypes.data_as(_IP)


def device_count():
    n = C.c_int(0)
    check(load().sco_device_count(C.byref(n)))
    return n.value


def default_qp_settings(**kw):
    s = QpSettings()
    load().sco_qp_default_settings(C.byref(s))
    for k, v in kw.items():
        if not hasattr(s, k):
            raise TypeError("unknown QP setting %r" % k)
        setattr(s, k, v)
    return s


def default_sqp_params(**kw):
    p = SqpParams()
    load().sco_sqp_default_params(C.byref(p))
    for k, v in kw.items():
        if not hasattr(p, k):
            raise TypeError("unknown SQP parameter %r" % k)
        setattr(p, k, v)
    return p


class BatchedQP(object):
    """RAII wrapper of a sco_qp handle: B QPs sharing one sparsity pattern."""

    def __init__(self, batch, n, m, Pp, Pi, Ap, Ai, device=0):
        self._h = C.c_void_p()
        self.batch, self.n, self.m = int(batch), int(n), int(m)
        self._pat = [np.ascontiguousarray(a, dtype=np.int32) for a in (Pp, Pi, Ap, Ai)]
        self.nnzP, self.nnzA = int(self._pat[0][-1]), int(self._pat[2][-1])
        check(load().sco_qp_create(device, self.batch, self.n, self.m,
                                   iptr(self._pat[0]), iptr(self._pat[1]),
                                   iptr(self._pat[2]), iptr(self._pat[3]), C.byref(self._h)))

    def close(self):
        if self._h:
            load().sco_qp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _arr(self, a, shape, dtype=np.float64):
        a = np.ascontiguousarray(a, dtype=dtype)
        if a.shape != shape:
            raise ValueError("expected shape %r, got %r" % (shape, a.shape))
        return a

    def load(self, Pval, q, Aval, l, u, row_weight=None):
        B = self.batch
        Pval = self._arr(Pval, (B, self.nnzP)); q = self._arr(q, (B, self.n))
        Aval = self._arr(Aval, (B, self.nnzA)); l = self._arr(l, (B, self.m)); u = self._arr(u, (B, self.m))
        w = None if row_weight is None else self._arr(row_weight, (B, self.m), np.int32)
        check(load().sco_qp_load(self._h, dptr(Pval), dptr(q), dptr(Aval), dptr(l), dptr(u), iptr(w)))

    def set_bounds(self, l, u):
        l = self._arr(l, (self.batch, self.m)); u = self._arr(u, (self.batch, self.m))
        check(load().sco_qp_set_bounds(self._h, dptr(l), dptr(u)))

    def solve(self, settings=None):
        st = settings if settings is not None else default_qp_settings()
        B = self.batch
        x = np.zeros((B, self.n)); y = np.zeros((B, max(self.m, 1)))
        status = np.zeros(B, dtype=np.int32); iters = np.zeros(B, dtype=np.int32)
        resid = np.zeros((B, 2))
        check(load().sco_qp_solve(self._h, C.byref(st), dptr(x), dptr(y), iptr(status), iptr(iters), dptr(resid)))
        return x, y[:, : self.m], status, iters, resid

    def info(self):
        out = np.zeros(4, dtype=np.int32)
        check(load().sco_qp_info(self._h, iptr(out)))
        return dict(n_elim=int(out[0]), n_core=int(out[1]), lds_admm=int(out[2]), ncpl=int(out[3]))

    def adaptive_info(self):
        """(rho each problem ended with, number of rho updates) of the last adaptive_rho solve."""
        rho = np.zeros(self.batch); upd = np.zeros(self.batch, dtype=np.int32)
        check(load().sco_qp_adaptive_info(self._h, dptr(rho), iptr(upd)))
        return rho, upd

    def last_timing(self):
        ms = np.zeros(2)
        check(load().sco_qp_last_timing(self._h, dptr(ms)))
        return dict(setup_ms=float(ms[0]), admm_ms=float(ms[1]))
