"""QP atoms and the assemble-and-solve seam of the SCO front-end.

Mirror of ``sco_py.sco_osqp.osqp_utils``
(/root/reference/sco_py/sco_osqp/osqp_utils.py).  The model classes keep the
reference's names and fields; :func:`optimize` keeps its signature and return
shape ``(solve_res, var_to_index_dict)`` but, instead of building dense matrices
and calling the third-party OSQP library (osqp_utils.py:146-216), assembles the
sparse triplets directly and hands them to the MI355X solver through the C ABI
(``sco_qp_*`` in include/sco_hip.h).  There is no CPU path.
"""
from types import SimpleNamespace
from typing import List

import atexit
import collections
import os
import threading

import numpy as np
import scipy.sparse as sp

from .. import _lib
from .variable import Variable  # noqa: F401  (type name used by callers, as in the reference)

DEFAULT_MAX_ITER = int(1e05)     # osqp_utils.py:10
DEFAULT_SIGMA = 5e-10            # osqp_utils.py:11
DEFAULT_RHO = 1e-01              # osqp_utils.py:12
DEFAULT_ADAPTIVE_RHO = False     # osqp_utils.py:13
DEFAULT_EPS_ABS = 1e-06          # osqp_utils.py:14
DEFAULT_EPS_REL = 1e-09          # osqp_utils.py:15


class OSQPVar(object):
    """One scalar QP variable: name, bounds (the trust region is written here),
    last solver value (osqp_utils.py:17-51)."""

    _created = 0     # creation counter: deterministic tie-break among equal names (all slacks share one)

    def __init__(self, var_name, lb=-np.inf, ub=np.inf, val=None):
        self.var_name = var_name
        self._lower_bound = lb
        self._upper_bound = ub
        self.val = val
        OSQPVar._created += 1
        self._serial = OSQPVar._created

    def __lt__(self, other_osqp_var):
        # ordering by name only: columns of the QP are the name-sorted variables
        return self.var_name < other_osqp_var.var_name

    def __repr__(self):
        return f"OSQPVar with name {self.var_name}"

    def get_lower_bound(self):
        return self._lower_bound

    def set_lower_bound(self, lb_val):
        assert isinstance(lb_val, float)      # np.float64 passes, int does not (osqp_utils.py:41)
        assert not np.isnan(lb_val)
        self._lower_bound = lb_val

    def get_upper_bound(self):
        return self._upper_bound

    def set_upper_bound(self, ub_val):
        assert isinstance(ub_val, float)
        assert not np.isnan(ub_val)
        self._upper_bound = ub_val


class OSQPLinearObj(object):
    """coeff * osqp_var (osqp_utils.py:54-68)."""

    def __init__(self, osqp_var, coeff):
        self.osqp_var = osqp_var
        self.coeff = coeff

    def __repr__(self):
        return f"OSQPLinearObj with osqp_var={self.osqp_var}, coeff={self.coeff}"

    def get_all_vars(self):
        return [self.osqp_var]


class OSQPQuadraticObj(object):
    """0.5 * sum_i coeffs[i] * osqp_vars1[i] * osqp_vars2[i] (osqp_utils.py:71-90)."""

    def __init__(self, osqp_vars1, osqp_vars2, coeffs):
        assert osqp_vars1.shape == osqp_vars2.shape == coeffs.shape
        assert len(osqp_vars1.shape) == 1
        self.osqp_vars1 = osqp_vars1
        self.osqp_vars2 = osqp_vars2
        self.coeffs = coeffs

    def __repr__(self):
        return (f"Quadratic Objective with osqp_vars1={self.osqp_vars1}, "
                f"osqp_vars2={self.osqp_vars2}, coeffs={self.coeffs}")

    def get_all_vars(self):
        return self.osqp_vars1.tolist() + self.osqp_vars2.tolist()


class OSQPLinearConstraint(object):
    """lb <= sum_i coeffs[i] * osqp_vars[i] <= ub (osqp_utils.py:93-110)."""

    def __init__(self, osqp_vars, coeffs, lb, ub):
        assert osqp_vars.shape == coeffs.shape
        self.osqp_vars = osqp_vars
        self.coeffs = coeffs
        self.lb = lb
        self.ub = ub

    def __repr__(self):
        return (f"OSQPLinearConstraint with osqp_vars={self.osqp_vars}, coeffs={self.coeffs}, "
                f"lb = {self.lb}, ub = {self.ub}")

    def get_all_vars(self):
        return self.osqp_vars.tolist()


# --------------------------------------------------------------------------
# assembly (S5) -- the reference's Python loops over dense matrices
# (osqp_utils.py:146-193) restated as triplet accumulation
# --------------------------------------------------------------------------
def _scalar(v):
    """Row bounds arrive as floats or as 1-element arrays (b[i] of an (r, 1) offset)."""
    return v if isinstance(v, float) else np.asarray(v, dtype=np.float64).reshape(-1)[0]


def fold_repeated_constraints(osqp_lin_cnt_exprs):
    """The reference appends the SAME penalty-row objects to its constraint list on
    every update_obj call (prob.py:508-509), so the k-th QP carries k
    identical copies of each of them.  Identical rows behave in ADMM exactly like one
    row of multiplicity k, which is what the device ABI takes
    (`row_weight`).  Returns (unique constraint objects in first-seen order, counts)."""
    first = {}
    uniq, count = [], []
    for cnt in osqp_lin_cnt_exprs:
        k = first.get(id(cnt))
        if k is None:
            first[id(cnt)] = len(uniq)
            uniq.append(cnt); count.append(1)
        else:
            count[k] += 1
    return uniq, np.asarray(count, dtype=np.int32)


def assemble_qp(osqp_vars, osqp_quad_objs, osqp_lin_objs, osqp_lin_cnt_exprs):
    """Returns (P_triu csc, q, A csc, l, u, var_to_index_dict).

    Column order: variables sorted by name (osqp_utils.py:136-142).  The reference leaves
    ties (every slack is named "z+_pos_osqp_var") and the order of the bound rows to set
    iteration, i.e. to object ids; here ties go by creation order and the
    bound rows follow the columns, so structurally identical problems get identical
    sparsity patterns (a QP's solution does not depend on either order).
    Row order: the constraints in list order, then one bound row per variable."""
    ordered = sorted(osqp_vars, key=lambda v: (v.var_name, getattr(v, "_serial", 0)))
    index = {v: k for k, v in enumerate(ordered)}
    n = len(osqp_vars)

    q = np.zeros(n)
    for term in osqp_lin_objs:
        q[index[term.osqp_var]] += term.coeff           # += (osqp_utils.py:148)

    # P: off-diagonal pairs put half the coefficient at [min, max], the diagonal
    # gets the full coefficient; repeated entries add up (osqp_utils.py:153-163)
    pr, pc, pv = [], [], []
    for quad in osqp_quad_objs:
        for k in range(quad.coeffs.shape[0]):
            a = index[quad.osqp_vars1[k]]
            b = index[quad.osqp_vars2[k]]
            pr.append(min(a, b)); pc.append(max(a, b))
            pv.append(quad.coeffs[k] if a == b else 0.5 * quad.coeffs[k])
    P = sp.coo_matrix((np.asarray(pv, dtype=np.float64), (pr, pc)), shape=(n, n)).tocsc()
    P.sum_duplicates()
    P.sort_indices()

    # A: assignment semantics -- a variable listed twice in one row keeps the LAST
    # coefficient (osqp_utils.py:179-181)
    m = n + len(osqp_lin_cnt_exprs)
    l = np.zeros(m)
    u = np.zeros(m)
    cells = {}
    for row, cnt in enumerate(osqp_lin_cnt_exprs):
        l[row] = _scalar(cnt.lb)
        u[row] = _scalar(cnt.ub)
        for k in range(cnt.coeffs.shape[0]):
            cells[(row, index[cnt.osqp_vars[k]])] = cnt.coeffs[k]
    row = len(osqp_lin_cnt_exprs)
    for v in ordered:
        cells[(row, index[v])] = 1.0
        l[row] = v.get_lower_bound()
        u[row] = v.get_upper_bound()
        row += 1
    if cells:
        rc = np.array(list(cells.keys()), dtype=np.int64)
        vals = np.array(list(cells.values()), dtype=np.float64)
        keep = vals != 0.0                     # csc_matrix(dense) drops exact zeros (osqp_utils.py:193)
        A = sp.coo_matrix((vals[keep], (rc[keep, 0], rc[keep, 1])), shape=(m, n)).tocsc()
    else:
        A = sp.csc_matrix((m, n))
    A.sort_indices()
    P.eliminate_zeros()
    return P, q, A, l, u, index


# Device handles are kept per sparsity pattern: the SQP loop sends the same pattern again for every
# trust-region retry and every SQP iteration (the re-appended rows are folded into weights), and
# creating a handle (symbolic analysis + device allocations) costs more than a well-conditioned QP.
# Extension (off by default, NOT reference behaviour): start every QP from the previous solution of the
# same pattern's handle, as OSQP's own warm start would; the reference builds a new OSQP object per QP and
# always starts cold (osqp_utils.py:195).  Iterates and iteration counts change, answers agree to the QP tolerances.
WARM_START = False

_HANDLE_CACHE = collections.OrderedDict()
_HANDLE_CACHE_MAX = 8
_TIER_SWITCHES = ("SCO_QP_NO_ELIM", "SCO_QP_NO_RL", "SCO_QP_NO_REG", "SCO_QP_NO_FAST", "SCO_QP_FORCE_BIG", "SCO_QP_NO_BT",
                  "SCO_QP_FACTOR_CHOLESKY")
_HANDLE_LOCK = threading.Lock()


def clear_handle_cache():
    """Free every cached device handle (also runs at interpreter exit)."""
    with _HANDLE_LOCK:
        while _HANDLE_CACHE:
            _, qp = _HANDLE_CACHE.popitem()
            qp.close()


atexit.register(clear_handle_cache)


def _cached_handle(B, n, m, P0, A0):
    key = (B, n, m, P0.indptr.tobytes(), P0.indices.tobytes(), A0.indptr.tobytes(), A0.indices.tobytes(),
           tuple(os.environ.get(k) for k in _TIER_SWITCHES))     # the tier is chosen when a handle is created
    qp = _HANDLE_CACHE.pop(key, None)
    if qp is None:
        qp = _lib.BatchedQP(B, n, m, P0.indptr, P0.indices, A0.indptr, A0.indices)
    _HANDLE_CACHE[key] = qp                      # most recently used last
    while len(_HANDLE_CACHE) > _HANDLE_CACHE_MAX:
        _, old = _HANDLE_CACHE.popitem(last=False)
        old.close()
    return qp


def _solve_qp_batch(requests):
    """Solve a list of QPs that share ONE sparsity pattern and one settings tuple in a
    single device launch.  Each request is a dict with P, q, A, l, u, w (row weights or
    None) and settings = (eps_abs, eps_rel, max_iter, rho, sigma, adaptive_rho).
    Returns a list of (x, status, iters)."""
    r0 = requests[0]
    P0, A0 = r0["P"], r0["A"]
    n, m = A0.shape[1], A0.shape[0]
    eps_abs, eps_rel, max_iter, rho, sigma, adaptive_rho = r0["settings"]
    B = len(requests)
    Pv = np.stack([r["P"].data for r in requests]) if P0.nnz else np.zeros((B, 0))
    Av = np.stack([r["A"].data for r in requests]) if A0.nnz else np.zeros((B, 0))
    w = None
    if any(r["w"] is not None for r in requests):
        w = np.stack([r["w"] if r["w"] is not None else np.ones(m, dtype=np.int32) for r in requests])
    st = _lib.default_qp_settings(rho=rho, sigma=sigma, eps_abs=eps_abs, eps_rel=eps_rel, max_iter=int(max_iter),
                                  warm_start=1 if WARM_START else 0, adaptive_rho=1 if adaptive_rho else 0)
    with _HANDLE_LOCK:
        qp = _cached_handle(B, n, m, P0, A0)
        try:
            qp.load(Pv, np.stack([r["q"] for r in requests]), Av,
                    np.stack([r["l"] for r in requests]), np.stack([r["u"] for r in requests]), w)
            x, _y, status, iters, _res = qp.solve(st)
        except Exception:
            _HANDLE_CACHE.popitem()              # do not keep a handle that failed
            qp.close()
            raise
    return [(x[b], int(status[b]), int(iters[b])) for b in range(B)]


def _solve_qp(P, q, A, l, u, eps_abs, eps_rel, max_iter, rho, sigma, w=None, adaptive_rho=False):
    """One QP on the GPU through the C ABI.  Inside ``batching.solve_many`` the call is
    parked until every concurrently running solve has reached its next QP, and QPs with
    the same pattern go to the device together.  Returns (x, status, iters)."""
    req = dict(P=P, q=q, A=A, l=l, u=u, w=w, settings=(eps_abs, eps_rel, max_iter, rho, sigma, 1.0 if adaptive_rho else 0.0))
    from . import batching
    server = batching.current_server()
    if server is not None:
        return server.submit(req)
    return _solve_qp_batch([req])[0]


# @profile
def optimize(
    osqp_vars: List[OSQPVar],
    _sco_vars: List[Variable],
    osqp_quad_objs: List[OSQPQuadraticObj],
    osqp_lin_objs: List[OSQPLinearObj],
    osqp_lin_cnt_exprs: List[OSQPLinearConstraint],
    eps_abs: float = DEFAULT_EPS_ABS,
    eps_rel: float = DEFAULT_EPS_REL,
    max_iter: int = DEFAULT_MAX_ITER,
    rho: float = DEFAULT_RHO,
    adaptive_rho: bool = DEFAULT_ADAPTIVE_RHO,
    sigma: float = DEFAULT_SIGMA,
    verbose: bool = False,
):
    """Assemble the current QP and solve it (osqp_utils.py:113-221).

    Returns ``(solve_res, var_to_index_dict)``; callers read ``solve_res.x`` and
    ``solve_res.info.status_val`` (prob.py:197, 202)."""
    uniq_cnts, counts = fold_repeated_constraints(osqp_lin_cnt_exprs)
    P, q, A, l, u, index = assemble_qp(osqp_vars, osqp_quad_objs, osqp_lin_objs, uniq_cnts)
    w = None
    if counts.size and int(counts.max()) > 1:
        w = np.concatenate([counts, np.ones(len(osqp_vars), dtype=np.int32)])     # bound rows appear once
    x, status, iters = _solve_qp(P, q, A, l, u, eps_abs, eps_rel, max_iter, rho, sigma, w=w,
                                 adaptive_rho=bool(adaptive_rho))
    solve_res = SimpleNamespace(x=x, info=SimpleNamespace(status_val=status, iter=iters))
    if status == -2 and verbose:
        print("ERROR! OSQP Solver hit max iteration limit. Either reduce your tolerances "
              "or increase the max iterations!")
    return (solve_res, index)


def update_osqp_vars(var_to_osqp_indices_dict, solver_values):
    """Scatter the solution vector into the atoms (osqp_utils.py:224-229)."""
    for atom, k in var_to_osqp_indices_dict.items():
        atom.val = solver_values[k]


def print_osqp_vars_and_sol(solve_res_x, var_to_index_dict):
    for atom, k in var_to_index_dict.items():
        print(f"{atom}, {solve_res_x[k]}")
