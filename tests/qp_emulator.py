"""NumPy emulation of qp_setup_kernel / qp_admm_kernel (csrc/sco_qp.hip).

Test infrastructure: runs the SAME index plans the C++ symbolic analysis
(csrc/qp_plan.cpp) produces through a line-by-line NumPy rendition of the device
phases, so the plans and the two-level (eliminate + dense core) linear algebra can
be checked against oracle/osqp_ref.c on a machine without a GPU.
"""
import ctypes as C

import numpy as np
import scipy.sparse as sp

INFTY, MIN_S, MAX_S = 1e30, 1e-4, 1e4
PLAN_FIELDS = ["Rp", "Rj", "Rpos", "Fp", "Fi", "Fpos", "Pdiag", "elim_var", "core_var", "elim_of",
               "core_of", "e_ptr", "pair_core", "pair_elim", "cp_ptr", "cp_row", "cp_pa", "cp_pe",
               "a_ptr", "a_pair", "s_a", "s_b", "s_ppos", "sa_ptr", "sa_row", "sa_pa", "sa_pb",
               "ss_ptr", "ss_k1", "ss_k2", "ss_e"]


def get_plan(lib, n, m, Pp, Pi, Ap, Ai, allow_elim=1):
    ip = lambda a: a.ctypes.data_as(C.POINTER(C.c_int))
    sizes = np.zeros(16, dtype=np.int32)
    rc = lib.sco_debug_plan_build(n, m, ip(Pp), ip(Pi), ip(Ap), ip(Ai), allow_elim, ip(sizes))
    assert rc == 0, rc
    plan = dict(n=n, m=m, Pp=Pp, Pi=Pi, Ap=Ap, Ai=Ai, n_e=int(sizes[0]), n_c=int(sizes[1]),
                ncpl=int(sizes[2]), nS=int(sizes[3]))
    cap = 1 << 22
    buf = np.zeros(cap, dtype=np.int32)
    for f in PLAN_FIELDS:
        k = lib.sco_debug_plan_get(f.encode(), ip(buf), cap)
        assert k >= 0, (f, k)
        plan[f] = buf[:k].copy()
    return plan


def _limit(v):
    v = np.where(v < MIN_S, 1.0, v)
    return np.minimum(v, MAX_S)


def emulate(plan, Pval, q, Aval, l, u, w=None, rho=0.1, sigma=5e-10, alpha=1.6, eps_abs=1e-6,
            eps_rel=1e-9, max_iter=100000, check=25, scaling=10):
    n, m = plan["n"], plan["m"]
    Pp, Pi, Ap, Ai = plan["Pp"], plan["Pi"], plan["Ap"], plan["Ai"]
    Rp, Rj, Rpos, Fp, Fi, Fpos = (plan[k] for k in ("Rp", "Rj", "Rpos", "Fp", "Fi", "Fpos"))
    n_e, n_c, ncpl = plan["n_e"], plan["n_c"], plan["ncpl"]
    w = np.ones(m) if w is None else np.asarray(w, dtype=np.float64)
    Ps, As, qs = Pval.astype(float).copy(), Aval.astype(float).copy(), q.astype(float).copy()
    D, E, c = np.ones(n), np.ones(m), 1.0
    Pcol = np.repeat(np.arange(n), np.diff(Pp)); Acol = np.repeat(np.arange(n), np.diff(Ap))
    Fcol = np.repeat(np.arange(n), np.diff(Fp)); Rrow = np.repeat(np.arange(m), np.diff(Rp))

    def colmax_P():
        v = np.zeros(n)
        np.maximum.at(v, Fcol, np.abs(Ps[Fpos]))
        return v

    for _ in range(scaling):
        Dt = colmax_P()
        np.maximum.at(Dt, Acol, np.abs(As))
        Et = np.zeros(m); np.maximum.at(Et, Ai, np.abs(As))
        Dt = 1.0 / np.sqrt(_limit(Dt)); Et = 1.0 / np.sqrt(_limit(Et))
        Ps = (Ps * Dt[Pi]) * Dt[Pcol]; As = (As * Et[Ai]) * Dt[Acol]
        qs = qs * Dt; D = D * Dt; E = E * Et
        ct = colmax_P().sum() / n if n else 0.0
        ct = max(ct, float(_limit(np.array([np.abs(qs).max() if n else 0.0]))[0]))
        ct = 1.0 / float(_limit(np.array([ct]))[0])
        Ps = Ps * ct; qs = qs * ct; c *= ct
    ls = np.maximum(l, -INFTY) * E; us = np.minimum(u, INFTY) * E
    rho_v = np.where((ls < -INFTY * MIN_S) & (us > INFTY * MIN_S), 1e-6,
                     np.where(us - ls < 1e-4, 1e3 * rho, rho))
    rw = rho_v * w
    # K_EE^-1, coupling, S
    kinv = np.zeros(n_e)
    for e in range(n_e):
        ve = plan["elim_var"][e]
        v = sigma + (Ps[plan["Pdiag"][ve]] if plan["Pdiag"][ve] >= 0 else 0.0)
        for t in range(Ap[ve], Ap[ve + 1]):
            v += rw[Ai[t]] * As[t] * As[t]
        kinv[e] = 1.0 / v
    cpl = np.zeros(ncpl)
    for k in range(ncpl):
        sl = slice(plan["cp_ptr"][k], plan["cp_ptr"][k + 1])
        cpl[k] = np.sum(rw[plan["cp_row"][sl]] * As[plan["cp_pa"][sl]] * As[plan["cp_pe"][sl]])
    S = np.zeros((n_c, n_c))
    for idx in range(plan["nS"]):
        a, b = plan["s_a"][idx], plan["s_b"][idx]
        v = sigma if a == b else 0.0
        if plan["s_ppos"][idx] >= 0:
            v += Ps[plan["s_ppos"][idx]]
        sl = slice(plan["sa_ptr"][idx], plan["sa_ptr"][idx + 1])
        v += np.sum(rw[plan["sa_row"][sl]] * As[plan["sa_pa"][sl]] * As[plan["sa_pb"][sl]])
        sl = slice(plan["ss_ptr"][idx], plan["ss_ptr"][idx + 1])
        v -= np.sum(cpl[plan["ss_k1"][sl]] * cpl[plan["ss_k2"][sl]] * kinv[plan["ss_e"][sl]])
        S[a, b] = v; S[b, a] = v
    W = np.linalg.inv(S) if n_c else np.zeros((0, 0))
    # sparse helpers for the iteration
    A_csc = sp.csc_matrix((As, Ai, Ap), shape=(m, n))
    A_csr = A_csc.tocsr()
    Pfull = sp.csc_matrix((Ps[Fpos], Fi, Fp), shape=(n, n))
    elim_of, core_of = plan["elim_of"], plan["core_of"]
    core_var, elim_var = plan["core_var"], plan["elim_var"]
    pair_core, pair_elim = plan["pair_core"], plan["pair_elim"]
    x = np.zeros(n); z = np.zeros(m); y = np.zeros(m); t = np.zeros(m)
    status, it = 0, 0
    for it in range(1, max_iter + 1):
        rhs = A_csc.T @ t + sigma * x - qs
        ge = rhs[elim_var] * kinv
        r = rhs[core_var].copy()
        np.subtract.at(r, pair_core, cpl * ge[pair_elim])
        xc = W @ r
        xt = np.zeros(n)
        xt[core_var] = xc
        acc = np.zeros(n_e)
        np.add.at(acc, pair_elim, cpl * xc[pair_core])
        xt[elim_var] = ge - kinv * acc
        zt = A_csr @ xt
        zr = alpha * zt + (1 - alpha) * z
        zn = np.minimum(np.maximum(zr + y / rho_v, ls), us)
        y = y + rho_v * (zr - zn); z = zn
        t = w * (rho_v * z - y)
        x = alpha * xt + (1 - alpha) * x
        if it % check == 0 or it == max_iter:
            ax = A_csr @ x
            pri = np.max(np.abs((ax - z) / E), initial=0.0)
            px = Pfull @ x; aty = A_csc.T @ (y * w)
            dua = np.max(np.abs((qs + px + aty) / D), initial=0.0) / c
            eps_p = eps_abs + eps_rel * max(np.max(np.abs(z / E), initial=0.0), np.max(np.abs(ax / E), initial=0.0))
            eps_d = eps_abs + eps_rel / c * max(np.max(np.abs(qs / D), initial=0.0),
                                                np.max(np.abs(aty / D), initial=0.0),
                                                np.max(np.abs(px / D), initial=0.0))
            if (m == 0 or pri < eps_p) and dua < eps_d:
                status = 1
                break
    if not status:
        status = -2
    return D * x, (E * y * w) / c, status, it
