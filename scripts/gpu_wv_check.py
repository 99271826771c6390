"""Wavefront ADMM tier (csrc/sco_admm_wv.hip) against the oracle and against the row-local tier on penalty QPs (dev script).

    python scripts/gpu_wv_check.py [quick]
"""
import ctypes as C, os, sys, time
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import osqp_ref as o
from sco_py_amd import _lib as L

os.environ["SCO_WV_MIN_PER_CU"] = "0"     # every launch on the wavefront tier (default: > 3.3 problems per CU)
lib = L.load()
lib.sco_debug_qp_tiers.restype = C.c_int; lib.sco_debug_qp_tiers.argtypes = [C.c_void_p]


def penalty_qp(rng, T, d, r, pc=10.0, pins=2):
    nx = T * d; ns = T * r; n = nx + ns
    Q = np.zeros((n, n))
    for t in range(T - 1):
        for j in range(d):
            a, b = t * d + j, (t + 1) * d + j
            Q[a, a] += 2; Q[b, b] += 2; Q[a, b] -= 2; Q[b, a] -= 2
    x0 = rng.standard_normal(nx) * 0.3
    rows = []; lo = []; hi = []
    for j in range(d):
        e = np.zeros(n); e[j] = 1; rows.append(e); v = x0[j] + 0.1 * rng.standard_normal(); lo.append(v); hi.append(v)
    if pins > 1:
        for j in range(d):
            e = np.zeros(n); e[(T - 1) * d + j] = 1; rows.append(e); v = x0[(T - 1) * d + j] + 0.1 * rng.standard_normal(); lo.append(v); hi.append(v)
    for t in range(T):
        for k in range(r):
            e = np.zeros(n); e[t * d:(t + 1) * d] = rng.standard_normal(d); e[nx + t * r + k] = -1
            rows.append(e); lo.append(-np.inf); hi.append(rng.standard_normal())
    for j in range(n):
        e = np.zeros(n); e[j] = 1; rows.append(e)
        if j < nx: lo.append(x0[j] - 1); hi.append(x0[j] + 1)
        else: lo.append(0.0); hi.append(np.inf)
    q = np.zeros(n); q[nx:] = pc
    return Q, q, np.array(rows), np.array(lo), np.array(hi)


def run(T, d, r, B, seed, weights=False, pc=10.0, n_oracle=8, settings=None, pins=2):
    rng = np.random.default_rng(seed)
    probs = [penalty_qp(rng, T, d, r, pc=pc, pins=pins) for _ in range(B)]
    P0, q0, A0, l0, u0 = probs[0]
    n = len(q0); m = len(l0)
    Pu = sp.triu(sp.csc_matrix(P0), format='csc'); Pu.sort_indices(); Ac = sp.csc_matrix(A0 != 0, dtype=float); Ac.sort_indices()
    Pp, Pi, Ap, Ai = Pu.indptr, Pu.indices, Ac.indptr, Ac.indices
    rows_of = np.asarray(Ai); cols_of = np.repeat(np.arange(n), np.diff(Ap))
    prow = np.asarray(Pi); pcol = np.repeat(np.arange(n), np.diff(Pp))
    Pval = np.stack([p[0][prow, pcol] for p in probs]); Aval = np.stack([p[2][rows_of, cols_of] for p in probs])
    q = np.stack([p[1] for p in probs]); l = np.stack([p[3] for p in probs]); u = np.stack([p[4] for p in probs])
    w = None
    if weights:
        w = np.ones((B, m), dtype=np.int32); w[:, pins * d:pins * d + T * r] = rng.integers(1, 5, size=(B, 1))
    res = {}
    for tier in ("wv", "rl"):
        if tier == "rl": os.environ["SCO_QP_NO_WV"] = "1"
        else: os.environ.pop("SCO_QP_NO_WV", None)
        qp = L.BatchedQP(B, n, m, Pp, Pi, Ap, Ai)
        tiers = lib.sco_debug_qp_tiers(qp._h)
        qp.load(Pval, q, Aval, l, u, w)
        st = settings if settings is not None else L.default_qp_settings()
        qp.solve(st)                                  # warm-up launch (code load)
        t = time.time(); x, y, stt, it, rs = qp.solve(st); dt = time.time() - t
        tm = qp.last_timing()
        res[tier] = (x, y, stt, it)
        print("  %s tiers=%d: admm %.2f ms, setup %.2f ms, iters mean %.0f max %d -> %.3f us per iteration (longest problem)"
              % (tier, tiers, tm["admm_ms"], tm["setup_ms"], it.mean(), it.max(), 1e3 * tm["admm_ms"] / max(it.max(), 1)))
        qp.close()
    os.environ.pop("SCO_QP_NO_WV", None)
    xw, yw, sw, iw = res["wv"]; xr, yr, s_r, ir = res["rl"]
    print("  wv vs rl: status equal %s, iters equal %s (%d differ), max|dx| %.2e, max|dy| %.2e"
          % (np.array_equal(sw, s_r), np.array_equal(iw, ir), int((iw != ir).sum()), np.abs(xw - xr).max(), np.abs(yw - yr).max()))
    bad = 0; worst = 0.0
    for b in range(min(B, n_oracle)):
        ro = o.solve(probs[b][0], probs[b][1], probs[b][2], probs[b][3], probs[b][4], w=None if w is None else w[b],
                     **({} if settings is None else dict(max_iter=settings.max_iter)))
        worst = max(worst, np.abs(xw[b] - ro.x).max())
        if sw[b] != ro.info.status_val or iw[b] != ro.info.iter: bad += 1; print("   oracle mismatch b", b, sw[b], ro.info.status_val, iw[b], ro.info.iter)
    print("  wv vs oracle (%d problems): max|dx| %.2e, status/iteration mismatches %d" % (min(B, n_oracle), worst, bad))
    return bad == 0 and np.array_equal(sw, s_r)


quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
ok = True
print("3x6 r=4 B=8"); ok &= run(6, 3, 4, 8, 1)
print("3x6 r=4 B=8 weights, one pin"); ok &= run(6, 3, 4, 8, 2, weights=True, pins=1)
print("2x8 r=3 B=8"); ok &= run(8, 2, 3, 8, 3)
print("5x9 r=6 B=8"); ok &= run(9, 5, 6, 8, 4)
print("7x20 r=10 B=64 weights"); ok &= run(20, 7, 10, 64, 5, weights=True)
print("7x20 r=10 B=64 stiff (pc 1e6: runs to max_iter)"); ok &= run(20, 7, 10, 64, 6, pc=1e6, n_oracle=2, settings=L.default_qp_settings(max_iter=20000))
if not quick:
    print("7x20 r=10 B=1024 stiff"); ok &= run(20, 7, 10, 1024, 7, pc=1e6, n_oracle=1, settings=L.default_qp_settings(max_iter=20000))
    print("7x20 r=10 B=4096 stiff"); ok &= run(20, 7, 10, 4096, 8, pc=1e6, n_oracle=0, settings=L.default_qp_settings(max_iter=20000))
print("ALL OK" if ok else "MISMATCH")
