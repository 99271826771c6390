"""Per-iteration time of the wavefront ADMM tier with phases switched off (dev script; SCO_WV_ABLATE: results wrong, timing only)."""
import os, sys
import numpy as np, scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.argv = [sys.argv[0], "noop"]
from sco_py_amd import _lib as L, _build
if os.environ.get("SCO_LIB_OVERRIDE"):
    _build.LIB = os.environ["SCO_LIB_OVERRIDE"]; print("library", _build.LIB)
os.environ["SCO_WV_MIN_PER_CU"] = "0"     # every launch on the wavefront tier (default: > 3.3 problems per CU)
import importlib.util
spec = importlib.util.spec_from_file_location("wvc", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_wv_check.py"))
src = open(spec.origin).read().split("quick = len(sys.argv)")[0]
ns = {"__file__": spec.origin, "__name__": "wvc"}
exec(compile(src, spec.origin, "exec"), ns)
penalty_qp = ns["penalty_qp"]

def timeit(B, ablate, iters=20000):
    rng = np.random.default_rng(7)
    T, d, r = 20, 7, 10
    probs = [penalty_qp(rng, T, d, r, pc=1e6) for _ in range(min(B, 64))]
    probs = [probs[i % len(probs)] for i in range(B)]
    P0, q0, A0, l0, u0 = probs[0]
    n = len(q0); m = len(l0)
    Pu = sp.triu(sp.csc_matrix(P0), format='csc'); Pu.sort_indices(); Ac = sp.csc_matrix(A0 != 0, dtype=float); Ac.sort_indices()
    Pp, Pi, Ap, Ai = Pu.indptr, Pu.indices, Ac.indptr, Ac.indices
    rows_of = np.asarray(Ai); cols_of = np.repeat(np.arange(n), np.diff(Ap)); prow = np.asarray(Pi); pcol = np.repeat(np.arange(n), np.diff(Pp))
    Pval = np.stack([p[0][prow, pcol] for p in probs]); Aval = np.stack([p[2][rows_of, cols_of] for p in probs])
    q = np.stack([p[1] for p in probs]); l = np.stack([p[3] for p in probs]); u = np.stack([p[4] for p in probs])
    os.environ["SCO_WV_ABLATE"] = str(ablate)
    qp = L.BatchedQP(B, n, m, Pp, Pi, Ap, Ai)
    qp.load(Pval, q, Aval, l, u, None)
    st = L.default_qp_settings(max_iter=iters, check_termination=0 if ablate in (1, 2, 3, 4) else 25)
    qp.solve(st); qp.solve(st)
    tm = qp.last_timing(); qp.close()
    return 1e3 * tm["admm_ms"] / iters

for B in ((1024,) if os.environ.get("SCO_LIB_OVERRIDE") else (64, 1024)):
    for ab, name in ((0, "full (with termination tests)"), (4, "full, no termination tests"), (64, "tests without infeasibility certificates"),
                     (16, "checked iterations, no test pass"), (1, "no sweeps"), (2, "no row passes")):
        print("B=%5d %-32s %.3f us per iteration" % (B, name, timeit(B, ab)))
