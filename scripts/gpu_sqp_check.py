"""Dev check of the device SQP loop against the flat oracle (not a test)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af, sco_ref as sr
from sco_py_amd import batch as sb, _lib

def run(B, first=0, params_kw=None, **kw):
    arrays, probs = af.make_batch(B, first=first, **kw)
    p = _lib.default_sqp_params(**(params_kw or {}))
    t = time.time(); res = sb.solve_batch(arrays, params=p); dt = time.time() - t
    print("B", B, kw, params_kw, "wall %.3fs" % dt, res.timing)
    worst = 0.0
    for b in range(min(B, 6)):
        okw = {}
        if params_kw:
            okw = {k: bool(v) if k in ("compound_penalty", "duplicate_rows") else v for k, v in params_kw.items()}
        o = sr.penalty_sqp(sr.trajopt_flat(probs[b]), sr.SolverParams(**okw), emulate_memo=False)
        tr = res.trace[b]
        same_kinds = tr.shape == o.trace.shape and np.array_equal(tr[:, 0], o.trace[:, 0])
        dx = np.abs(res.x[b] - o.x).max(); worst = max(worst, dx)
        print("  b", b, "dx %.2e" % dx, "success", res.success[b], o.success, "sqp", res.sqp_iters[b], o.sqp_iters,
              "qp", res.qp_solves[b], o.qp_solves, "admm", res.admm_iters[b], o.admm_iters, "kinds", same_kinds,
              "viol %.2e %.2e" % (res.max_violation[b], o.max_violation))
        if not same_kinds or dx > 1e-6:
            print("   gpu trace\n", tr[:, [0, 1, 2, 3, 4, 6, 7]])
            print("   ora trace\n", o.trace[:, [0, 1, 2, 3, 4, 6, 7]])
        else:
            print("   max merit diff %.2e" % np.abs(tr[:, 1:4] - o.trace[:, 1:4]).max())
    return worst

run(4, d=3, T=6, K=2, O=2)
run(4, d=3, T=6, K=2, O=2, params_kw=dict(compound_penalty=0, duplicate_rows=0))
run(4, d=3, T=6, K=2, O=2, params_kw=dict(initial_penalty_coeff=10.0, max_merit_coeff_increases=3))
run(4)
run(64)
