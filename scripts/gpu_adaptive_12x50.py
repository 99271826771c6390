"""12-DOF x 50 (BASELINE configs[4] shape), quirks off: fixed rho vs adaptive rho, B = 1 latency and a small batch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
for B in (1, 64):
    arrays, _ = af.make_batch(B, d=12, T=50, K=10, O=10)
    for ad in (1, 0):
        if B > 1 and ad == 0:
            continue                      # the fixed-rho batch takes minutes
        p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0, max_sqp_iters=20)
        st = _lib.default_qp_settings(adaptive_rho=ad)
        t = time.time(); res = sb.solve_batch(arrays, params=p, qp_settings=st, analytic_jac=True); dt = time.time() - t
        print("B=%d adaptive=%d wall %.2fs sco_it/s %.2f success %.3f admm iters/problem %.0f qp_solves mean %.1f rounds %d stages %s" % (
            B, ad, dt, res.sqp_iters.sum() / dt, res.success.mean(), res.admm_iters.mean(), res.qp_solves.mean(),
            res.timing["rounds"], {k: round(v) for k, v in res.timing.items() if k.endswith("_ms")}), flush=True)
