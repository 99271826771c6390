"""Intended mode (quirks off), 7x20, B = 1024, at most 20 QPs per problem: cold vs warm-started QPs."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
arrays, _ = af.make_batch(1024)
for warm in (0, 1):
    p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0, max_sqp_iters=20, warm_start_qps=warm)
    res = sb.solve_batch(arrays, params=p)
    t = time.time(); res = sb.solve_batch(arrays, params=p); dt = time.time() - t
    print("warm=%d wall %.2fs sco_it/s %.0f success %.3f admm iters/problem %.0f qp_solves mean %.1f merit median %.4f" % (
        warm, dt, res.sqp_iters.sum() / dt, res.success.mean(), res.admm_iters.mean(), res.qp_solves.mean(), np.median(res.merit)), flush=True)
