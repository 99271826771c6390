"""Intended mode + adaptive rho, 7x20, B = 1024: time slice sweep."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
arrays, _ = af.make_batch(B)
ref = None
for sl in (6250, 3000, 1500, 750, 400, 200):
    p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0, max_sqp_iters=20, admm_slice=sl)
    st = _lib.default_qp_settings(adaptive_rho=1)
    t = time.time(); res = sb.solve_batch(arrays, params=p, qp_settings=st); dt = time.time() - t
    same = ref is None or (np.array_equal(ref.x, res.x) and np.array_equal(ref.admm_iters, res.admm_iters))
    ref = ref or res
    print("slice %d wall %.2fs sco_it/s %.0f success %.3f rounds %d identical %s stages %s" % (
        sl, dt, res.sqp_iters.sum() / dt, res.success.mean(), res.timing["rounds"], same,
        {k: round(v) for k, v in res.timing.items() if k.endswith("_ms")}), flush=True)
