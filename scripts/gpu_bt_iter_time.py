"""Per-iteration time of the structured global-memory kernel (12-DOF x 50) against the number of problems in flight:
the same problem replicated B times, first penalty QP capped at 4000 ADMM iterations."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sco_py_amd import _build
if os.environ.get("SCO_LIB_OVERRIDE"):
    _build.LIB = os.environ["SCO_LIB_OVERRIDE"]; print("library", _build.LIB)
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
one, _ = af.make_batch(1, d=12, T=50, K=10, O=10)
IT = 4000
for B in [int(v) for v in os.environ.get("BS", "1,8,16,32,64,128,256").split(",")]:
    arrays = dict(one); arrays["B"] = B
    for k in ("x0", "start", "goal", "link_len", "obstacles"):
        arrays[k] = np.repeat(one[k], B, axis=0)
    p = _lib.default_sqp_params(max_sqp_iters=1)
    st = _lib.default_qp_settings(max_iter=IT)
    res = sb.solve_batch(arrays, params=p, qp_settings=st)
    tm = res.timing
    its = res.admm_iters[0]
    print("B=%3d admm iterations/problem %d admm %.1f ms -> %.1f us per iteration; footprint ~%.0f MB" % (
        B, its, tm["admm_ms"], 1e3 * tm["admm_ms"] / max(its, 1), B * 1.49), flush=True)
