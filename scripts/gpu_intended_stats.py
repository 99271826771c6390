import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
arrays, _ = af.make_batch(1024)
p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0)
t = time.time(); res = sb.solve_batch(arrays, params=p); dt = time.time() - t
q = res.qp_solves
print("wall %.1fs" % dt, res.timing)
print("qp_solves: mean %.1f median %d p90 %d p99 %d max %d" % (q.mean(), np.median(q), np.percentile(q, 90), np.percentile(q, 99), q.max()))
print("admm iters per problem: mean %.0f max %d" % (res.admm_iters.mean(), res.admm_iters.max()))
print("success", res.success.mean(), "sqp_iters mean", res.sqp_iters.mean())
worst = int(np.argmax(q)); print("worst problem", worst, "trace kinds", res.trace[worst][:, 0].astype(int).tolist()[:80])
print("its", res.trace[worst][:, 7].astype(int).tolist()[:40])
