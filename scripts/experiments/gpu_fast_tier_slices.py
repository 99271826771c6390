import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
arrays, _ = af.make_batch(1024, vel_limit=0.3, joint_limit=0.2)
for sl in (-1, 12500, 6250, 3000):
    p = _lib.default_sqp_params(admm_slice=sl)
    res = sb.solve_batch(arrays, params=p)
    t = time.time(); res = sb.solve_batch(arrays, params=p); dt = time.time() - t
    print("slice", sl, "wall %.2fs sco_it/s %.0f rounds %d" % (dt, res.sqp_iters.sum() / dt, res.timing["rounds"]), {k: round(v) for k, v in res.timing.items() if k.endswith("_ms")}, flush=True)
