"""Timing of BASELINE config 5 (12-DOF x 50) on the GPU: B = 1 and a small batch."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb
for B in [int(a) for a in sys.argv[1:]] or [1, 16]:
    arrays, _ = af.make_batch(B, d=12, T=50, K=10, O=10)
    t = time.time(); res = sb.solve_batch(arrays); dt = time.time() - t
    it = res.admm_iters
    print("B=%d wall %.2fs" % (B, dt), res.timing, "admm iters max %d sum %d" % (it.max(), it.sum()),
          "us/iter (critical path) %.2f" % (res.timing["admm_ms"] * 1e3 / it.max()), flush=True)
