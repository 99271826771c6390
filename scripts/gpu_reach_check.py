"""Timing of the reach family (SCO_FAM_ARM_REACH) at 7-DOF x 20, B = 1024."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for reach in (False, True):
    arrays, _ = af.make_batch(B, reach=reach)
    res = sb.solve_batch(arrays)
    t = time.time(); res = sb.solve_batch(arrays); dt = time.time() - t
    it = res.admm_iters
    print("reach=%s B=%d wall %.2fs sco_it/s %.0f" % (reach, B, dt, res.sqp_iters.sum() / dt), res.timing,
          "admm iters mean %.0f" % it.mean(), "us per problem-iteration at 256 CUs %.2f" % (res.timing["admm_ms"] * 1e3 * 256 / it.sum()), flush=True)
