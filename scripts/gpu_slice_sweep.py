"""Headline workload: SCO it/s as a function of the ADMM slice length (scheduling only)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
arrays, _ = af.make_batch(B)
with sb.TrajOptBatch(B, 7, 20, 5, 2) as tb:
    tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
    ref = None
    for sl in (-1, 50000, 25000, 12500, 6250, 5000, 2500, 1250):
        p = _lib.default_sqp_params(admm_slice=sl)
        tb.solve(p)
        t = time.time(); tb.solve(p); tb.solve(p); dt = (time.time() - t) / 2
        res = tb.fetch(with_merit=False)
        if ref is None: ref = res
        same = np.array_equal(res.x, ref.x) and np.array_equal(res.admm_iters, ref.admm_iters)
        print("slice %6d: %.1f ms/step, %.0f SCO it/s, identical to unsliced: %s" % (sl, dt * 1e3, res.sqp_iters.sum() / dt, same), flush=True)
