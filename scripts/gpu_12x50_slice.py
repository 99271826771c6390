"""12-DOF x 50 step at batch B for several ADMM time slices (scheduling only): ms per step and problem-iterations per second."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sco_py_amd import workloads as af
from sco_py_amd import batch as sb, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
dims = dict(d=12, T=50, K=10, O=10)
arrays, _ = af.make_batch(B, **dims)
with sb.TrajOptBatch(B, 12, 50, 10, 10) as tb:
    tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
    for sl in [int(a) for a in sys.argv[2:]] or [6250]:
        t = time.time(); tb.solve(_lib.default_sqp_params(admm_slice=sl)); dt = time.time() - t
        r = tb.fetch(); tm = tb.last_timing()
        print("B=%d slice %6d: %.2f s per step, %.3f M problem-iterations/s, %d launches, %.2f SCO it/s" % (
            B, sl, dt, r.admm_iters.sum() / (tm["admm_ms"] * 1e-3) / 1e6, tm["launches"], r.sqp_iters.sum() / dt), flush=True)
