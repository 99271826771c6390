"""Experiment: the batch split into independently advancing handles driven by host threads.
Do their kernels overlap on the device?  (kernel trace via rocprofv3 when run under it)"""
import sys, os, time, threading
import numpy as np
from types import SimpleNamespace
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, dist as _dist
from sco_py_amd.batch import TrajOptBatch
exec(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "groups_class.py")).read())
B = 1024
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4
arrays, _ = af.make_batch(B)
with TrajOptBatchGroups(B, 7, 20, 5, 2, groups=G) as tb:
    tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
    tb.solve()
    t = time.time(); tb.solve(); dt = time.time() - t
    res = tb.fetch()
    print("groups %d: wall %.3f s -> %.0f SCO it/s" % (G, dt, res.sqp_iters.sum() / dt))
