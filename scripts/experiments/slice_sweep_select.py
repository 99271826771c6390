"""ms per 1024-problem step against admm_slice under round selection (one device)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sco_py_amd import workloads as af, _lib, batch as sb
dims = dict(d=7, T=20, K=5, O=2)
arrays, _ = af.make_batch(1024, first=0, **dims)
tb = sb.TrajOptBatch(1024, 7, 20, 5, 2)
tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
q = _lib.default_qp_settings()
tb.solve(_lib.default_sqp_params(), q)
for sl in (3125, 5000, 6250, 7500, 8350, 10000, 12500):
    p = _lib.default_sqp_params(admm_slice=sl)
    tb.solve(p, q); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): tb.solve(p, q)
    dt = (time.perf_counter() - t0) / 2
    tm = tb.last_timing()
    print("slice %5d: %.1f ms per step, admm %.1f ms, %d rounds" % (sl, 1e3 * dt, tm["admm_ms"], tm["rounds"]), flush=True)
