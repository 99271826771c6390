"""Host-side cost of one bench step outside the device time: solve() wall against its HIP-event total, fetch(), records."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sco_py_amd import workloads as af, _lib, batch as sb, dist as sd
dims = dict(d=7, T=20, K=5, O=2)
B = 1024
arrays, _ = af.make_batch(B, first=0, **dims)
tb = sb.TrajOptBatch(B, dims["d"], dims["T"], dims["K"], dims["O"], device=0)
tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
p = _lib.default_sqp_params(); q = _lib.default_qp_settings()
tb.solve(p, q); torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); tb.solve(p, q); t1 = time.perf_counter()
    res = tb.fetch(with_merit=True); t2 = time.perf_counter()
    rec = sd.pack_results(res.merit, res.max_violation, res.success, res.sqp_iters); t3 = time.perf_counter()
    tm = tb.last_timing(); t4 = time.perf_counter()
    print("solve %.2f ms (device total %.2f) fetch %.2f pack %.2f timing %.2f" % (
        1e3 * (t1 - t0), tm["total_ms"], 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
import ctypes as C
lib = _lib.load()
for rep in range(2):
    tb.solve(p, q)
    for k in range(4):
        t0 = time.perf_counter()
        x = np.zeros((B, tb.n_x)); success = np.zeros(B, dtype=np.int32); si = np.zeros(B, dtype=np.int32); qs_ = np.zeros(B, dtype=np.int32)
        admm = np.zeros(B, dtype=np.int64); merit = np.zeros(B); viol = np.zeros(B)
        t1 = time.perf_counter()
        lib.sco_sqp_fetch(tb._h, _lib.dptr(x), _lib.iptr(success), _lib.iptr(si), _lib.iptr(qs_), admm.ctypes.data_as(C.POINTER(C.c_longlong)), _lib.dptr(merit), _lib.dptr(viol))
        t2 = time.perf_counter()
        nc = np.zeros(B, dtype=np.uint32); lib.sco_sqp_fetch_groups(tb._h, nc.ctypes.data_as(C.POINTER(C.c_uint)))
        t3 = time.perf_counter()
        fl = np.zeros(B, dtype=np.int32); lib.sco_sqp_fetch_flags(tb._h, _lib.iptr(fl))
        t4 = time.perf_counter()
        print("rep %d call %d: alloc %.2f fetch %.2f groups %.2f flags %.2f ms" % (rep, k, 1e3*(t1-t0), 1e3*(t2-t1), 1e3*(t3-t2), 1e3*(t4-t3)), flush=True)
