"""Does the GPU overlap the lock-step rounds of independent handles?  Solves the 1024-problem bench batch as one
handle and as G handles of 1024/G problems driven from G host threads (ctypes releases the GIL), same device."""
import os, sys, time, threading
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from sco_py_amd import workloads as af, _lib, batch as sb

dims = dict(d=7, T=20, K=5, O=2)
B = int(os.environ.get("B", "1024"))


def make(G):
    hs = []
    for g in range(G):
        lo, hi = g * B // G, (g + 1) * B // G
        arrays, _ = af.make_batch(hi - lo, first=lo, **dims)
        tb = sb.TrajOptBatch(hi - lo, dims["d"], dims["T"], dims["K"], dims["O"], device=0)
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        hs.append(tb)
    return hs


def step(hs):
    p = _lib.default_sqp_params(); q = _lib.default_qp_settings()
    th = [threading.Thread(target=tb.solve, args=(p, q)) for tb in hs]
    for t in th: t.start()
    for t in th: t.join()


for G in [int(g) for g in os.environ.get("GS", "1,2,4,8").split(",")]:
    hs = make(G)
    step(hs); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(2): step(hs)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 2
    it = sum(int(tb.fetch(with_merit=False).sqp_iters.sum()) for tb in hs)
    print("G=%d  %.1f ms per step  %.0f SCO it/s  admm_ms per handle %s" % (
        G, 1e3 * dt, it / dt, [round(tb.last_timing()["admm_ms"], 1) for tb in hs]), flush=True)
    for tb in hs: tb.close()
