import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sco_py_amd import _build
_build.LIB = os.environ["SCO_LIB_OVERRIDE"]
from sco_py_amd import _lib, batch as sb
from sco_py_amd import workloads as af
arrays, _ = af.make_batch(256)
with sb.TrajOptBatch(256, 7, 20, 5, 2) as tb:
    tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
    tb.solve()
    lib = _lib.load(); lib.sco_debug_stamps.argtypes = [C.POINTER(C.c_double)]
    out = np.zeros(128); print("rc", lib.sco_debug_stamps(out.ctypes.data_as(C.POINTER(C.c_double))))
    st = out[:64].reshape(8, 8); ck = out[64:].reshape(8, 8)
    names = ["(1) rhs", "barrier1", "(3) W+dpp", "barrier3", "(Y) rows", "barrierY", "looptop"]
    it = st[0, 7]
    print("problem 0, last launch: iterations", it, " timing:", tb.last_timing())
    print("cycles per iteration by wave (rows) and segment (cols):", names)
    np.set_printoptions(linewidth=200, precision=0, suppress=True)
    print(st[:, :7] / it)
    print("sum per wave", (st[:, :7].sum(axis=1) / it))
    print("termination test, cycles per test by wave: [checked step + row dots, column dot, P x + norms, w4 values, block reduction, primal branch, dual branch + closing barrier], tests:", ck[0, 7])
    print(ck[:, :7] / np.maximum(ck[:, 7:8], 1))
    print("sum per test", (ck[:, :7].sum(axis=1) / np.maximum(ck[:, 7], 1)))
