import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sco_py_amd import _build
_build.LIB = os.environ["SCO_LIB_OVERRIDE"]
from sco_py_amd import _lib, batch as sb
from oracle import arm_family as af
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
arrays, _ = af.make_batch(B, d=12, T=50, K=10, O=10)
with sb.TrajOptBatch(B, 12, 50, 10, 10) as tb:
    tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
    tb.solve()
    lib = _lib.load(); lib.sco_debug_stamps_bt.argtypes = [C.POINTER(C.c_double)]
    out = np.zeros(256); print("rc", lib.sco_debug_stamps_bt(out.ctypes.data_as(C.POINTER(C.c_double))))
    st = out.reshape(16, 16)
    names = ["(1) colsum", "bar", "fwd", "bar", "mid+bar", "bwd", "bar", "(Y) dense", "x upd", "bar", "check", "top", "(Y) generic", "-"]
    it = st[0, 15]
    print("problem 0, last launch: iterations", it, " timing:", tb.last_timing())
    print("cycles per iteration by wave (rows) and segment (cols):", names)
    np.set_printoptions(linewidth=220, precision=0, suppress=True)
    print(st[:8, :14] / it)
    print("sum per wave", (st[:8, :14].sum(axis=1) / it))
