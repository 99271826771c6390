import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sco_py_amd import _build
_build.LIB = os.environ["SCO_LIB_OVERRIDE"]
from sco_py_amd import _lib, batch as sb
from oracle import arm_family as af
B = 4
arrays, _ = af.make_batch(B)
with sb.TrajOptBatch(B, 7, 20, 5, 2) as tb:
    tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"], arrays["point_frac"], arrays["obstacles"])
    tb.solve()
    lib = _lib.load(); lib.sco_debug_setup_stamps.argtypes = [C.POINTER(C.c_double)]
    out = np.zeros(16); print("rc", lib.sco_debug_setup_stamps(out.ctypes.data_as(C.POINTER(C.c_double))))
    names = ["load", "ruiz x10", "bounds/rho/publish", "Kee+cpl", "S assembly", "cholesky", "inverse", "W=M'M"]
    calls = out[15]
    print("setups of problem 0 (incl. the tiny projection QP):", calls, tb.last_timing())
    for k, nm in enumerate(names):
        print("%-22s %10.0f cycles per setup" % (nm, out[k] / calls))
    print("total %.0f" % (out[:8].sum() / calls))
