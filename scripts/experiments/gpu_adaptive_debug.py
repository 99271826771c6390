"""Adaptive rho in the device loop vs the oracle with the same rule, 7x20: decision traces side by side."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import arm_family as af, sco_ref as sr
from sco_py_amd import _lib, batch as sb
np.set_printoptions(linewidth=200, precision=6, suppress=False)
nb = int(sys.argv[1]) if len(sys.argv) > 1 else 3
arrays, probs = af.make_batch(nb)
st = _lib.default_qp_settings(adaptive_rho=1)
res = sb.solve_batch(arrays, qp_settings=st)
fixed = sb.solve_batch(arrays)
print("timing adaptive", res.timing); print("timing fixed", fixed.timing)
for b in range(nb):
    ref = sr.penalty_sqp(sr.trajopt_flat(probs[b]), qp_settings=dict(adaptive_rho=1))
    print("problem", b, "device success", res.success[b], "oracle", ref.success, "fixed", fixed.success[b],
          "|dx|", np.abs(res.x[b] - ref.x).max(), "admm", res.admm_iters[b], ref.admm_iters, fixed.admm_iters[b])
    print(" device"); print(res.trace[b][:, [0, 1, 3, 4, 5, 6, 7]])
    print(" oracle"); print(ref.trace[:, [0, 1, 3, 4, 5, 6, 7]])
