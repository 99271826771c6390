"""7x20, B = 1024: fixed rho vs adaptive rho (opt-in), parity mode and intended mode (quirks off, at most 20 QPs)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
arrays, _ = af.make_batch(B)
for mode, pk in (("parity", {}), ("intended", dict(compound_penalty=0, duplicate_rows=0, max_sqp_iters=20))):
    for ad, warm in ((0, 0), (1, 0)) + (((0, 1), (1, 1)) if mode == "intended" else ()):
        p = _lib.default_sqp_params(warm_start_qps=warm, **pk)
        st = _lib.default_qp_settings(adaptive_rho=ad)
        t = time.time(); res = sb.solve_batch(arrays, params=p, qp_settings=st); dt = time.time() - t
        print("%s warm=%d adaptive=%d wall %.2fs sco_it/s %.0f success %.3f admm iters/problem %.0f qp_solves mean %.1f "
              "merit median %.4f rounds %d stages %s" % (
                  mode, warm, ad, dt, res.sqp_iters.sum() / dt, res.success.mean(), res.admm_iters.mean(), res.qp_solves.mean(),
                  np.median(res.merit), res.timing["rounds"],
                  {k: round(v) for k, v in res.timing.items() if k.endswith("_ms")}), flush=True)
