"""7x20, B = 1024, parity mode: SCO it/s of the device families / flags (which ADMM tier each pattern lands on)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb, _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
import os as _os
ONLY = _os.environ.get("FAMILIES")
for name, kw in (("circles", {}), ("reach", dict(reach=True)), ("vel", dict(vel_limit=0.3)), ("jl", dict(joint_limit=0.2)),
                 ("vel+jl", dict(vel_limit=0.3, joint_limit=0.2)), ("reach+vel+jl", dict(reach=True, vel_limit=0.3, joint_limit=0.2)),
                 # r04, the wider template: general affine rows, weighted smoothing objective (any family), program parameters per timestep
                 ("circles+rows", dict(lin_rows=True)), ("circles+acceleration term", dict(acc_weights=True)), ("circles+weights", dict(obj_weights=True)), ("vel+weights", dict(vel_limit=0.3, obj_weights=True)),
                 ("program d=2 T=20", dict(d=2, T=20, K=1, program=True)), ("program+steps", dict(d=2, T=20, K=1, program=True, per_step=True)),
                 ("program+circles (two kinds)", dict(d=2, T=20, K=1, program=True, circles=3)),
                 ("sweep+steps (span 2)", dict(d=2, T=20, K=1, program=True, variant="sweep", per_step=True)),
                 ("attract+steps+weights", dict(d=2, T=20, K=1, program=True, variant="attract", per_step=True, obj_weights=True)),
                 # shapes other than 7 x 20 on the patterns the wavefront tier takes (which tier is faster where)
                 ("point d=2 T=20", dict(d=2, T=20, K=1, O=3, point=True)), ("quad d=3 T=12", dict(d=3, T=12, K=1, O=4, quadratic=True)),
                 ("arm 4x24", dict(d=4, T=24, K=3, O=2)), ("arm 7x12", dict(T=12)), ("arm 5x16", dict(d=5, T=16, K=4, O=2)), ("arm 3x6", dict(d=3, T=6, K=2, O=2))):
    if ONLY and name not in ONLY.split(','):
        continue
    arrays, _ = af.make_batch(B, **kw)
    res = sb.solve_batch(arrays)
    t = time.time(); res = sb.solve_batch(arrays); dt = time.time() - t
    tm = res.timing
    tier = "wavefront %d + other %d launches" % (tm.get("wv_launches", 0), tm.get("other_launches", 0))
    print("%-24s [%s] wall %.2fs sco_it/s %.0f success %.3f admm iters/problem %.0f -> %.2f us per problem-iteration (admm %.0f ms, setup %.0f ms)" % (
        name, tier, dt, res.sqp_iters.sum() / dt, res.success.mean(), res.admm_iters.mean(),
        1e3 * tm["admm_ms"] / (res.admm_iters.sum() / 256.0), tm["admm_ms"], tm["qp_setup_ms"]), flush=True)
