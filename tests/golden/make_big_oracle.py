"""Oracle answer for one 12-DOF x 50-step problem (BASELINE configs[4]); takes minutes on
one CPU core, so the result is committed:  python tests/golden/make_big_oracle.py"""
import os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import arm_family as af, sco_ref as sr
pr = af.make_problem(0, d=12, T=50, K=10, O=10)
t = time.time()
out = sr.penalty_sqp(sr.trajopt_flat(pr), emulate_memo=True)
print("success", out.success, "sqp", out.sqp_iters, "qp", out.qp_solves, "admm", out.admm_iters, "%.1f s" % (time.time() - t))
np.savez_compressed(os.path.join(HERE, "trajopt_12x50_oracle.npz"), x=out.x, trace=out.trace, success=out.success,
                    max_violation=out.max_violation)
