"""Host-path latency of one QP through the per-problem seam (handle creation vs load vs solve)."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import conftest as ct, trajopt_build as tb
from oracle import arm_family as af
from sco_py_amd import _lib
from sco_py_amd.sco_osqp import osqp_utils
log = []
real = osqp_utils._solve_qp_batch
def timed(reqs):
    t = time.perf_counter(); out = real(reqs); log.append((time.perf_counter() - t, out[0][2])); return out
osqp_utils._solve_qp_batch = timed
mods = ct.mirror_mods()
for rep in range(2):
    del log[:]
    prob, traj, _, _ = tb.build_prob(mods, af.make_problem(0), analytic_jac=True)
    t = time.perf_counter(); ok = mods.Solver().solve(prob, method="penalty_sqp"); dt = time.perf_counter() - t
    print("Solver.solve %.1f ms, QPs:" % (dt * 1e3), [("%.1f ms" % (a * 1e3), it) for a, it in log], "host rest %.1f ms" % ((dt - sum(a for a, _ in log)) * 1e3))
# breakdown of one QP call
r = None
def grab(reqs):
    global r; r = reqs; return real(reqs)
osqp_utils._solve_qp_batch = grab
prob, traj, _, _ = tb.build_prob(mods, af.make_problem(0), analytic_jac=True)
prob.convexify(); prob.update_obj(1e3); prob.save(); prob.add_trust_region(1.0); prob.optimize()
q = r[0]; P0, A0 = q["P"], q["A"]; n, m = A0.shape[1], A0.shape[0]
for rep in range(3):
    t0 = time.perf_counter(); qp = _lib.BatchedQP(1, n, m, P0.indptr, P0.indices, A0.indptr, A0.indices)
    t1 = time.perf_counter(); qp.load(P0.data[None], q["q"][None], A0.data[None], q["l"][None], q["u"][None], None)
    t2 = time.perf_counter(); st = _lib.default_qp_settings(max_iter=50); qp.solve(st)
    t3 = time.perf_counter(); qp.close(); t4 = time.perf_counter()
    print("create %.2f ms, load %.2f ms, solve(50 it) %.2f ms, close %.2f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3))
