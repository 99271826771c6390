import sys, os
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import conftest as ct, trajopt_build as tb
from oracle import arm_family as af
from sco_py_amd.sco_osqp import osqp_utils as ou
M = ct.mirror_mods()
real = ou._solve_qp_batch
def logged(reqs):
    out = real(reqs)
    r = reqs[0]
    print("QP n=%d m=%d nnzA=%d w=%s iters=%d cache=%d handle=%s" % (r["A"].shape[1], r["A"].shape[0], r["A"].nnz, None if r["w"] is None else int(r["w"].max()), out[0][2], len(ou._HANDLE_CACHE), id(next(reversed(ou._HANDLE_CACHE.values())))))
    return out
ou._solve_qp_batch = logged
for warm in (False, True):
    ou.clear_handle_cache(); ou.WARM_START = warm
    print("warm", warm)
    pr = af.make_problem(3, d=3, T=6, K=2, O=2)
    prob, traj, _, _ = tb.build_prob(M, pr)
    s = M.Solver(); s.initial_penalty_coeff = 10.0; s.max_merit_coeff_increases = 3
    print(s.solve(prob, method="penalty_sqp"))

# direct: capture QP 2 and 3 and replay on one handle
cap = []
def grab(reqs):
    cap.append(reqs[0]); return real(reqs)
ou._solve_qp_batch = grab
ou.clear_handle_cache(); ou.WARM_START = False
pr = af.make_problem(3, d=3, T=6, K=2, O=2)
prob, traj, _, _ = tb.build_prob(M, pr)
s = M.Solver(); s.initial_penalty_coeff = 10.0; s.max_merit_coeff_increases = 3
s.solve(prob, method="penalty_sqp")
from sco_py_amd import _lib
r2, r3 = cap[1], cap[2]
P0, A0 = r2["P"], r2["A"]
print("same pattern:", np.array_equal(r2["A"].indices, r3["A"].indices), np.array_equal(r2["A"].indptr, r3["A"].indptr))
qp = _lib.BatchedQP(1, A0.shape[1], A0.shape[0], P0.indptr, P0.indices, A0.indptr, A0.indices)
print("info", qp.info())
for warm in (0, 1):
    for r in (r2, r3):
        qp.load(r["P"].data[None], r["q"][None], r["A"].data[None], r["l"][None], r["u"][None], None if r["w"] is None else r["w"][None])
        st = _lib.default_qp_settings(warm_start=warm, eps_abs=r["settings"][0], eps_rel=r["settings"][1])
        x, y, stt, it, res = qp.solve(st)
        print("warm", warm, "iters", it[0], "status", stt[0], "x checksum %.17g" % float(np.sum(x[0] * np.arange(1, x.shape[1] + 1))), "resid", res[0])
qp.close()
