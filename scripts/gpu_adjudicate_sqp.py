"""The two problems of r03's 2 / 512 parity misses through the device SQP loop, trace row by trace row beside the oracle's
(VERDICT r03 item 7): where the states of the two loops stand when the QP whose iteration count differs is built.

    python scripts/gpu_adjudicate_sqp.py >> gpurun_out/r04_adjudicate.txt
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af, sco_ref as sr
KW = dict(program=True, variant="dynamics", d=3, T=10, K=1)
refs = {i: sr.penalty_sqp(sr.trajopt_flat(af.make_problem(i, **KW), analytic_jac=True), None, emulate_memo=True) for i in (43, 57)}
from sco_py_amd import batch as sb
for i in (43, 57):
    arrays, _ = af.make_batch(1, first=i, **KW)
    for aj in (True, False):
        res = sb.solve_batch(arrays, analytic_jac=aj)
        g, tr = res.trace[0], refs[i].trace[:64]
        print("problem %d, device %s Jacobians vs oracle forward-mode: device %d QPs, oracle %d; success %s / %s; |dx| %.3e" % (
            i, "forward-mode" if aj else "finite-difference", len(g), len(tr), bool(res.success[0]), refs[i].success, np.abs(res.x[0] - refs[i].x).max()))
        for k in range(min(len(g), len(tr))):
            rel = lambda a, b: abs(a - b) / max(abs(b), 1e-300)
            print("   QP %2d kind %d/%d status %d/%d iters %6d/%6d   rel. difference of merit %.1e model %.1e new merit %.1e   penalty %.0e" % (
                k + 1, g[k, 0], tr[k, 0], g[k, 6], tr[k, 6], g[k, 7], tr[k, 7], rel(g[k, 1], tr[k, 1]), rel(g[k, 2], tr[k, 2]), rel(g[k, 3], tr[k, 3]), tr[k, 5]))
