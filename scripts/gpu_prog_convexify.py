"""Convexify stage of the program family with finite-difference and with forward-mode Jacobians (r02 verdict item 4):
HIP-event stage times of one solve, B problems of a variant (scripts/../sco_py_amd/workloads.py:variant_program)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
for variant, d, T in (("dynamics", 3, 20), ("sweep", 2, 20), (None, 2, 20)):
    arrays, _ = af.make_batch(B, d=d, T=T, K=1, program=True, variant=variant)
    for aj in (False, True):
        sb.solve_batch(arrays, analytic_jac=aj)
        res = sb.solve_batch(arrays, analytic_jac=aj)
        tm = res.timing
        print("%-9s B=%d T=%d  %-14s convexify %7.2f ms  qp_setup %6.2f  admm %8.1f  total %8.1f  (%d SQP iterations, %d rounds)" % (
            variant or "corridor", B, T, "forward-mode" if aj else "central diff.", tm["convexify_ms"], tm["qp_setup_ms"], tm["admm_ms"],
            tm["total_ms"], int(res.sqp_iters.sum()), tm["rounds"]), flush=True)
