"""Golden vectors for the program family (SCO_FAM_STATE_PROGRAM: per timestep one LEqExpr block on an Expr(f) whose f is a
compiled closed-form row program run as a NumPy callable -- rippled discs, a wavy wall, an exponential bump; numeric
Jacobians as the reference's default) recorded from the REFERENCE's own modules, with the same stand-ins as make_golden.py:
    python tests/golden/make_golden_prog.py  ->  tests/golden/trajopt_prog.npz, trajopt_prog2.npz
trajopt_prog2.npz (r03): blocks of two timesteps (swept-volume keep-outs, a unicycle dynamics EQUALITY), an equality row on
one timestep, objective programs, and runs in which the reference's Expr gets the forward-mode ``grad`` of the rows.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import arm_family as af           # noqa: E402
from prog_cases import CASES, CASES2         # noqa: E402


def main():
    mg.install_standins()
    mods = mg.import_reference()
    out = {}
    for prefix, kw, i in CASES:
        mg.pack(prefix, mg.run_trajopt(mods, af.make_problem(i, **kw)), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_prog.npz"), **out)
    print("trajopt_prog.npz", os.path.getsize(os.path.join(HERE, "trajopt_prog.npz")), "bytes")
    out = {}
    for prefix, kw, i, aj in CASES2:
        mg.pack(prefix, mg.run_trajopt(mods, af.make_problem(i, **kw), analytic_jac=aj), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_prog2.npz"), **out)
    print("trajopt_prog2.npz", os.path.getsize(os.path.join(HERE, "trajopt_prog2.npz")), "bytes")


if __name__ == "__main__":
    main()
