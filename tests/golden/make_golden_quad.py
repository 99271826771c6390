"""Golden vectors for the quadratic-row family (SCO_FAM_STATE_QUADRATIC: per timestep one LEqExpr block on an Expr with
g[r](x) = 1/2 x' Q_r x + a_r' x + c_r -- keep-out ellipsoids (concave), a half-space, a keep-in ball --, the same lowering
as the arm's obstacle rows, prob.py:251-278) recorded from the REFERENCE's own modules, with the same stand-ins as
make_golden.py:   python tests/golden/make_golden_quad.py  ->  tests/golden/trajopt_quad.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import arm_family as af           # noqa: E402
from quad_cases import CASES, CASES2         # noqa: E402


def main():
    mg.install_standins()
    mods = mg.import_reference()
    out = {}
    for prefix, kw, i in CASES:
        mg.pack(prefix, mg.run_trajopt(mods, af.make_problem(i, **kw)), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_quad.npz"), **out)
    print("trajopt_quad.npz", os.path.getsize(os.path.join(HERE, "trajopt_quad.npz")), "bytes")
    out = {}                                  # r03: an equality row per timestep (EqExpr on a quadratic Expr)
    for prefix, kw, i, aj in CASES2:
        mg.pack(prefix, mg.run_trajopt(mods, af.make_problem(i, **kw), analytic_jac=aj), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_quad2.npz"), **out)
    print("trajopt_quad2.npz", os.path.getsize(os.path.join(HERE, "trajopt_quad2.npz")), "bytes")


if __name__ == "__main__":
    main()
