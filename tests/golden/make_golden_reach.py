"""Golden vectors for the reach variant (non-linear equality -> abs penalty) recorded from the
REFERENCE's own modules, with the same stand-ins as make_golden.py (see its header):
    python tests/golden/make_golden_reach.py   ->  tests/golden/trajopt_reach.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import arm_family as af           # noqa: E402


def main():
    mg.install_standins()
    mods = mg.import_reference()
    out = {}
    for i in range(3):
        pr = af.make_problem(i, d=3, T=6, K=2, O=2, reach=True)
        mg.pack("p%d_" % i, mg.run_trajopt(mods, pr), out)
    pr = af.make_problem(1, d=3, T=6, K=2, O=2, reach=True)
    mg.pack("p1a_", mg.run_trajopt(mods, pr, analytic_jac=True), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_reach.npz"), **out)
    print("trajopt_reach.npz", os.path.getsize(os.path.join(HERE, "trajopt_reach.npz")), "bytes")


if __name__ == "__main__":
    main()
