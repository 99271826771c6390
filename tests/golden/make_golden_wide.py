"""Golden vectors for the r04 extensions of the device template -- weighted smoothing objectives and per-timestep program
parameters -- recorded from the REFERENCE's own modules, with the same stand-ins as make_golden.py:
    python tests/golden/make_golden_wide.py  ->  tests/golden/trajopt_wide.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import arm_family as af           # noqa: E402
from wide_cases import CASES                 # noqa: E402


def main():
    mg.install_standins()
    mods = mg.import_reference()
    out = {}
    for prefix, kw, i, aj in CASES:
        mg.pack(prefix, mg.run_trajopt(mods, af.make_problem(i, **kw), analytic_jac=aj), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_wide.npz"), **out)
    print("trajopt_wide.npz", os.path.getsize(os.path.join(HERE, "trajopt_wide.npz")), "bytes")


if __name__ == "__main__":
    main()
