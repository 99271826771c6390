"""Golden vectors for joint limits (two more LEqExpr(AffExpr) blocks: theta <= hi, -theta <= -lo; they go straight
into every QP, prob.py:126-131, 317-346) recorded from the REFERENCE's own modules, with the same
stand-ins as make_golden.py:   python tests/golden/make_golden_jl.py  ->  tests/golden/trajopt_jl.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import arm_family as af           # noqa: E402
from jl_cases import CASES                   # noqa: E402


def main():
    mg.install_standins()
    mods = mg.import_reference()
    out = {}
    for prefix, kw, i in CASES:
        mg.pack(prefix, mg.run_trajopt(mods, af.make_problem(i, **kw)), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_jl.npz"), **out)
    print("trajopt_jl.npz", os.path.getsize(os.path.join(HERE, "trajopt_jl.npz")), "bytes")


if __name__ == "__main__":
    main()
