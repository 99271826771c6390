"""Golden vectors for constraint groups (prob.add_cnt_expr(..., group_ids=...), solver.py:155-161,
209-235) recorded from the REFERENCE's own modules, with the same stand-ins as make_golden.py:
    python tests/golden/make_golden_groups.py   ->  tests/golden/trajopt_groups.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import arm_family as af           # noqa: E402

from group_cases import CASES            # noqa: E402


def main():
    mg.install_standins()
    mods = mg.import_reference()
    out = {}
    for prefix, kw, i, knobs in CASES:
        mg.pack(prefix, mg.run_trajopt(mods, af.make_problem(i, **kw), solver_attrs=knobs), out)
    np.savez_compressed(os.path.join(HERE, "trajopt_groups.npz"), **out)
    print("trajopt_groups.npz", os.path.getsize(os.path.join(HERE, "trajopt_groups.npz")), "bytes")


if __name__ == "__main__":
    main()
