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

SMALL = dict(d=3, T=6, K=2, O=2)
KNOBS = dict(initial_penalty_coeff=10.0, max_merit_coeff_increases=3)
# (prefix, problem kwargs, problem index, Solver attribute overrides)
CASES = [("h%d_" % i, dict(SMALL, groups="halves"), i, None) for i in (0, 10, 13)] + \
        [("hk%d_" % i, dict(SMALL, groups="halves"), i, KNOBS) for i in (0, 10)] + \
        [("s%d_" % i, dict(SMALL, groups="split", reach=True), i, None) for i in (22, 35, 38, 58)] + \
        [("sk%d_" % i, dict(SMALL, groups="split", reach=True), i, KNOBS) for i in (0, 58)]


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
