"""Cases of tests/golden/trajopt_groups.npz: (prefix, make_problem kwargs, problem index, Solver attribute
overrides).  Shared by make_golden_groups.py (which runs the reference) and tests/test_golden.py."""
SMALL = dict(d=3, T=6, K=2, O=2)
KNOBS = dict(initial_penalty_coeff=10.0, max_merit_coeff_increases=3)
CASES = [("h%d_" % i, dict(SMALL, groups="halves"), i, None) for i in (0, 10, 13)] + \
        [("hk%d_" % i, dict(SMALL, groups="halves"), i, KNOBS) for i in (0, 10)] + \
        [("s%d_" % i, dict(SMALL, groups="split", reach=True), i, None) for i in (22, 35, 38, 58)] + \
        [("sk%d_" % i, dict(SMALL, groups="split", reach=True), i, KNOBS) for i in (0, 58)]
