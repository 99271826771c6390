"""Cases of tests/golden/trajopt_point.npz: (prefix, make_problem kwargs, problem index) -- the point-robot family."""
SMALL = dict(d=2, T=8, K=1, O=3, point=True)
CASES = [("p%d_" % i, dict(SMALL), i) for i in range(4)] + \
        [("pv%d_" % i, dict(SMALL, vel_limit=0.45), i) for i in range(2)] + \
        [("pb%d_" % i, dict(SMALL, joint_limit=0.15), i) for i in range(2)] + \
        [("p3_%d_" % i, dict(SMALL, d=3), i) for i in range(2)] + \
        [("pg%d_" % i, dict(SMALL, groups="split"), i) for i in range(1)]
