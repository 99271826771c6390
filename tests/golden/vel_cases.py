"""Cases of tests/golden/trajopt_vel.npz: (prefix, make_problem kwargs, problem index)."""
SMALL = dict(d=3, T=6, K=2, O=2)
CASES = [("v%d_" % i, dict(SMALL, vel_limit=0.6), i) for i in range(3)] + \
        [("t%d_" % i, dict(SMALL, vel_limit=0.25), i) for i in (1,)] + \
        [("vr%d_" % i, dict(SMALL, vel_limit=0.6, reach=True), i) for i in range(2)] + \
        [("x0_", dict(SMALL, vel_limit=0.05), 0), ("x1_", dict(SMALL, vel_limit=0.15), 1)]   # infeasible pins + limits
