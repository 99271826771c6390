"""Cases of tests/golden/trajopt_prog.npz: (prefix, make_problem kwargs, problem index) -- the program family."""
SMALL = dict(d=2, T=8, K=1, program=True)
CASES = [("g%d_" % i, dict(SMALL), i) for i in range(4)] + \
        [("g3_%d_" % i, dict(SMALL, d=3, T=6), i) for i in range(2)] + \
        [("gv%d_" % i, dict(SMALL, vel_limit=0.5), i) for i in range(1)] + \
        [("gb%d_" % i, dict(SMALL, joint_limit=0.25, groups="split"), i) for i in range(1)]
