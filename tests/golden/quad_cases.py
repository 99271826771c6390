"""Cases of tests/golden/trajopt_quad.npz: (prefix, make_problem kwargs, problem index) -- the quadratic-row family."""
SMALL = dict(d=2, T=8, K=1, O=3, quadratic=True)
CASES = [("q%d_" % i, dict(SMALL), i) for i in range(4)] + \
        [("q3_%d_" % i, dict(SMALL, d=3, T=6, O=4), i) for i in range(2)] + \
        [("qv%d_" % i, dict(SMALL, vel_limit=0.5), i) for i in range(2)] + \
        [("qb%d_" % i, dict(SMALL, joint_limit=0.2), i) for i in range(1)] + \
        [("qg%d_" % i, dict(SMALL, groups="halves"), i) for i in range(1)]

# r03: quadratic rows with an EQUALITY row per timestep (the state stays on a sphere through start and goal) ->
# tests/golden/trajopt_quad2.npz.  (prefix, make_problem kwargs, problem index, analytic_jac)
CASES2 = [("qe%d_" % i, dict(SMALL, n_eq=1), i, False) for i in range(2)] + \
         [("qe3_", dict(SMALL, d=3, T=6, O=4, n_eq=1), 0, False), ("qea_", dict(SMALL, n_eq=1), 2, True)]
