"""Cases of tests/golden/trajopt_prog.npz: (prefix, make_problem kwargs, problem index) -- the program family."""
SMALL = dict(d=2, T=8, K=1, program=True)
CASES = [("g%d_" % i, dict(SMALL), i) for i in range(4)] + \
        [("g3_%d_" % i, dict(SMALL, d=3, T=6), i) for i in range(2)] + \
        [("gv%d_" % i, dict(SMALL, vel_limit=0.5), i) for i in range(1)] + \
        [("gb%d_" % i, dict(SMALL, joint_limit=0.25, groups="split"), i) for i in range(1)]

# r03: blocks of two timesteps (span 2), equality rows, objective programs, and runs in which the reference gets the
# forward-mode ``grad`` of the rows (Expr(f, grad)) -> tests/golden/trajopt_prog2.npz.
# (prefix, make_problem kwargs, problem index, analytic_jac)
V = dict(K=1, program=True)
CASES2 = [("sw%d_" % i, dict(V, d=2, T=8, variant="sweep"), i, False) for i in range(2)] + \
         [("sw3_", dict(V, d=3, T=6, variant="sweep"), 0, False)] + \
         [("dy%d_" % i, dict(V, d=3, T=8, variant="dynamics"), i, False) for i in range(2)] + \
         [("cu0_", dict(V, d=3, T=8, variant="curve"), 0, False)] + \
         [("at%d_" % i, dict(V, d=2, T=8, variant="attract"), i, False) for i in range(2)] + \
         [("swv_", dict(V, d=2, T=8, variant="sweep", vel_limit=0.6, groups="split"), 1, False)] + \
         [("ja_sw_", dict(V, d=2, T=8, variant="sweep"), 0, True), ("ja_dy_", dict(V, d=3, T=8, variant="dynamics"), 0, True),
          ("ja_co_", dict(V, d=2, T=8), 0, True)]
