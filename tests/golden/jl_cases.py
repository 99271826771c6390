"""Cases of tests/golden/trajopt_jl.npz: (prefix, make_problem kwargs, problem index)."""
SMALL = dict(d=3, T=6, K=2, O=2)
CASES = [("j%d_" % i, dict(SMALL, joint_limit=0.3), i) for i in range(3)] + \
        [("jt%d_" % i, dict(SMALL, joint_limit=0.05), i) for i in (0, 2)] + \
        [("jv%d_" % i, dict(SMALL, joint_limit=0.3, vel_limit=0.6), i) for i in range(2)] + \
        [("jr%d_" % i, dict(SMALL, joint_limit=0.3, reach=True), i) for i in range(2)]
