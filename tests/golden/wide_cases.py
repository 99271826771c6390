"""Cases of tests/golden/trajopt_wide.npz (r04, the wider device template): (prefix, make_problem kwargs, problem index, analytic_jac).
w*: per-joint weights of the smoothing objective (a QuadExpr from a weighted difference matrix); ps*: program rows whose
parameters differ from timestep to timestep (every timestep's Expr closes over its own data); ac* / jk*: constraint blocks on
THREE and FOUR consecutive timesteps (acceleration and jerk limits as non-linear rows on a Variable of 3 / 4 timesteps); lr*:
GENERAL affine rows (LEqExpr / EqExpr on an AffExpr over the whole trajectory: joint couplings, a two-step limit, one equality);
mx*: TWO kinds of non-linear rows per timestep (the point's keep-out discs and program rows, two BoundExprs on one Variable);
aa*: an ACCELERATION term in the quadratic objective (QuadExpr from first and second difference matrices, per-joint weights)."""
ARM = dict(d=3, T=6, K=2, O=2)
P = dict(K=1, program=True)
CASES = [("w%d_" % i, dict(ARM, obj_weights=True), i, False) for i in range(2)] + \
        [("wr_", dict(ARM, obj_weights=True, reach=True), 1, False),
         ("wj_", dict(ARM, obj_weights=True, joint_limit=0.3, vel_limit=0.6), 0, False),
         ("wc_", dict(ARM, obj_weights=True, ee_cost_weight=1.0), 2, False),
         ("wp_", dict(d=2, T=8, K=1, O=3, point=True, obj_weights=True), 1, True)] + \
        [("ps%d_" % i, dict(P, d=2, T=8, per_step=True), i, False) for i in range(2)] + \
        [("pssw_", dict(P, d=2, T=8, variant="sweep", per_step=True), 0, False),
         ("psdy_", dict(P, d=3, T=8, variant="dynamics", per_step=True), 1, True),
         ("psat_", dict(P, d=2, T=8, variant="attract", per_step=True), 0, False),
         ("psw_", dict(P, d=2, T=8, per_step=True, obj_weights=True, groups="split"), 2, False)] + \
        [("ac%d_" % i, dict(P, d=2, T=8, variant="accel"), i, False) for i in range(2)] + \
        [("acs_", dict(P, d=2, T=8, variant="accel", per_step=True, obj_weights=True), 2, False),
         ("ja_ac_", dict(P, d=3, T=8, variant="accel"), 0, True),
         ("jk_", dict(P, d=2, T=9, variant="jerk"), 0, False), ("ja_jk_", dict(P, d=2, T=9, variant="jerk", groups="split"), 1, True)] + \
        [("lr%d_" % i, dict(ARM, lin_rows=True), i, False) for i in range(2)] + \
        [("lrv_", dict(ARM, lin_rows=True, vel_limit=0.6, joint_limit=0.3, obj_weights=True), 1, False),
         ("lrp_", dict(d=2, T=8, K=1, O=3, point=True, lin_rows=True), 0, True),
         ("lrg_", dict(P, d=2, T=8, lin_rows=True, per_step=True), 1, False)] + \
        [("mx%d_" % i, dict(P, d=2, T=8, circles=2), i, False) for i in range(2)] + \
        [("mxa_", dict(P, d=2, T=8, variant="attract", circles=1, per_step=True, groups="split"), 0, False),
         ("ja_mx_", dict(P, d=3, T=6, circles=2, lin_rows=True), 1, True)] + \
        [("aa%d_" % i, dict(ARM, acc_weights=True), i, False) for i in range(2)] + \
        [("aaw_", dict(ARM, acc_weights=True, obj_weights=True, vel_limit=0.6), 1, False),
         ("aap_", dict(P, d=2, T=8, acc_weights=True, per_step=True), 0, True)]
