/*
 * sco_hip.h -- C ABI of libsco_hip.so, the MI355X (gfx950) implementation of the
 * sco_py `sco_osqp` hot path.
 *
 * The reference (Algorithmic-Alignment-Lab/sco_py) is pure Python and has no FFI
 * of its own: its boundary is the Python object API.  This library sits UNDER
 * that API.  Each entry point below names the reference interface it replaces
 * (paths relative to the reference tree).  Plain C, caller-owned host buffers,
 * int return codes, no pointer retained past a call except through a handle.
 * All arithmetic is float64.
 *
 * Two layers:
 *   sco_qp_*   batched QP solve, B problems sharing one sparsity pattern.
 *              Replaces osqp.OSQP().setup(...) / .solve() as called from
 *              sco_py/sco_osqp/osqp_utils.py:195-216 (one QP per call there).
 *   sco_sqp_*  the whole penalty-SQP loop for a batch of trajectory problems,
 *              device resident.  Replaces, per problem,
 *              Solver._penalty_sqp / _min_merit_fn (sco_py/sco_osqp/solver.py:62-253)
 *              together with the Prob methods it drives: find_closest_feasible_point
 *              (prob.py:369-412), convexify (:522-544), update_obj (:414-426),
 *              add_trust_region (:514-519), optimize (:146-205), get_value (:547-579),
 *              get_approx_value (:605-630), get_max_cnt_violation (:592-603),
 *              save/restore (:639-652).
 */
#ifndef SCO_HIP_H
#define SCO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- return codes ------------------------------------------------------- */
#define SCO_OK              0
#define SCO_ERR_ARG        -1   /* bad argument (null pointer, negative size, bad pattern) */
#define SCO_ERR_DEVICE     -2   /* HIP runtime error (see sco_last_error)                  */
#define SCO_ERR_NO_GPU     -3   /* no gfx950 device visible                                */
#define SCO_ERR_STATE      -4   /* call order violated (e.g. solve before load)            */
#define SCO_ERR_CAPACITY   -5   /* problem too large for the on-chip working set            */

/* ---- per-problem QP status: OSQP's status_val, which the reference tests at
 *      prob.py:197 (success iff status in {1, 2}) and osqp_utils.py:218 ---------- */
#define SCO_QP_SOLVED                 1
#define SCO_QP_SOLVED_INACCURATE      2
#define SCO_QP_MAX_ITER_REACHED      -2
#define SCO_QP_PRIMAL_INFEASIBLE     -3
#define SCO_QP_PRIMAL_INFEASIBLE_INACCURATE 3
#define SCO_QP_DUAL_INFEASIBLE       -4
#define SCO_QP_DUAL_INFEASIBLE_INACCURATE   4
#define SCO_QP_NON_CVX               -7
#define SCO_QP_UNSOLVED             -10

const char *sco_last_error(void);
int sco_version(void);
/* Number of usable gfx950 devices (0 is a valid answer; never initialises a context). */
int sco_device_count(int *count);

/* ---- QP layer ----------------------------------------------------------- */

/* Settings handed to osqp.OSQP().setup at osqp_utils.py:197-214 plus the OSQP
 * defaults the reference leaves untouched (alpha, scaling, check_termination,
 * infeasibility tolerances).  sco_qp_default_settings fills the values the
 * reference uses: osqp_utils.py:10-15 and OSQP 0.6 defaults. */
typedef struct sco_qp_settings {
  double rho;               /* osqp_utils.py:12  DEFAULT_RHO = 0.1       */
  double sigma;             /* osqp_utils.py:11  DEFAULT_SIGMA = 5e-10   */
  double alpha;             /* OSQP default 1.6                          */
  double eps_abs;           /* osqp_utils.py:14  1e-6                    */
  double eps_rel;           /* osqp_utils.py:15  1e-9                    */
  double eps_prim_inf;      /* OSQP default 1e-4                         */
  double eps_dual_inf;      /* OSQP default 1e-4                         */
  int max_iter;             /* osqp_utils.py:10  100000                  */
  int check_termination;    /* OSQP default 25                           */
  int scaling;              /* OSQP default 10 Ruiz passes               */
  int warm_start;           /* 0 (default, what the reference does: a new OSQP object per QP, cold start,
                               osqp_utils.py:195) | 1: start ADMM from the handle's previous solution
                               (x, y; z = A x), as OSQP's own warm start does.  NOT parity mode: iterates and
                               iteration counts change, the solution agrees to the QP tolerances.  Honoured by
                               the row-local tier (the one 7-DOF x 20 runs on) and the structured global-memory
                               tier (12-DOF x 50); the other tiers start cold.  */
  int adaptive_rho;         /* osqp_utils.py:13  DEFAULT_ADAPTIVE_RHO = False (solver.py:39 lets the caller turn it on).
                               1: OSQP's rho update -- every adaptive_rho_interval iterations, after the termination
                               test, rho <- rho sqrt(normalised primal / normalised dual residual) clipped to
                               [1e-6, 1e6], taken when it leaves [rho / tol, rho tol]; the reduced system is then
                               refactored and the solve resumes from its iterates (the solve is parked and resumed
                               around every update).  Runs on every on-chip tier and on the structured form of the
                               global-memory tier; its dense form answers SCO_ERR_CAPACITY.  Not part of parity mode.  */
  int adaptive_rho_interval;/* 0 = 4 x check_termination (OSQP's value when it does not time itself: 100)         */
  double adaptive_rho_tolerance; /* OSQP default 5                                                                  */
} sco_qp_settings;

void sco_qp_default_settings(sco_qp_settings *s);

typedef struct sco_qp sco_qp;

/* Create a solver for `batch` QPs
 *     min 1/2 x'Px + q'x   s.t.  l <= Ax <= u
 * that share one sparsity pattern: P (n x n) by its UPPER triangle in CSC form
 * (P_colptr[n+1], P_rowidx[nnzP]; the reference builds an upper-triangular P,
 * osqp_utils.py:153-163) and A (m x n) in CSC form (A_colptr[n+1],
 * A_rowidx[nnzA]).  Row indices inside a column must be strictly increasing.
 * The symbolic analysis (elimination set, dense core, index plans) happens here,
 * once. */
int sco_qp_create(int device, int batch, int n, int m,
                  const int *P_colptr, const int *P_rowidx,
                  const int *A_colptr, const int *A_rowidx,
                  sco_qp **out);
int sco_qp_destroy(sco_qp *qp);

/* Upload values, problem-major: P_val[batch][nnzP], q[batch][n], A_val[batch][nnzA],
 * l[batch][m], u[batch][m].  Infinite bounds may be +-inf or anything beyond
 * +-1e30 (OSQP's Python wrapper clamps at 1e30 the same way).
 * row_weight[batch][m] (may be NULL = all 1) says how many times each row is
 * present: the reference re-appends its penalty rows on every update_obj call
 * (prob.py:508-509), which this ABI represents as an integer multiplicity. */
int sco_qp_load(sco_qp *qp, const double *P_val, const double *q,
                const double *A_val, const double *l, const double *u,
                const int *row_weight);

/* Replace only l and u (trust-region retry: variable.py:43-45 changes nothing else). */
int sco_qp_set_bounds(sco_qp *qp, const double *l, const double *u);

/* Solve all problems of the batch from a cold start (the reference rebuilds the
 * OSQP object for every QP, osqp_utils.py:195, so x0 = y0 = 0).
 * Outputs (any may be NULL): x[batch][n], y[batch][m], status[batch] (SCO_QP_*),
 * iters[batch], resid[batch][2] = {primal, dual} residual at termination. */
int sco_qp_solve(sco_qp *qp, const sco_qp_settings *settings,
                 double *x, double *y, int *status, int *iters, double *resid);

/* Sizes chosen by the symbolic analysis (for tests and DESIGN.md):
 * info[0] = size of the eliminated set, info[1] = dense core order,
 * info[2] = LDS bytes of the ADMM kernel, info[3] = nnz of the coupling block. */
int sco_qp_info(const sco_qp *qp, int info[4]);

/* After a solve with adaptive_rho: the rho each problem ended with (rho[batch]) and how many times it changed
 * (updates[batch]); either pointer may be NULL.  (OSQP reports the same as info.rho_estimate / info.rho_updates.) */
int sco_qp_adaptive_info(sco_qp *qp, double *rho, int *updates);

/* Device time of the last sco_qp_solve, split by kernel, in milliseconds,
 * measured with HIP events on the library's own stream:
 * ms[0] = setup (scaling + factor), ms[1] = ADMM loop. */
int sco_qp_last_timing(const sco_qp *qp, double ms[2]);

/* ---- SQP layer ---------------------------------------------------------- */

/* Solver knobs, one-to-one with the attributes of sco_py.sco_osqp.solver.Solver
 * (solver.py:17-28). */
typedef struct sco_sqp_params {
  double improve_ratio_threshold;    /* 0.25  */
  double min_trust_region_size;      /* 1e-4  */
  double min_approx_improve;         /* 1e-8  */
  double trust_shrink_ratio;         /* 0.1   */
  double trust_expand_ratio;         /* 1.5   */
  double cnt_tolerance;              /* 1e-4  */
  double merit_coeff_increase_ratio; /* 10    */
  double initial_trust_region_size;  /* 1     */
  double initial_penalty_coeff;      /* 1e3   */
  int max_merit_coeff_increases;     /* 1     */
  /* Reference behaviours that look unintended but are load-bearing for result
   * parity.  1 = reproduce (default), 0 = the "intended" form. */
  int compound_penalty;     /* Q1: slack cost multiplied in place by penalty_coeff on
                               every update_obj (prob.py:424-426)                      */
  int duplicate_rows;       /* Q2: penalty rows appended again on every update_obj
                               (prob.py:508-509) -> row multiplicity k                 */
  int max_sqp_iters;        /* safety cap on QP solves per problem; the reference's
                               loops are unbounded (solver.py:126, 136), 0 = 10000     */
  int memoize_rounded;      /* Q3: Expr.eval and Eq/LEqExpr.convexify memoise on the point
                               rounded to 6 decimals (expr.py:13, 31-41, 323-332, 362-371): a
                               point within rounding of an earlier one reuses that point's
                               f values / affine model.  1 = reproduce (default)            */
  int warm_start_qps;       /* 0 (default, reference behaviour) | 1: every penalty QP of a problem starts from
                               the solution of its previous one (sco_qp_settings.warm_start); beyond parity   */
  int admm_slice;           /* scheduling only, results are bit-identical: ADMM iterations per device launch.
                               Problems whose QP ends inside a slice go on to their next QP while the others
                               continue (no waiting for the slowest QP of a round).  With more live problems
                               than CUs a round runs whole passes of the chip as a compact launch over the problems
                               with most in front of them, the others sit the round out.
                               0 = default (6250; 2000 with adaptive rho; off when the batch has at most one
                               problem per CU), < 0 = off (one launch per QP, lock-step rounds)              */
} sco_sqp_params;

void sco_sqp_default_params(sco_sqp_params *p);

/* Device-evaluable constraint families (the reference differentiates arbitrary
 * Python callables, expr.py:22-41; a GPU cannot). */
#define SCO_FAM_ARM_CIRCLES 1  /* planar serial arm, link points vs circular obstacles:
                                  g[k*O + o](theta) = r_o - || p_k(theta) - c_o ||  <= 0 */
#define SCO_FAM_ARM_REACH 2    /* the same, but the goal pin theta[horizon-1] = goal is replaced by the
                                  NON-LINEAR EQUALITY ee(theta[horizon-1]) = target (2 rows, end-effector
                                  position): EqExpr on an Expr, lowered to the abs penalty with two slack
                                  variables per row (prob.py:280-315); target via sco_sqp_load_target */
#define SCO_FAM_POINT_CIRCLES 3 /* a point robot in the plane instead of the arm: the state of a timestep is dof >= 2
                                  numbers, the first two its position, and g[o](x) = r_o - || x[0:2] - c_o || <= 0
                                  (n_points must be 1; link_len / point_link / point_frac of sco_sqp_load are not
                                  read).  Objective, pins and the linear-row flags below work as for the arm
                                  (velocity limits = longest step per axis, joint limits = workspace box); the
                                  objective-term flag SCO_FAM_FLAG_EE_COST is not available for it */
#define SCO_FAM_STATE_QUADRATIC 4 /* general quadratic rows on the state of a timestep (no kinematics): n_obstacles rows
                                  g[r](x) = 1/2 x' Q_r x + a_r' x + c_r <= 0 per timestep with per-problem coefficients
                                  (sco_sqp_load_quadratic), Q_r symmetric and of either sign -- keep-out ellipses, keep-in
                                  discs, half-planes, products of coordinates ...; dof <= 16, n_points must be 1; link_len /
                                  point_* / obstacles of sco_sqp_load are not read.  The last n_eq_rows rows may be equalities
                                  (EqExpr -> abs penalty, prob.py:280-315).  Flags as for SCO_FAM_POINT_CIRCLES */
#define SCO_FAM_STATE_PROGRAM 5 /* closed-form rows given as small postfix programs over the state of a constraint block and a
                                  per-problem parameter vector (sco_sqp_load_program) -- what the reference's Expr(f) is for
                                  any f one can write down with + - * / sin cos sqrt exp.  A block is `span` (1 .. 4)
                                  consecutive timesteps: block t binds the rows to the Variable (theta[t], .., theta[t+span-1])
                                  (the reference binds any Expr to any Variable, expr.py:413-437, prob.py:112-144; swept-volume
                                  and dynamics constraints live on two timesteps), state = their concatenation, span * dof <= 32,
                                  horizon - span + 1 blocks.  n_obstacles rows per block: the first n_obstacles - n_eq_rows are
                                  inequalities g[r](x, p) <= 0 (LEqExpr -> hinge penalty, prob.py:251-278), the last n_eq_rows
                                  equalities g[r](x, p) = 0 (EqExpr -> abs penalty with two slacks per row, prob.py:280-315).
                                  Jacobians by the device's central differences (analytic_jac = 0: the reference's default
                                  for an Expr without grad) or by forward-mode differentiation of the program itself
                                  (analytic_jac = 1: what a caller who supplies grad gets, expr.py:86-100).  n_points must be 1.
                                  Linear-row flags as for SCO_FAM_POINT_CIRCLES; SCO_FAM_FLAG_OBJ_PROGRAM below.
                                  Program words are pairs (op, arg): */
#define SCO_OP_END 0      /* end of a row's program: its value is the one number left on the stack */
#define SCO_OP_X 1        /* push x[arg] (state coordinate)        */
#define SCO_OP_P 2        /* push params[problem][arg]             */
#define SCO_OP_C 3        /* push consts[arg]                      */
#define SCO_OP_ADD 4      /* a b -> a + b                          */
#define SCO_OP_SUB 5      /* a b -> a - b                          */
#define SCO_OP_MUL 6      /* a b -> a * b                          */
#define SCO_OP_DIV 7      /* a b -> a / b                          */
#define SCO_OP_NEG 8      /* a -> -a                               */
#define SCO_OP_SIN 9
#define SCO_OP_COS 10
#define SCO_OP_SQRT 11
#define SCO_OP_EXP 12
#define SCO_OP_SQUARE 13  /* a -> a * a                            */
#define SCO_PROGRAM_STACK 16   /* deepest stack a row's program may need */

/* Structure of a batch of trajectory problems (shared by all `batch` problems):
 * variables theta[t][j], t < horizon, j < dof, flattened time-major (n_x = horizon*dof);
 * objective   sum_t || theta[t+1] - theta[t] ||^2      (a QuadExpr, expr.py:184-213)
 * linear      theta[0] = start, theta[horizon-1] = goal (EqExpr(AffExpr), prob.py:126-128)
 * nonlinear   one LEqExpr block of n_points*n_obstacles rows per timestep (family above),
 *             Jacobians by central finite differences on device. */
#define SCO_FAM_FLAG_VEL_LIMITS 16 /* OR-ed into `family`: joint-velocity limits
                                  |theta[t+1][j] - theta[t][j]| <= vmax as LINEAR inequality rows
                                  (LEqExpr on an AffExpr: they go straight into every QP, the projection
                                  QP included, prob.py:126-131, 317-346); vmax via sco_sqp_load_vel_limit */
#define SCO_FAM_FLAG_JOINT_LIMITS 32 /* OR-ed into `family`: joint limits lo_j <= theta[t][j] <= hi_j at every
                                  timestep as two more LINEAR inequality blocks (theta <= hi, then -theta <= -lo,
                                  after the velocity rows); lo, hi via sco_sqp_load_joint_limits */
#define SCO_FAM_FLAG_EE_COST 64      /* OR-ed into `family`: a NON-QUADRATIC objective term per timestep,
                                  weight * || ee(theta[t]) - target ||^2 (ee = end effector of the planar arm).
                                  The reference routes such an Expr to Prob._nonquad_obj_exprs (prob.py:88-104) and
                                  convexifies it to degree 2 on every SQP iteration: numeric Hessian, shifted by its
                                  smallest eigenvalue when that is negative, numeric gradient (expr.py:102-156,
                                  prob.py:532-534); the model goes into P and q (prob.py:348-367).  dof <= 16;
                                  weight, target via sco_sqp_load_ee_cost */

#define SCO_FAM_FLAG_OBJ_PROGRAM 128  /* OR-ed into SCO_FAM_STATE_PROGRAM (span 1, dof <= 16): one more program -- row index
                                  n_obstacles of sco_sqp_load_program -- is a NON-QUADRATIC OBJECTIVE TERM f(theta[t], p) per
                                  timestep (Prob.add_obj_expr on a plain Expr, prob.py:88-104): convexified to degree 2 on
                                  every SQP iteration exactly like SCO_FAM_FLAG_EE_COST (numeric Hessian, eigenvalue shift,
                                  numeric gradient, expr.py:102-156; model into P and q, prob.py:348-367) */

#define SCO_FAM_FLAG_ACC_COST 256     /* r04, any family: the quadratic objective carries an ACCELERATION term
                                  sum_t sum_j a_j (theta[t+2][j] - 2 theta[t+1][j] + theta[t][j])^2 next to the velocity term (a QuadExpr built
                                  from first and second difference matrices, prob.py:88-104, 348-367): P gets its second
                                  super-diagonal block; weights via sco_sqp_load_acc_weights (0 until then); horizon >= 3 */

typedef struct sco_trajopt_desc {
  int batch;
  int dof;
  int horizon;
  int n_points;      /* link points per arm configuration   */
  int n_obstacles;   /* circular obstacles per problem (rows per block for the state families) */
  int family;        /* SCO_FAM_*                           */
  int analytic_jac;  /* 0: finite differences (reference default, expr.py:86-87), 1: analytic */
  int prox_count;    /* how many Variables with a value hold each atom: the projection QP of
                        find_closest_feasible_point adds one (x_i - x0_i)^2 per Variable
                        (prob.py:381-404); 0 is read as 1.  With span > 1 this is the count of an atom that ONE
                        block Variable covers; an atom covered by k blocks counts prox_count - 1 + k */
  int span;          /* SCO_FAM_STATE_PROGRAM: timesteps per constraint block, 1 .. 4 (0 is read as 1; r04: 3 and 4) */
  int n_eq_rows;     /* SCO_FAM_STATE_PROGRAM and SCO_FAM_STATE_QUADRATIC: how many of the n_obstacles rows of a block (the last
                        ones) are equalities */
} sco_trajopt_desc;

typedef struct sco_sqp sco_sqp;

int sco_sqp_create(int device, const sco_trajopt_desc *desc, sco_sqp **out);
/* r04: the same with n_rows GENERAL affine rows over the trajectory variables, placed behind the built-in linear rows (pins,
 * velocity limits, joint limits) in every QP -- what a caller of the reference adds with
 * prob.add_cnt_expr(BoundExpr(EqExpr / LEqExpr(AffExpr(A, b), val), traj)) (prob.py:126-131, 317-346; rows of AffExpr
 * constraints go straight into the QP, they are never penalised).  The sparsity pattern is shared by the batch and given in CSR
 * form: row_ptr[n_rows + 1], col_idx[row_ptr[n_rows]] strictly increasing inside a row, column t * dof + j = coordinate j of
 * timestep t (the name-sorted atoms, osqp_utils.py:136-143); row_is_eq[r] != 0: a x = rhs (lb = ub, prob.py:339-346), else
 * a x <= rhs (lb = -inf, prob.py:329-338).  Coefficients and right-hand sides are per-problem values:
 * sco_sqp_load_linear_rows(h, vals[batch][nnz] in CSR order, rhs[batch][n_rows]) after sco_sqp_load, before sco_sqp_solve. */
int sco_sqp_create_rows(int device, const sco_trajopt_desc *desc, int n_rows, const int *row_ptr, const int *col_idx,
                        const int *row_is_eq, sco_sqp **out);
int sco_sqp_load_linear_rows(sco_sqp *h, const double *vals, const double *rhs);
/* r04: TWO kinds of non-linear rows in one problem.  SCO_FAM_STATE_PROGRAM with span 1 and dof >= 2, before sco_sqp_load_program:
 * the first n_rows rows of every block are keep-out rows  r_o - || x[0:2] - c_o || <= 0  of the point (x[0], x[1]) against
 * obstacles[problem][0 .. n_rows) of sco_sqp_load (the rows of SCO_FAM_POINT_CIRCLES, closed-form gradients with analytic_jac), the
 * program supplies the remaining n_obstacles - n_rows rows (its equality rows, if any, stay the last rows of the block).  In the
 * reference these are two BoundExprs on the same timestep Variable, the circle Expr added first (prob.py:112-144).  n_rows = 0
 * restores the plain program family; changing it invalidates a loaded program. */
int sco_sqp_set_circle_rows(sco_sqp *h, int n_rows);
int sco_sqp_destroy(sco_sqp *h);

/* Upload per-problem data (host pointers, problem-major):
 *   x0[batch][horizon*dof]   initial trajectory
 *   start[batch][dof], goal[batch][dof]
 *   link_len[batch][dof]
 *   point_link[n_points] (shared), point_frac[n_points] (shared): point k sits at
 *       fraction point_frac[k] along link point_link[k]
 *   obstacles[batch][n_obstacles][3] = (cx, cy, radius) */
int sco_sqp_load(sco_sqp *h, const double *x0, const double *start, const double *goal,
                 const double *link_len, const int *point_link, const double *point_frac,
                 const double *obstacles);
/* SCO_FAM_ARM_REACH only, after sco_sqp_load: target[batch][2] end-effector position the last
 * timestep must reach (`goal` of sco_sqp_load is then unused and may repeat `start`). */
int sco_sqp_load_target(sco_sqp *h, const double *target);
/* SCO_FAM_STATE_QUADRATIC only, after sco_sqp_load: Q[batch][n_obstacles][dof*dof] (symmetric), a[batch][n_obstacles][dof],
 * c[batch][n_obstacles].  With n_eq_rows > 0 the last n_eq_rows rows of every timestep are equalities g = 0 (r03). */
int sco_sqp_load_quadratic(sco_sqp *h, const double *Q, const double *a, const double *c);
/* SCO_FAM_STATE_PROGRAM only, after sco_sqp_load: row r's program is words[2 * row_ptr[r] .. 2 * row_ptr[r+1]) as (op, arg)
 * pairs, the last one SCO_OP_END; consts[n_consts]; params[batch][n_params] (n_params may be 0).  There are R = n_obstacles
 * programs (the rows of a block: inequalities first, then the n_eq_rows equalities; SCO_OP_X addresses the span * dof numbers
 * of the block's state), with SCO_FAM_FLAG_OBJ_PROGRAM one more -- the objective term of a timestep, SCO_OP_X < dof --
 * and row_ptr[R] (or row_ptr[R + 1]) = n_words.  row_ptr is checked as a whole (first entry 0, strictly increasing, last =
 * n_words) before any word is read through it, then every program (stack depth, operand indices, one result) before anything
 * is uploaded.  May be called again (new parameters per solve): the handle reuses its buffers. */
int sco_sqp_load_program(sco_sqp *h, int n_words, const int *words, const int *row_ptr, int n_consts, const double *consts,
                         int n_params, const double *params);
/* r04: the same with one parameter vector per problem AND timestep, params[batch][horizon][n_params]: block t (the Variable of
 * timesteps t .. t + span - 1) and the objective term of timestep t read params[problem][t] -- what a caller of the reference
 * gets by closing each timestep's Expr over its own data (moving obstacles, time-varying references: expr.py:22-41,
 * prob.py:112-144).  Rows t >= horizon - span + 1 of the parameter array are read by the objective term only. */
int sco_sqp_load_program_steps(sco_sqp *h, int n_words, const int *words, const int *row_ptr, int n_consts, const double *consts,
                               int n_params, const double *params);
/* r04: per-problem, per-joint weights of the smoothing objective, w[batch][dof] finite and >= 0:
 *     sum_t sum_j w_j (theta[t+1][j] - theta[t][j])^2
 * i.e. the QuadExpr a caller of the reference builds from a weighted difference matrix (prob.py:88-104, 348-367; the unweighted
 * form is the default).  After sco_sqp_load; NULL restores all weights 1.  Any family. */
int sco_sqp_load_obj_weights(sco_sqp *h, const double *w);
/* SCO_FAM_FLAG_ACC_COST only, after sco_sqp_load: a[batch][dof] finite and >= 0; NULL = all 0. */
int sco_sqp_load_acc_weights(sco_sqp *h, const double *a);
/* SCO_FAM_FLAG_VEL_LIMITS only, after sco_sqp_load: vmax[batch] > 0, one limit per problem. */
int sco_sqp_load_vel_limit(sco_sqp *h, const double *vmax);
/* SCO_FAM_FLAG_JOINT_LIMITS only, after sco_sqp_load: lo[batch][dof] < hi[batch][dof]. */
int sco_sqp_load_joint_limits(sco_sqp *h, const double *lo, const double *hi);
/* SCO_FAM_FLAG_EE_COST only, after sco_sqp_load: weight[batch] >= 0, target[batch][2]
 * (what a caller of the reference passes as Expr(f) to Prob.add_obj_expr, prob.py:88-104). */
int sco_sqp_load_ee_cost(sco_sqp *h, const double *weight, const double *target);

/* Constraint groups (prob.add_cnt_expr(bound_expr, group_ids), prob.py:112-142): n_groups <= 32 group ids
 * in SORTED order (the reference sorts them, prob.py:538, 559); block_mask[n_blocks], n_blocks = horizon
 * (+ 1 for the equality block of SCO_FAM_ARM_REACH, last): bit g set = that constraint block belongs to
 * group g.  Groups sharing a block overlap (prob.py:139-142).  Never called = one group "all" holding
 * every block (prob.py:135-136).  Shared by the batch; call before sco_sqp_solve. */
int sco_sqp_set_groups(sco_sqp *h, int n_groups, const unsigned int *block_mask);
/* nonconverged[batch]: bit g set = group g is in prob.nonconverged_groups after the last solve
 * (violated and no longer improving when _min_merit_fn last returned, solver.py:209-235). */
int sco_sqp_fetch_groups(sco_sqp *h, unsigned int *nonconverged);
/* stalled[batch]: bit g set = group g ENDED the last minimisation (violated, its model predicts no progress and no group
 * sharing a constraint with it progresses, solver.py:209-228).  A subset of sco_sqp_fetch_groups' mask; the reference's
 * prob.nonconverged_groups list is these groups followed by all of that mask (solver.py:232-234), which is what the
 * object-API path (sco_osqp/compile.py: write_back) rebuilds. */
int sco_sqp_fetch_stalled_groups(sco_sqp *h, unsigned int *stalled);

/* Run Solver.solve(prob, method="penalty_sqp") for every problem of the batch
 * (solver.py:30-105) starting from the loaded state; blocks until all are done. */
int sco_sqp_solve(sco_sqp *h, const sco_sqp_params *params, const sco_qp_settings *qp_settings);

/* Results (any pointer may be NULL): x[batch][n_x] final trajectories,
 * success[batch] (return value of Solver.solve), sqp_iters[batch] (passes of the
 * outer loop body solver.py:126-253), qp_solves[batch], admm_iters[batch] (sum),
 * merit[batch], max_violation[batch] (prob.py:592-603). */
int sco_sqp_fetch(sco_sqp *h, double *x, int *success, int *sqp_iters, int *qp_solves,
                  long long *admm_iters, double *merit, double *max_violation);

/* Per-problem diagnostics of the last solve (bit flags): where this implementation had to leave the
 * reference's unbounded behaviour. */
#define SCO_SQP_FLAG_MEMO_FULL 1   /* a point could not be added to the Q3 memo histories (40 evaluated / 24
                                      convexified points per constraint block); the reference caches without
                                      bound, so a LATER point rounding onto it would be recomputed here          */
#define SCO_SQP_FLAG_CAPPED 2      /* max_sqp_iters stopped the problem (the reference's loops are unbounded)   */
#define SCO_SQP_FLAG_TRACE_FULL 4  /* more decisions than sco_sqp_trace keeps (64)                              */
int sco_sqp_fetch_flags(sco_sqp *h, int *flags);
/* Number of device rounds (pre -> QP setup -> ADMM launch -> post) of the last solve, the projection round
 * included; with time slicing one QP spans several rounds.  Default schedule: ONE group with round selection (a
 * batch larger than the CU count runs whole passes of the problems with most work in front of them); SCO_SQP_GROUPS=2..4
 * in the environment cuts the batch into stream groups whose rounds run side by side (since r03 each group selects its
 * own whole passes; results do not depend on either).
 * Scheduling only, results unchanged.  `rounds` counts the group that needed most. */
int sco_sqp_last_rounds(const sco_sqp *h, int *rounds);
/* Round launches of the last solve summed over its stream groups (projection round excluded), and the number of
 * groups it used (either pointer may be null). */
int sco_sqp_last_launches(const sco_sqp *h, int *launches, int *groups);

/* The ADMM launches of the last solve by kernel tier (diagnostics; bench.py prices each kernel against its own time):
 * index 0 = the wavefront tier (csrc/sco_admm_wv.hip: one wavefront per problem, rounds with >= ~3 live problems per CU),
 * 1 = the other ADMM kernels (row-local kernel ...: the tail of a step, small batches).  ms: HIP-event time of those launches
 * (ms[0] + ms[1] = ms[2] of sco_sqp_last_timing); iters[0]: problem-iterations the wavefront kernel ran, iters[1] = -1 (the
 * rest of sco_sqp_fetch's admm_iters sum, the projection QPs included); launches: rounds on either. */
int sco_sqp_last_tiers(const sco_sqp *h, double ms[2], long long iters[2], int launches[2]);

/* Per-problem decision trace of the last solve, for stage-wise parity checks:
 * trace[batch][cap][8] = {kind, merit, model_merit, new_merit, trust, penalty,
 * qp_status, qp_iters}; n_entries[batch].  kind: 0 projection QP, 1 accepted step,
 * 2 shrink, 3 y-converged, 4 x-converged, 5 bad model, 6 group-converged. */
int sco_sqp_trace(sco_sqp *h, int cap, double *trace, int *n_entries);

/* Device milliseconds of the last sco_sqp_solve by stage (HIP events):
 * ms[0] convexify+assemble, ms[1] qp setup, ms[2] admm, ms[3] merit/decision, ms[4] total. */
int sco_sqp_last_timing(const sco_sqp *h, double ms[5]);

#ifdef __cplusplus
}
#endif
#endif /* SCO_HIP_H */
