// sco_sqp.hip -- device-resident penalty-SQP loop for a batch of trajectory
// problems (SQP layer of libsco_hip).
//
// Per problem this is Solver.solve(prob, method="penalty_sqp") of the reference
// (/root/reference/sco_py/sco_osqp/solver.py:30-253) together with the Prob methods
// it drives (prob.py:369-652), restated as a per-problem state machine so that a
// whole batch advances in lock-step "rounds", one QP solve per active problem per
// round:
//
//   round 0   projection QP           find_closest_feasible_point (prob.py:369-412)
//   round r   sqp_pre_kernel          convexify (prob.py:522-544, expr.py:130-142,
//                                     353-371) with finite-difference Jacobians
//                                     (expr.py:61-69), update_obj (prob.py:414-512,
//                                     incl. quirks Q1/Q2), get_value (prob.py:571-579),
//                                     save (prob.py:639-645), add_trust_region
//                                     (variable.py:37-45), all written straight into the
//                                     QP value arrays (no dense P/A as in
//                                     osqp_utils.py:146-193)
//             qp setup + ADMM         sco_qp.hip
//             sqp_post_kernel         get_approx_value / get_value (prob.py:605-630,
//                                     571-579), the accept / shrink / converge decisions
//                                     (solver.py:151-251), penalty escalation
//                                     (solver.py:84-105)
//
// One workgroup of 256 threads per problem; problems are independent, so there is
// no inter-workgroup communication.  The host only reads one "active problems"
// counter per round.
#include "sco_internal.h"

#include <algorithm>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define NWAVE (SCO_BLOCK / 64)
#define FD_LEVELS 4
#define FD_BASE (1.0 / 64.0)

enum { ST_PROJECT = 0, ST_CONVEXIFY = 1, ST_TRIAL = 2, ST_DONE = 3 };
enum { STEP_PROJECT = 0, STEP_ACCEPT = 1, STEP_SHRINK = 2, STEP_YCONV = 3, STEP_XCONV = 4, STEP_BAD = 5, STEP_GROUP = 6 };

#define TRACE_W 8

struct SqpScalars {
  double penalty, trust, slack_cost, merit, merit_viol;
  long long admm_iters;
  int state, k, sqp_iters, qp_solves, success, escalations, n_trace, spawned;
  int flags;      // SCO_SQP_FLAG_*
};

#define SQP_MAX_GROUPS 4
#define SQP_DEPTH 2        // rounds kept in flight per stream group

struct SqpDev {
  int b0;            // launch window of a stream group: workgroup g of the round kernels works on problem b0 + g
  int batch, d, T, K, O, R, n_x, n_slack, n, m_lin, m_nl, m, prox_count, analytic_jac, trace_cap;
  // SCO_FAM_ARM_REACH: NE = 2 equality rows (end-effector x, y) on the last timestep, block index T;
  // NB = number of constraint blocks (T or T + 1), RM = widest block (history strides)
  int NE, NB, RM;
  // constraint blocks of the state families: block t covers the S (span) timesteps t .. t + S - 1, its state is the
  // ds = S d numbers x[t d .. t d + ds) (consecutive timesteps are contiguous), there are NBt = T - S + 1 of them, and the
  // last Req of a block's R rows are equalities (r03).  Arm / point families: S = 1, ds = d, NBt = T, Req = 0.
  int S, ds, NBt, Req;
  int point;         // 1 SCO_FAM_POINT_CIRCLES: the rows are distances of the point x[0:2] itself (no arm kinematics);
                     // 2 SCO_FAM_STATE_QUADRATIC: general quadratic rows with the coefficients below
  double *qQ, *qa, *qc;   // [B][O][d*d], [B][O][d], [B][O]
                     // 3 SCO_FAM_STATE_PROGRAM: rows as postfix programs (shared) over the state and per-problem parameters
  const int *pw, *pptr; const double *pconst; const double *ppar; int n_par;
  int R1;            // r04 (SCO_FAM_STATE_PROGRAM): the first R1 rows of a block are keep-out rows of the point (x[0], x[1]) against
                     // obstacles[b][0 .. R1) as in SCO_FAM_POINT_CIRCLES -- a SECOND kind of non-linear rows on the same Variable
                     // (sco_sqp_set_circle_rows); programs supply the remaining O - R1 rows
  int par_step;      // r04: 0 = one parameter vector per problem; n_par = one per problem AND timestep (sco_sqp_load_program_steps:
                     // block t and the objective term of timestep t read params[problem][t])
  int acc;                // r04 (SCO_FAM_FLAG_ACC_COST): P holds the second super-diagonal block too
  const double *accw;     // r04: [B][d] weights of the acceleration term sum_t sum_j a_j (x[t+2][j] - 2 x[t+1][j] + x[t][j])^2, nullptr = all 0
  const double *objw;     // r04: [B][d] weights of the smoothing objective sum_t sum_j w_j (x[t+1][j] - x[t][j])^2, nullptr = all 1
  // linear rows: m_pin pins (start, and goal unless reach), then m_vel velocity-limit rows, then m_jl joint-limit
  // rows (theta <= hi for every trajectory variable, then -theta <= -lo)
  int m_pin, m_vel, m_jl;
  // r04: m_gen GENERAL affine rows behind them (a shared CSR pattern over the trajectory variables, values and right-hand
  // sides per problem: sco_sqp_create_rows / sco_sqp_load_linear_rows): row r is an equality (gen_eq[r]) a x = rhs or an
  // inequality a x <= rhs; gpos0 / gpos1: where entry k of the pattern sits in the A values of the projection / penalty QP
  int m_gen, nnz_gen;
  const int *gen_eq, *gpos0, *gpos1;
  const double *gen_rhs, *gen_val;      // [B][m_gen], [B][nnz_gen]
  // SCO_FAM_FLAG_EE_COST: non-quadratic objective term weight * ||ee(theta_t) - target||^2 per timestep, convexified to
  // degree 2 every SQP iteration (expr.py:143-153): oH = Hessian after the eigenvalue shift, oA, ob = the model's
  // linear and constant part; ppos = position in qp1's P values of entry (row (t, 0), column (t, j))
  int cost;          // 1 = the arm's end-effector term, 2 = an objective PROGRAM (SCO_FAM_FLAG_OBJ_PROGRAM: program index R)
  double *cw, *ctgt, *oH, *oA, *ob;   // [B], [B][2], [B][T][d*d], [B][T][d], [B][T]
  const int *ppos;                    // [n_x]
  const double *a0c, *a1c;      // constant parts of the A values of the projection / penalty QP (shared)
  double *vmax;                 // [B]
  double *jlo, *jhi;            // [B][d]
  // problem data
  double *x0, *start, *goal, *link_len, *obstacles, *target;   // [B][...]
  const int *point_link; const double *point_frac;    // [K]
  // SQP state
  double *x, *x_saved, *gsave, *J, *bmod, *trace;
  unsigned char *mask;
  SqpScalars *sc;
  int *active, *n_active;
  int *newqp;        // [B] 1 = this round's pre kernel prepared a new QP for the problem (setup mask)
  const int *list;   // [B] or null: workgroup g of a round kernel works on problem list[g] (< 0: none) -- the problems
                     // sqp_select_kernel lets take part in this round; the others are not visited at all
  int *list_buf;     // [B] storage of `list`
  const int *jpos;   // [T*d] CSC position of J[t][0][j] in qp1's A values
  const int *epos;   // [d]   CSC position of the first equality-row entry of column (T-1, j)
  // constraint groups (prob.py:81-86, 135-142): G = 0 means the default single group "all"
  int G;
  const unsigned int *gmask;      // [NB] groups of every constraint block
  const unsigned int *goverlap;   // [32] groups sharing a block with group g
  double *mvec;                   // [B][32] per-group violation at the convexification point
  unsigned int *nonconv;          // [B] prob.nonconverged_groups: the violated groups under the y threshold (solver.py:232-234)
  unsigned int *stalled;          // [B] the groups that ended the minimisation (solver.py:209-228); the reference lists them first
  const int *bpos;   // CSC positions are not needed for bounds: rows are contiguous
  // Q3 emulation: per timestep block, the points already seen (keys = rint(x * 1e6), the
  // reference's tuple(x.round(6))) with their f values, and the points already
  // convexified with their affine model
  int H, HC;
  double *hkey, *hval, *ckey, *cJ, *cb;   // [B][NB][H][d], [B][NB][H][RM], [B][NB][HC][d], [B][NB][HC][RM*d], [B][NB][HC][RM]
  int *hn, *cn;                            // [B][NB]
};

struct sco_sqp {
  int device = 0;
  sco_trajopt_desc desc{};
  sco_qp *qp0 = nullptr, *qp1 = nullptr;
  SqpDev d{};
  hipStream_t stream = nullptr;
  // stream groups of the round loop (sco_sqp_solve): group 0 runs on `stream`, group g > 0 on gstream[g - 1]
  hipStream_t gstream[SQP_MAX_GROUPS - 1] = {};
  std::vector<hipEvent_t> gevents[SQP_MAX_GROUPS], done;
  double *fetch_buf = nullptr;       // [2][B] merit and max violation of sco_sqp_fetch
  int *host_active = nullptr;        // pinned: [SQP_MAX_GROUPS][SQP_DEPTH] active counts read back per round
  int groups_used = 1;
  std::vector<void *> allocs;
  std::vector<hipEvent_t> events;
  void *prog_buf[4] = {nullptr, nullptr, nullptr, nullptr};   // sco_sqp_load_program: words, row starts, constants, parameters
  void *objw_buf = nullptr;                                    // sco_sqp_load_obj_weights
  void *accw_buf = nullptr;                                    // sco_sqp_load_acc_weights
  std::vector<int> gen_ptr, gen_col, gen_iseq;                 // sco_sqp_create_rows: the general affine rows' pattern
  bool gen_loaded = false;
  size_t prog_bytes[4] = {0, 0, 0, 0};
  bool loaded = false, solved = false, target_loaded = false, vel_loaded = false, jl_loaded = false, cost_loaded = false, quad_loaded = false, prog_loaded = false;
  double last_ms[5] = {0, 0, 0, 0, 0};
  int rounds = 0;          // 1 (projection) + the rounds of the group that needed most
  int launches = 0;        // round launches over all stream groups
  int wv_rounds = 0;       // of them on the wavefront tier
  double wv_ms = 0.0;      // ADMM time of those rounds (part of last_ms[2])
};

// --------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------
__device__ __forceinline__ double wsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ double wmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
// sums v[0..NS) and maxes v[NS..NS+NM) over the workgroup; result in every thread
template <int NS, int NM>
__device__ __forceinline__ void block_reduce_sm(double (&v)[NS + NM], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NS; k++) v[k] = wsum(v[k]);
#pragma unroll
  for (int k = NS; k < NS + NM; k++) v[k] = wmax(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NS + NM; k++) red[wv * (NS + NM) + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NS + NM; k++) {
    double r = red[k];
#pragma unroll
    for (int w = 1; w < NWAVE; w++) {
      const double o = red[w * (NS + NM) + k];
      r = (k < NS) ? r + o : fmax(r, o);
    }
    v[k] = r;
  }
}

// SCO_FAM_ARM_CIRCLES: row (k, o) of one timestep block,
//   g = r_o - || p_k(theta) - c_o ||,  p_k = planar forward kinematics.
// `pert` >= 0 adds `h` to joint `pert` (finite differences).
__device__ __forceinline__ double arm_row(const double *th, const double *len, int lk, double frac,
                                          double cx, double cy, double rad, int pert, double h) {
  double phi = 0.0, px = 0.0, py = 0.0;
  for (int i = 0; i <= lk; i++) {
    double a = th[i];
    if (i == pert) a += h;
    phi += a;
    double s, c;
    sincos(phi, &s, &c);
    const double L = (i == lk) ? frac * len[i] : len[i];
    px += L * c; py += L * s;
  }
  const double dx = px - cx, dy = py - cy;
  return rad - sqrt(dx * dx + dy * dy);
}

// analytic d g / d theta_j of the same row
__device__ __forceinline__ double arm_row_grad(const double *th, const double *len, int lk, double frac,
                                               double cx, double cy, int j) {
  if (j > lk) return 0.0;
  double phi = 0.0, px = 0.0, py = 0.0, sx = 0.0, sy = 0.0;
  for (int i = 0; i <= lk; i++) {
    phi += th[i];
    double s, c;
    sincos(phi, &s, &c);
    const double L = (i == lk) ? frac * len[i] : len[i];
    px += L * c; py += L * s;
    if (i >= j) { sx += -L * s; sy += L * c; }
  }
  const double dx = px - cx, dy = py - cy;
  return -(dx * sx + dy * sy) / sqrt(dx * dx + dy * dy);
}

// SCO_FAM_ARM_REACH: component `comp` (0 = x, 1 = y) of the end-effector position
__device__ __forceinline__ double arm_ee(const double *th, const double *len, int d, int comp, int pert, double h) {
  double phi = 0.0, p = 0.0;
  for (int i = 0; i < d; i++) {
    double a = th[i];
    if (i == pert) a += h;
    phi += a;
    double sn, cs;
    sincos(phi, &sn, &cs);
    p += len[i] * (comp == 0 ? cs : sn);
  }
  return p;
}
__device__ __forceinline__ double arm_ee_grad(const double *th, const double *len, int d, int comp, int j) {
  double phi = 0.0, g = 0.0;
  for (int i = 0; i < d; i++) {
    phi += th[i];
    double sn, cs;
    sincos(phi, &sn, &cs);
    if (i >= j) g += comp == 0 ? -(len[i] * sn) : len[i] * cs;
  }
  return g;
}

// SCO_FAM_FLAG_EE_COST: weight * ||ee(theta) - target||^2 with up to two perturbed coordinates (finite differences)
__device__ __forceinline__ double arm_ee_cost(const double *th, const double *len, int d, double tx, double ty, double w,
                                              int pi, double hi, int pj, double hj) {
  double phi = 0.0, ex = 0.0, ey = 0.0;
  for (int i = 0; i < d; i++) {
    double a = th[i];
    if (i == pi) a += hi;
    if (i == pj) a += hj;
    phi += a;
    double sn, cs;
    sincos(phi, &sn, &cs);
    ex += len[i] * cs; ey += len[i] * sn;
  }
  ex -= tx; ey -= ty;
  return w * (ex * ex + ey * ey);
}

// Richardson extrapolation of a central-difference ladder (halving steps): sco_py_amd/numdiff.py:_richardson
__device__ __forceinline__ double richardson(double (&tab)[FD_LEVELS]) {
  double p4 = 4.0;
#pragma unroll
  for (int i = 1; i < FD_LEVELS; i++) {
    const double fac = 1.0 / (p4 - 1.0);
#pragma unroll
    for (int lv = FD_LEVELS - 1; lv >= i; lv--) tab[lv] = tab[lv] + (tab[lv] - tab[lv - 1]) * fac;
    p4 *= 4.0;
  }
  return tab[FD_LEVELS - 1];
}

// smallest eigenvalue of the symmetric d x d matrix H (d <= 16) by cyclic Jacobi rotations on a private copy
// (the reference calls scipy.linalg.eigvalsh, expr.py:145; oracle/sco_ref.py:min_eig_jacobi is this sweep)
#define OBJ_DMAX 16
#define SCO_STATE_MAX 32      // widest state of a constraint block of the program family (span x dof)
__device__ double min_eig_jacobi(const double *H, int d) {
  double A[OBJ_DMAX * OBJ_DMAX];
  for (int i = 0; i < d * d; i++) A[i] = H[i];
  for (int sweep = 0; sweep < 12; sweep++)
    for (int p = 0; p < d - 1; p++)
      for (int q = p + 1; q < d; q++) {
        const double apq = A[p * d + q];
        if (fabs(apq) < 1e-300) continue;
        const double theta = (A[q * d + q] - A[p * d + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        for (int k = 0; k < d; k++) {            // columns p, q
          const double rp = A[k * d + p], rq = A[k * d + q];
          A[k * d + p] = c * rp - sn * rq; A[k * d + q] = sn * rp + c * rq;
        }
        for (int k = 0; k < d; k++) {            // rows p, q
          const double rp = A[p * d + k], rq = A[q * d + k];
          A[p * d + k] = c * rp - sn * rq; A[q * d + k] = sn * rp + c * rq;
        }
      }
  double lam = A[0];
  for (int i = 1; i < d; i++) lam = fmin(lam, A[i * d + i]);
  return lam;
}

// Non-linear row e of a problem: hinge rows (timestep-major, R per timestep) first, then the NE
// equality rows of the reach variant on the last timestep (constraint block index T).
// `eq`: 0 hinge row, 1 equality row of the reach variant (block NBt = T on the last timestep, arm kinematics),
// 2 equality row inside a block of the state families (the last Req rows of the block, r03)
struct RowRef { int blk, t, r, eq; };
struct RowLay { int T, NBt, R, Req; };
__device__ __forceinline__ RowRef row_ref(int e, const RowLay &L) {
  RowRef q;
  if (e < L.NBt * L.R) { q.blk = e / L.R; q.t = q.blk; q.r = e % L.R; q.eq = (q.r >= L.R - L.Req) ? 2 : 0; }
  else { q.blk = L.NBt; q.t = L.T - 1; q.r = e - L.NBt * L.R; q.eq = 1; }
  return q;
}

struct RowCtx { const double *len, *obs, *target; const int *point_link; const double *point_frac; int d, O, point;
                const double *qQ, *qa, *qc;         // SCO_FAM_STATE_QUADRATIC: this problem's row coefficients (point == 2)
                const int *pw, *pptr; const double *pconst, *ppar; int par_step, R1; };   // SCO_FAM_STATE_PROGRAM (point == 3): words, row starts, constants, this problem's parameters (block t: ppar + t * par_step)
// (for the state families `d` is the dimension of a BLOCK's state, span x dof)

// SCO_FAM_STATE_PROGRAM: value of program `o` at th, up to two coordinates perturbed (finite differences)
__device__ __forceinline__ double prog_eval(const RowCtx &c, const double *par, int o, const double *th, int pi, double hi, int pj, double hj) {
  double st[SCO_PROGRAM_STACK];
  int sp = 0;
  for (int w = c.pptr[o];; w++) {
    const int op = c.pw[2 * w], arg = c.pw[2 * w + 1];
    if (op == SCO_OP_END) break;
    switch (op) {
      case SCO_OP_X: st[sp++] = th[arg] + (arg == pi ? hi : 0.0) + (arg == pj ? hj : 0.0); break;
      case SCO_OP_P: st[sp++] = par[arg]; break;
      case SCO_OP_C: st[sp++] = c.pconst[arg]; break;
      case SCO_OP_ADD: sp--; st[sp - 1] = st[sp - 1] + st[sp]; break;
      case SCO_OP_SUB: sp--; st[sp - 1] = st[sp - 1] - st[sp]; break;
      case SCO_OP_MUL: sp--; st[sp - 1] = st[sp - 1] * st[sp]; break;
      case SCO_OP_DIV: sp--; st[sp - 1] = st[sp - 1] / st[sp]; break;
      case SCO_OP_NEG: st[sp - 1] = -st[sp - 1]; break;
      case SCO_OP_SIN: st[sp - 1] = sin(st[sp - 1]); break;
      case SCO_OP_COS: st[sp - 1] = cos(st[sp - 1]); break;
      case SCO_OP_SQRT: st[sp - 1] = sqrt(st[sp - 1]); break;
      case SCO_OP_EXP: st[sp - 1] = exp(st[sp - 1]); break;
      default: st[sp - 1] = st[sp - 1] * st[sp - 1]; break;       // SCO_OP_SQUARE
    }
  }
  return st[0];
}
// d(program o) / d x_j by forward-mode differentiation: every stack slot carries (value, derivative along e_j); the rules,
// in this order of operations, are the ones sco_py_amd/rowexpr.py:Program.jacobian applies on the host (the `grad` a caller
// hands to the reference's Expr, expr.py:86-100), so host and device agree to rounding
__device__ __forceinline__ double prog_dual(const RowCtx &c, const double *par, int o, const double *th, int j) {
  // no fused multiply-adds here: the host applies the rules operation by operation (NumPy), and a derivative that differs in
  // its last bit can move the iteration count of a slowly converging QP by many termination checks
#pragma clang fp contract(off)
  double sv[SCO_PROGRAM_STACK], sd[SCO_PROGRAM_STACK];
  int sp = 0;
  for (int w = c.pptr[o];; w++) {
    const int op = c.pw[2 * w], arg = c.pw[2 * w + 1];
    if (op == SCO_OP_END) break;
    switch (op) {
      case SCO_OP_X: sv[sp] = th[arg]; sd[sp++] = (arg == j) ? 1.0 : 0.0; break;
      case SCO_OP_P: sv[sp] = par[arg]; sd[sp++] = 0.0; break;
      case SCO_OP_C: sv[sp] = c.pconst[arg]; sd[sp++] = 0.0; break;
      case SCO_OP_ADD: sp--; sv[sp - 1] = sv[sp - 1] + sv[sp]; sd[sp - 1] = sd[sp - 1] + sd[sp]; break;
      case SCO_OP_SUB: sp--; sv[sp - 1] = sv[sp - 1] - sv[sp]; sd[sp - 1] = sd[sp - 1] - sd[sp]; break;
      case SCO_OP_MUL: sp--; sd[sp - 1] = sd[sp - 1] * sv[sp] + sv[sp - 1] * sd[sp]; sv[sp - 1] = sv[sp - 1] * sv[sp]; break;
      case SCO_OP_DIV: { sp--; const double qv = sv[sp - 1] / sv[sp]; sd[sp - 1] = (sd[sp - 1] - qv * sd[sp]) / sv[sp]; sv[sp - 1] = qv; break; }
      case SCO_OP_NEG: sv[sp - 1] = -sv[sp - 1]; sd[sp - 1] = -sd[sp - 1]; break;
      case SCO_OP_SIN: sd[sp - 1] = cos(sv[sp - 1]) * sd[sp - 1]; sv[sp - 1] = sin(sv[sp - 1]); break;
      case SCO_OP_COS: sd[sp - 1] = -(sin(sv[sp - 1]) * sd[sp - 1]); sv[sp - 1] = cos(sv[sp - 1]); break;
      case SCO_OP_SQRT: { const double r = sqrt(sv[sp - 1]); sd[sp - 1] = sd[sp - 1] / (2.0 * r); sv[sp - 1] = r; break; }
      case SCO_OP_EXP: { const double ev = exp(sv[sp - 1]); sd[sp - 1] = ev * sd[sp - 1]; sv[sp - 1] = ev; break; }
      default: sd[sp - 1] = (2.0 * sv[sp - 1]) * sd[sp - 1]; sv[sp - 1] = sv[sp - 1] * sv[sp - 1]; break;       // SCO_OP_SQUARE
    }
  }
  return sd[0];
}
// f of row q at th (the raw function value: the right-hand side val is 0 for hinge rows and the equality rows of the state
// families and target[r] for the reach rows and is applied by the callers, in the reference's order)
__device__ __forceinline__ double row_value(const RowCtx &c, const RowRef &q, const double *th, int pert, double h) {
  if (q.eq == 1) return arm_ee(th, c.len, c.d, q.r, pert, h);
  const int kp = q.r / c.O, o = q.r % c.O;
  if (c.point == 3) {                          // SCO_FAM_STATE_PROGRAM: the row's postfix program; r04: behind R1 circle rows of the point x[0:2]
    if (q.r < c.R1) {
      const double dx = th[0] + (pert == 0 ? h : 0.0) - c.obs[3 * q.r], dy = th[1] + (pert == 1 ? h : 0.0) - c.obs[3 * q.r + 1];
      return c.obs[3 * q.r + 2] - sqrt(dx * dx + dy * dy);
    }
    return prog_eval(c, c.ppar + q.t * c.par_step, q.r - c.R1, th, pert, h, -1, 0.0);
  }
  if (c.point == 2) {                          // SCO_FAM_STATE_QUADRATIC: 1/2 x' Q x + a' x + c of row o
    const double *Q = c.qQ + (size_t)o * c.d * c.d, *av = c.qa + (size_t)o * c.d;
    double val = c.qc[o];
    for (int i = 0; i < c.d; i++) {
      const double xi = th[i] + (i == pert ? h : 0.0);
      double qx = 0.0;
      for (int j2 = 0; j2 < c.d; j2++) qx += Q[i * c.d + j2] * (th[j2] + (j2 == pert ? h : 0.0));
      val += xi * (av[i] + 0.5 * qx);
    }
    return val;
  }
  if (c.point) {                               // SCO_FAM_POINT_CIRCLES: r_o - || x[0:2] - c_o ||
    const double dx = th[0] + (pert == 0 ? h : 0.0) - c.obs[3 * o], dy = th[1] + (pert == 1 ? h : 0.0) - c.obs[3 * o + 1];
    return c.obs[3 * o + 2] - sqrt(dx * dx + dy * dy);
  }
  return arm_row(th, c.len, c.point_link[kp], c.point_frac[kp], c.obs[3 * o], c.obs[3 * o + 1], c.obs[3 * o + 2], pert, h);
}
__device__ __forceinline__ double row_grad(const RowCtx &c, const RowRef &q, const double *th, int j) {
  if (q.eq == 1) return arm_ee_grad(th, c.len, c.d, q.r, j);
  const int kp = q.r / c.O, o = q.r % c.O;
  if (c.point == 3) {                          // forward-mode differentiation of the row's program (r03); circle rows in closed form
    if (q.r < c.R1) {
      if (j > 1) return 0.0;
      const double dx = th[0] - c.obs[3 * q.r], dy = th[1] - c.obs[3 * q.r + 1];
      return -(j == 0 ? dx : dy) / sqrt(dx * dx + dy * dy);
    }
    return prog_dual(c, c.ppar + q.t * c.par_step, q.r - c.R1, th, j);
  }
  if (c.point == 2) {                          // a_j + sum_i Q_ji x_i  (Q symmetric)
    const double *Q = c.qQ + (size_t)o * c.d * c.d;
    double g = c.qa[(size_t)o * c.d + j];
    for (int i = 0; i < c.d; i++) g += Q[j * c.d + i] * th[i];
    return g;
  }
  if (c.point) {
    if (j > 1) return 0.0;
    const double dx = th[0] - c.obs[3 * o], dy = th[1] - c.obs[3 * o + 1];
    return -(j == 0 ? dx : dy) / sqrt(dx * dx + dy * dy);
  }
  return arm_row_grad(th, c.len, c.point_link[kp], c.point_frac[kp], c.obs[3 * o], c.obs[3 * o + 1], j);
}
// non-quadratic objective term of a timestep with up to two perturbed coordinates: the arm's end-effector term
// (SCO_FAM_FLAG_EE_COST) or the objective program (SCO_FAM_FLAG_OBJ_PROGRAM: program index O)
struct ObjCtx { int kind; const double *len; int d; double tx, ty, w; };
__device__ __forceinline__ double obj_value(const ObjCtx &oc, const RowCtx &c, int t, const double *th, int pi, double hi, int pj, double hj) {
  if (oc.kind == 2) return prog_eval(c, c.ppar + t * c.par_step, c.O - c.R1, th, pi, hi, pj, hj);
  return arm_ee_cost(th, oc.len, oc.d, oc.tx, oc.ty, oc.w, pi, hi, pj, hj);
}
__device__ __forceinline__ double row_rhs(const RowCtx &c, const RowRef &q) { return q.eq == 1 ? c.target[q.r] : 0.0; }
// violation of a row with value g = f - val: |g| for equality rows, max(g, 0) for hinge rows (prob.py:582-590)
__device__ __forceinline__ double row_viol(const RowRef &q, double g) { return q.eq ? fabs(g) : fmax(g, 0.0); }

// index of the stored point whose rounded key equals that of x (d coordinates), or -1
__device__ __forceinline__ int memo_find(const double *keys, int count, int d, const double *x) {
  for (int h = 0; h < count; h++) {
    const double *k = keys + (size_t)h * d;
    bool same = true;
    for (int j = 0; j < d; j++) same = same && (k[j] == rint(x[j] * 1e6));
    if (same) return h;
  }
  return -1;
}

// w: the problem's objective weights (r04, QuadExpr with per-joint weights, prob.py:348-367) or nullptr = all 1
__device__ __forceinline__ double traj_obj_partial(const double *x, int d, int T, int tid, const double *w, const double *aw) {
  double s = 0.0;
  for (int e = tid; e < (T - 1) * d; e += SCO_BLOCK) {
    const double df = x[e + d] - x[e];
    s += w ? w[e % d] * (df * df) : df * df;
  }
  if (aw)                                  // acceleration term (r04): second differences
    for (int e = tid; e < (T - 2) * d; e += SCO_BLOCK) {
      const double dd = (x[e + 2 * d] - 2.0 * x[e + d]) + x[e];
      s += aw[e % d] * (dd * dd);
    }
  return s;
}
// entries of the objective's P = Q (0.5 x'Qx form) in column (t, j): (t - 2, j), (t - 1, j), (t, j); wv / wa the joint's weights
__device__ __forceinline__ double obj_p_diag(int t, int T, double wv, double wa) {
  double v = ((t == 0 || t == T - 1) ? 2.0 : 4.0) * wv;
  if (wa != 0.0) {
    if (t <= T - 3) v += 2.0 * wa;                    // first point of window t
    if (t >= 1 && t <= T - 2) v += 8.0 * wa;          // middle point of window t - 1
    if (t >= 2) v += 2.0 * wa;                        // last point of window t - 2
  }
  return v;
}
__device__ __forceinline__ double obj_p_off1(int t, int T, double wv, double wa) {      // entry ((t - 1, j), (t, j)), t >= 1
  double v = -2.0 * wv;
  if (wa != 0.0) {
    if (t <= T - 2) v -= 4.0 * wa;                    // (first, middle) of window t - 1
    if (t >= 2) v -= 4.0 * wa;                        // (middle, last) of window t - 2
  }
  return v;
}

// upper bound of linear inequality row i (m_pin <= i < m_lin): velocity rows, then theta <= hi, then -theta <= -lo, then the
// general affine rows (prob.py:329-338: lb = -inf, ub = val - b; an equality among the general rows: lb = ub, prob.py:339-346)
__device__ __forceinline__ bool lin_row_is_eq(const SqpDev &s, int i) {
  return i >= s.m_lin - s.m_gen && s.gen_eq[i - (s.m_lin - s.m_gen)] != 0;
}
__device__ __forceinline__ double lin_ineq_hi(const SqpDev &s, int b, int i) {
  if (i >= s.m_lin - s.m_gen) return s.gen_rhs[(size_t)b * s.m_gen + (i - (s.m_lin - s.m_gen))];      // general rows (r04)
  const int v = i - s.m_pin;
  if (v < s.m_vel) return s.vmax[b];
  const int k = v - s.m_vel;
  return k < s.n_x ? s.jhi[(size_t)b * s.d + k % s.d] : -s.jlo[(size_t)b * s.d + (k - s.n_x) % s.d];
}

// --------------------------------------------------------------------------
// round 0: projection QP values  (prob.py:369-412)
//   P_ii = 2 c, q_i = -2 c x0_i  (c = number of Variables holding atom i),
//   rows = [pins ; bound rows (-inf, inf)]
// --------------------------------------------------------------------------
__global__ __launch_bounds__(SCO_BLOCK) void sqp_proj_assemble_kernel(SqpDev s, QpDev q0) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n_x = s.n_x, d = s.d, m0 = q0.m;
  const double *x0 = s.x0 + (size_t)b * n_x;
  for (int i = tid; i < n_x; i += SCO_BLOCK) {
    // one (x_i - x0_i)^2 per Variable that holds the atom (prob.py:381-404): the descriptor's count is that of an atom
    // covered by one block Variable; with blocks of S > 1 timesteps timestep t sits in `nblk` of them
    const int ts = i / d, nblk = (ts < s.NBt - 1 ? ts : s.NBt - 1) - (ts - s.S + 1 > 0 ? ts - s.S + 1 : 0) + 1;
    const double c = (double)(s.prox_count - 1 + nblk);
    // entries whose initial value is unknown (NaN) take no part in the distance (prob.py:394-404)
    const bool known = !isnan(x0[i]);
    q0.Pval[(size_t)b * q0.nnzP + i] = known ? 2.0 * c : 0.0;
    q0.q[(size_t)b * n_x + i] = known ? (-2.0 * x0[i]) * c : 0.0;
    s.x[(size_t)b * n_x + i] = x0[i];
  }
  for (int t = tid; t < q0.nnzA; t += SCO_BLOCK) q0.Aval[(size_t)b * q0.nnzA + t] = s.a0c[t];
  if (s.nnz_gen) {                       // per-problem coefficients of the general affine rows over the shared constants
    __syncthreads();
    for (int k = tid; k < s.nnz_gen; k += SCO_BLOCK) q0.Aval[(size_t)b * q0.nnzA + s.gpos0[k]] = s.gen_val[(size_t)b * s.nnz_gen + k];
  }
  for (int i = tid; i < m0; i += SCO_BLOCK) {
    double lo, hi;
    if (i < d) lo = hi = s.start[(size_t)b * d + i];
    else if (i < s.m_pin) lo = hi = s.goal[(size_t)b * d + (i - d)];
    else if (i < s.m_lin) { hi = lin_ineq_hi(s, b, i); lo = lin_row_is_eq(s, i) ? hi : -INFINITY; }
    else { lo = -INFINITY; hi = INFINITY; }
    q0.l[(size_t)b * m0 + i] = lo; q0.u[(size_t)b * m0 + i] = hi;
    q0.w[(size_t)b * m0 + i] = 1;
  }
  if (tid == 0) {
    SqpScalars &sc = s.sc[b];
    sc.state = ST_PROJECT; sc.k = 0; sc.sqp_iters = 0; sc.qp_solves = 0; sc.success = 0;
    sc.escalations = 0; sc.n_trace = 0; sc.spawned = 0; sc.admm_iters = 0; sc.flags = 0;
    sc.slack_cost = 1.0; sc.merit = 0.0; sc.merit_viol = 0.0;
    s.active[b] = 1; s.nonconv[b] = 0u; s.stalled[b] = 0u;
  }
  for (int t = tid; t < s.NB; t += SCO_BLOCK) { s.hn[(size_t)b * s.NB + t] = 0; s.cn[(size_t)b * s.NB + t] = 0; }
}

struct SqpParamsDev {
  double improve_ratio_threshold, min_trust_region_size, min_approx_improve, trust_shrink_ratio,
      trust_expand_ratio, cnt_tolerance, merit_coeff_increase_ratio, initial_trust_region_size,
      initial_penalty_coeff;
  int max_merit_coeff_increases, compound_penalty, duplicate_rows, max_qp_solves, memo;
};

__device__ __forceinline__ void trace_row(const SqpDev &s, int b, SqpScalars &sc, int kind, double merit,
                                          double model, double newm, double trust, double pen, int status,
                                          int iters) {
  if (sc.n_trace < s.trace_cap) {
    double *r = s.trace + ((size_t)b * s.trace_cap + sc.n_trace) * TRACE_W;
    r[0] = kind; r[1] = merit; r[2] = model; r[3] = newm; r[4] = trust; r[5] = pen; r[6] = status; r[7] = iters;
  }
  sc.n_trace++;
}

__global__ __launch_bounds__(SCO_BLOCK) void sqp_proj_post_kernel(SqpDev s, QpDev q0, QpDev q1, SqpParamsDev p) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const int n_x = s.n_x, n = s.n, m = s.m, d = s.d;
  const int status = q0.status[b];
  const bool ok = (status == 1 || status == 2);
  if (ok) for (int i = tid; i < n_x; i += SCO_BLOCK) s.x[(size_t)b * n_x + i] = q0.x[(size_t)b * n_x + i];
  // constant parts of the penalty QP (prob.py:251-278, 348-367; osqp_utils.py:185-189)
  {
    double *Pv = q1.Pval + (size_t)b * q1.nnzP;
    // pattern built on the host: column (t, j) holds [super-diagonal (-2) if t > 0, diagonal]
    // (with SCO_FAM_FLAG_EE_COST the column also holds rows (t, 0 .. j - 1) of its diagonal block: the Hessian of the
    // objective term, written by every convexify)
    int pos = 0;
    for (int col = 0; col < n_x; col++) {   // cheap: n_x entries, one thread
      if (tid == 0) {
        const int t = col / d;
        const double wj = s.objw ? s.objw[(size_t)b * d + col % d] : 1.0;
        const double aj = s.accw ? s.accw[(size_t)b * d + col % d] : 0.0;
        if (s.acc && t > 1) Pv[pos++] = 2.0 * aj;                            // ((t - 2, j), (t, j)): first x last of window t - 2
        if (t > 0) Pv[pos++] = obj_p_off1(t, s.T, wj, aj);
        if (s.cost) for (int i = 0; i < col % d; i++) Pv[pos++] = 0.0;
        Pv[pos++] = obj_p_diag(t, s.T, wj, aj);
      }
    }
    double *qv = q1.q + (size_t)b * n;
    for (int i = tid; i < n; i += SCO_BLOCK) qv[i] = 0.0;
    double *Av = q1.Aval + (size_t)b * q1.nnzA;
    // constant entries (pins, velocity rows, slack and bound entries; zeros where convexify writes the
    // Jacobian), built once on the host (sco_sqp_create)
    for (int t = tid; t < q1.nnzA; t += SCO_BLOCK) Av[t] = s.a1c[t];
    if (s.nnz_gen) {
      __syncthreads();
      for (int k = tid; k < s.nnz_gen; k += SCO_BLOCK) Av[s.gpos1[k]] = s.gen_val[(size_t)b * s.nnz_gen + k];
    }
    double *l = q1.l + (size_t)b * m, *u = q1.u + (size_t)b * m;
    int *w = q1.w + (size_t)b * m;
    for (int i = tid; i < m; i += SCO_BLOCK) {
      double lo, hi;
      if (i < d) lo = hi = s.start[(size_t)b * d + i];
      else if (i < s.m_pin) lo = hi = s.goal[(size_t)b * d + (i - d)];
      else if (i < s.m_lin) { hi = lin_ineq_hi(s, b, i); lo = lin_row_is_eq(s, i) ? hi : -INFINITY; }
      else if (i < s.m_lin + s.m_nl) {                                  // hinge rows; equality rows are set by convexify
        const RowRef qr = row_ref(i - s.m_lin, RowLay{s.T, s.NBt, s.R, s.Req});
        lo = qr.eq ? 0.0 : -INFINITY; hi = 0.0;
      }
      else if (i < s.m_lin + s.m_nl + n_x) { lo = -INFINITY; hi = INFINITY; }
      else { lo = 0.0; hi = INFINITY; }
      l[i] = lo; u[i] = hi; w[i] = 1;
    }
  }
  if (tid == 0) {
    SqpScalars &sc = s.sc[b];
    sc.qp_solves = 1; sc.admm_iters = q0.iters[b];
    sc.penalty = p.initial_penalty_coeff; sc.trust = p.initial_trust_region_size;
    trace_row(s, b, sc, STEP_PROJECT, 0.0, 0.0, 0.0, sc.trust, sc.penalty, status, q0.iters[b]);
    if (ok) { sc.state = ST_CONVEXIFY; atomicAdd(s.n_active, 1); }
    else { sc.state = ST_DONE; sc.success = 0; s.active[b] = 0; }   // solver.py:81-82
  }
}

// --------------------------------------------------------------------------
// sqp_pre: convexify + update_obj + merit + save (for problems that start a new
// SQP iteration), then the trust-region bounds (for every active problem)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(SCO_BLOCK) void sqp_pre_kernel(SqpDev s, QpDev q1, SqpParamsDev p) {
  const int g = blockIdx.x + s.b0, tid = threadIdx.x;
  const int b = s.list ? s.list[g] : g;        // round selection: only the listed problems are visited
  if (b < 0) return;
  SqpScalars &sc = s.sc[b];
  const int state = sc.state;
  // a problem whose QP is parked between two ADMM slices keeps its QP untouched
  const bool parked = q1.prog && q1.prog[b] > 0;
  if (tid == 0) s.newqp[b] = (state != ST_DONE && !parked) ? 1 : 0;
  if (state == ST_DONE || parked) return;
  __shared__ double red[NWAVE * 4];
  __shared__ double of0[260];           // f of the non-quadratic objective terms at the convexification point
  const int n_x = s.n_x, d = s.d, T = s.T, R = s.R, O = s.O, n = s.n, m = s.m;
  const int ds = s.ds, NBt = s.NBt;             // state dimension of a block, number of timestep blocks
  const RowLay L{T, NBt, R, s.Req};
  double *x = s.x + (size_t)b * n_x, *xs = s.x_saved + (size_t)b * n_x;
  const double *len = s.link_len + (size_t)b * d;
  const double *obs = s.obstacles + (size_t)b * O * 3;
  double *gs = s.gsave + (size_t)b * s.m_nl, *J = s.J + (size_t)b * s.m_nl * ds, *bm = s.bmod + (size_t)b * s.m_nl;
  unsigned char *mask = s.mask + (size_t)b * s.m_nl * ds;
  const double penalty = sc.penalty, trust = sc.trust;
  const int spawned = sc.spawned;
  int k_rows = sc.k;
  double slack_cost = sc.slack_cost;

  if (state == ST_CONVEXIFY) {
    // Q3: which blocks sit on an already-seen / already-convexified rounded point
    __shared__ int ev_hit[260], cv_hit[260];
    const int H = s.H, HC = s.HC, NB = s.NB, RM = s.RM, m_nl = s.m_nl;
    const RowCtx rc{len, obs, s.target + (size_t)b * 2, s.point_link, s.point_frac, ds, O, s.point,
                  s.qQ + (size_t)b * O * ds * ds, s.qa + (size_t)b * O * ds, s.qc + (size_t)b * O,
                  s.pw, s.pptr, s.pconst, s.ppar + (size_t)b * s.n_par * (s.par_step ? s.T : 1), s.par_step, s.R1};
    double *hkey = s.hkey + (size_t)b * NB * H * ds, *hval = s.hval + (size_t)b * NB * H * RM;
    double *ckey = s.ckey + (size_t)b * NB * HC * ds, *cJ = s.cJ + (size_t)b * NB * HC * RM * ds, *cb = s.cb + (size_t)b * NB * HC * RM;
    int *hn = s.hn + (size_t)b * NB, *cn = s.cn + (size_t)b * NB;
    for (int t = tid; t < NB; t += SCO_BLOCK) {
      const double *xb = x + (t < NBt ? t : T - 1) * d;    // block NBt (reach equality rows) lives on the last timestep
      ev_hit[t] = p.memo ? memo_find(hkey + (size_t)t * H * ds, hn[t], ds, xb) : -1;
      cv_hit[t] = p.memo ? memo_find(ckey + (size_t)t * HC * ds, cn[t], ds, xb) : -1;
    }
    __syncthreads();
    // S1: f(x) per row, memoised on the rounded point (expr.py:34-41)
    for (int e = tid; e < m_nl; e += SCO_BLOCK) {
      const RowRef q = row_ref(e, L);
      double g;
      if (ev_hit[q.blk] >= 0) g = hval[((size_t)q.blk * H + ev_hit[q.blk]) * RM + q.r];
      else {
        g = row_value(rc, q, x + q.t * d, -1, 0.0);
        if (p.memo && hn[q.blk] < H) hval[((size_t)q.blk * H + hn[q.blk]) * RM + q.r] = g;
      }
      gs[e] = g;
    }
    // S1/S2: Jacobian entry per thread (expr.py:61-69 numeric / :88 analytic); a block whose
    // rounded point was convexified before reuses that affine model (expr.py:362-365, 323-332)
    for (int e = tid; e < m_nl * ds; e += SCO_BLOCK) {
      const int j = e % ds;
      const RowRef q = row_ref(e / ds, L);
      const double *th = x + q.t * d;
      double val;
      if (q.eq == 1 && j >= d) {
        val = 0.0;                               // (a reach row lives on ONE timestep; its entry slots beyond dof do not exist)
      } else if (cv_hit[q.blk] >= 0) {
        val = cJ[(((size_t)q.blk * HC + cv_hit[q.blk]) * RM + q.r) * ds + j];
      } else if (s.analytic_jac) {
        val = row_grad(rc, q, th, j);
      } else {
        // central differences on a halving ladder + Richardson extrapolation
        const double h0 = FD_BASE * fmax(1.0, fabs(th[j]));
        double tab[FD_LEVELS];
#pragma unroll
        for (int lv = 0; lv < FD_LEVELS; lv++) {
          const double h = h0 / (double)(1 << lv);
          const double fp = row_value(rc, q, th, j, h);
          const double fm = row_value(rc, q, th, j, -h);
          tab[lv] = (fp - fm) / (2.0 * h);
        }
        double p4 = 4.0;
#pragma unroll
        for (int i = 1; i < FD_LEVELS; i++) {
          const double fac = 1.0 / (p4 - 1.0);
#pragma unroll
          for (int lv = FD_LEVELS - 1; lv >= i; lv--) tab[lv] = tab[lv] + (tab[lv] - tab[lv - 1]) * fac;
          p4 *= 4.0;
        }
        val = tab[FD_LEVELS - 1];
      }
      J[e] = val;
      if (p.memo && cv_hit[q.blk] < 0 && cn[q.blk] < HC) cJ[(((size_t)q.blk * HC + cn[q.blk]) * RM + q.r) * ds + j] = val;
      if (!spawned) mask[e] = (val != 0.0) ? 1 : 0;   // creation-time pattern (prob.py:264, 440)
    }
    __syncthreads();
    // S2: affine model b = f - J x - val (expr.py:141, 367 / 328);  S3: rows
    if (p.duplicate_rows) k_rows += 1; else k_rows = 1;
    slack_cost = p.compound_penalty ? slack_cost * penalty : penalty;
    double *Av = q1.Aval + (size_t)b * q1.nnzA;
    double *l = q1.l + (size_t)b * m, *u = q1.u + (size_t)b * m;
    int *w = q1.w + (size_t)b * m;
    for (int e = tid; e < m_nl; e += SCO_BLOCK) {
      const RowRef q = row_ref(e, L);
      const int nj = q.eq == 1 ? d : ds;
      double acc = 0.0;
      for (int j = 0; j < nj; j++) acc += J[(size_t)e * ds + j] * x[q.t * d + j];
      double bb = (gs[e] - acc) - row_rhs(rc, q);
      if (cv_hit[q.blk] >= 0) bb = cb[((size_t)q.blk * HC + cv_hit[q.blk]) * RM + q.r];
      else if (p.memo && cn[q.blk] < HC) cb[((size_t)q.blk * HC + cn[q.blk]) * RM + q.r] = bb;
      bm[e] = bb;
      u[s.m_lin + e] = -bb;            // hinge: -inf <= a x - t <= -b  (prob.py:265-275, 486)
      if (q.eq) l[s.m_lin + e] = -bb;  // abs:   a x - p + n = -b        (prob.py:303-312, 480-484)
      w[s.m_lin + e] = k_rows;         // row present k times (prob.py:508-509)
    }
    for (int e = tid; e < m_nl * ds; e += SCO_BLOCK) {
      const int j = e % ds;
      const RowRef q = row_ref(e / ds, L);
      if (q.eq == 1 && j >= d) continue;
      // column c = (timestep q.t + j / d, coordinate j % d) holds, below its linear entries, the R slots of every block
      // that covers its timestep, in block order (sco_sqp_create)
      const int c = q.t * d + j, tfirst = (c / d - s.S + 1 > 0) ? c / d - s.S + 1 : 0;
      const int pos = q.eq == 1 ? s.epos[j] + q.r : s.jpos[c] + (q.blk - tfirst) * R + q.r;
      Av[pos] = mask[e] ? J[e] : 0.0;   // prob.py:493-504
    }
    double *qv = q1.q + (size_t)b * n;
    for (int i = tid; i < s.n_slack; i += SCO_BLOCK) qv[n_x + i] = slack_cost;   // prob.py:424-426
    if (s.cost) {
      // ---- non-quadratic objective terms: Expr.convexify(degree 2) (expr.py:143-153) per timestep block
      const ObjCtx oc{s.cost, len, d, s.ctgt[(size_t)b * 2], s.ctgt[(size_t)b * 2 + 1], s.cw[b]};
      double *oH = s.oH + (size_t)b * T * d * d, *oA = s.oA + (size_t)b * T * d, *ob = s.ob + (size_t)b * T;
      // f(x), memoised on the rounded point like every Expr.eval (expr.py:34-41); the term shares the point history of
      // its timestep's constraint block (same Variable, same evaluation points): its value is the block's last column
      for (int t = tid; t < T; t += SCO_BLOCK) {
        double f;
        if (ev_hit[t] >= 0) f = hval[((size_t)t * H + ev_hit[t]) * RM + (RM - 1)];
        else {
          f = obj_value(oc, rc, t, x + t * d, -1, 0.0, -1, 0.0);
          if (p.memo && hn[t] < H) hval[((size_t)t * H + hn[t]) * RM + (RM - 1)] = f;
        }
        of0[t] = f;
      }
      __syncthreads();
      // numeric Hessian (expr.py:102-109; second central differences on the halving ladder, Richardson: numdiff.py)
      const int npair = d * (d + 1) / 2;
      for (int e = tid; e < T * npair; e += SCO_BLOCK) {
        const int t = e / npair;
        int k = e % npair, i = 0;
        while (k >= d - i) { k -= d - i; i++; }
        const int j = i + k;
        const double *th = x + t * d;
        const double si = FD_BASE * fmax(1.0, fabs(th[i])), sj = FD_BASE * fmax(1.0, fabs(th[j]));
        double tab[FD_LEVELS];
#pragma unroll
        for (int lv = 0; lv < FD_LEVELS; lv++) {
          const double hi = si / (double)(1 << lv), hj = sj / (double)(1 << lv);
          if (i == j) {
            const double fp = obj_value(oc, rc, t, th, i, hi, -1, 0.0);
            const double fm = obj_value(oc, rc, t, th, i, -hi, -1, 0.0);
            tab[lv] = (fp - 2.0 * of0[t] + fm) / (hi * hi);
          } else {
            const double fpp = obj_value(oc, rc, t, th, i, hi, j, hj);
            const double fpm = obj_value(oc, rc, t, th, i, hi, j, -hj);
            const double fmp = obj_value(oc, rc, t, th, i, -hi, j, hj);
            const double fmm = obj_value(oc, rc, t, th, i, -hi, j, -hj);
            tab[lv] = (fpp - fpm - fmp + fmm) / (4.0 * hi * hj);
          }
        }
        const double hv = richardson(tab);
        oH[(size_t)t * d * d + i * d + j] = hv; oH[(size_t)t * d * d + j * d + i] = hv;
      }
      // numeric gradient (expr.py:61-69), parked in oA until the block's thread turns it into the model's A
      for (int e = tid; e < T * d; e += SCO_BLOCK) {
        const int t = e / d, j = e % d;
        const double *th = x + t * d;
        const double h0 = FD_BASE * fmax(1.0, fabs(th[j]));
        double tab[FD_LEVELS];
#pragma unroll
        for (int lv = 0; lv < FD_LEVELS; lv++) {
          const double h = h0 / (double)(1 << lv);
          tab[lv] = (obj_value(oc, rc, t, th, j, h, -1, 0.0) - obj_value(oc, rc, t, th, j, -h, -1, 0.0)) / (2.0 * h);
        }
        oA[e] = richardson(tab);
      }
      __syncthreads();
      // eigenvalue shift, then Q = H, A = g - x'H, b = 1/2 x'Hx - g.x + f (expr.py:145-152)
      for (int t = tid; t < T; t += SCO_BLOCK) {
        double *Ht = oH + (size_t)t * d * d;
        const double *th = x + t * d;
        const double lam = min_eig_jacobi(Ht, d);
        if (lam < 0.0) for (int i = 0; i < d; i++) Ht[i * d + i] -= lam;
        double xHx = 0.0, gx = 0.0, g[OBJ_DMAX], xH[OBJ_DMAX];
        for (int j = 0; j < d; j++) {
          g[j] = oA[t * d + j];
          double acc = 0.0;
          for (int i = 0; i < d; i++) acc += th[i] * Ht[i * d + j];
          xH[j] = acc;
        }
        for (int j = 0; j < d; j++) { xHx += xH[j] * th[j]; gx += g[j] * th[j]; oA[t * d + j] = g[j] - xH[j]; }
        ob[t] = (0.5 * xHx - gx) + of0[t];
      }
      __syncthreads();
      // QuadExpr lowering (prob.py:348-367; osqp_utils.py:153-163): Q into the upper triangle of P, A into q
      double *Pv = q1.Pval + (size_t)b * q1.nnzP;
      for (int e = tid; e < T * npair; e += SCO_BLOCK) {
        const int t = e / npair;
        int k = e % npair, i = 0;
        while (k >= d - i) { k -= d - i; i++; }
        const int j = i + k;
        const double base = (i == j) ? obj_p_diag(t, T, s.objw ? s.objw[(size_t)b * d + i] : 1.0, s.accw ? s.accw[(size_t)b * d + i] : 0.0) : 0.0;
        Pv[s.ppos[t * d + j] + i] = base + oH[(size_t)t * d * d + i * d + j];
      }
      for (int e = tid; e < n_x; e += SCO_BLOCK) qv[e] = oA[e];
    }
    // S7: merit at the convexification point (prob.py:571-579), S4 prerequisite: save
    double v[2] = {traj_obj_partial(x, d, T, tid, s.objw ? s.objw + (size_t)b * d : nullptr, s.accw ? s.accw + (size_t)b * d : nullptr) + ((s.cost && tid < T) ? of0[tid] : 0.0), 0.0};
    for (int e = tid; e < m_nl; e += SCO_BLOCK) {
      const RowRef q = row_ref(e, L);
      v[1] += row_viol(q, gs[e] - row_rhs(rc, q));
    }
    block_reduce_sm<2, 0>(v, red);
    if (s.G > 0) {
      // get_value(vectorize=True): per-block sums, then per-group sums in block order (prob.py:558-570)
      __shared__ double bsum[260];
      for (int blk = tid; blk < NB; blk += SCO_BLOCK) {
        const int e0 = blk < NBt ? blk * R : NBt * R, cnt = blk < NBt ? R : s.NE;
        double acc = 0.0;
        for (int e = e0; e < e0 + cnt; e++) { const RowRef q = row_ref(e, L); acc += row_viol(q, gs[e] - row_rhs(rc, q)); }
        bsum[blk] = acc;
      }
      __syncthreads();
      for (int g = tid; g < s.G; g += SCO_BLOCK) {
        double acc = 0.0;
        for (int blk = 0; blk < NB; blk++) if ((s.gmask[blk] >> g) & 1u) acc += bsum[blk];
        s.mvec[(size_t)b * 32 + g] = acc;
      }
      __syncthreads();
    }
    for (int i = tid; i < n_x; i += SCO_BLOCK) xs[i] = x[i];
    // commit the new history entries (keys last, after every value has been written)
    if (p.memo)
      for (int t = tid; t < NB; t += SCO_BLOCK) {
        const double *xb = x + (t < NBt ? t : T - 1) * d;
        if (ev_hit[t] < 0 && hn[t] < H) {
          for (int j = 0; j < ds; j++) hkey[((size_t)t * H + hn[t]) * ds + j] = rint(xb[j] * 1e6);
          hn[t] += 1;
        } else if (ev_hit[t] < 0) atomicOr(&sc.flags, SCO_SQP_FLAG_MEMO_FULL);
        if (cv_hit[t] < 0 && cn[t] < HC) {
          for (int j = 0; j < ds; j++) ckey[((size_t)t * HC + cn[t]) * ds + j] = rint(xb[j] * 1e6);
          cn[t] += 1;
        } else if (cv_hit[t] < 0) atomicOr(&sc.flags, SCO_SQP_FLAG_MEMO_FULL);
      }
    if (tid == 0) {
      sc.merit_viol = v[1];
      sc.merit = v[0] + penalty * v[1];
      sc.k = k_rows; sc.slack_cost = slack_cost; sc.spawned = 1;
      sc.sqp_iters += 1;
      sc.state = ST_TRIAL;
    }
    __syncthreads();
  }
  // S4: trust box around the saved point (variable.py:43-45)
  {
    double *l = q1.l + (size_t)b * m, *u = q1.u + (size_t)b * m;
    const int base = s.m_lin + s.m_nl;
    for (int i = tid; i < n_x; i += SCO_BLOCK) { l[base + i] = xs[i] - trust; u[base + i] = xs[i] + trust; }
  }
}

// --------------------------------------------------------------------------
// sqp_post: model merit, new merit, decision
// --------------------------------------------------------------------------
__global__ __launch_bounds__(SCO_BLOCK) void sqp_post_kernel(SqpDev s, QpDev q1, SqpParamsDev p) {
  const int g = blockIdx.x + s.b0, tid = threadIdx.x;
  const int b = s.list ? s.list[g] : g;
  if (b < 0) return;
  SqpScalars &sc = s.sc[b];
  if (sc.state != ST_TRIAL) return;
  if (q1.prog && q1.prog[b] > 0) {             // its QP is parked between two ADMM slices: nothing to decide yet
    if (tid == 0) atomicAdd(s.n_active, 1);
    return;
  }
  __shared__ double red[NWAVE * 6];
  const int n_x = s.n_x, d = s.d, T = s.T, R = s.R, O = s.O, n = s.n;
  const int ds = s.ds, NBt = s.NBt;
  const RowLay L{T, NBt, R, s.Req};
  double *x = s.x + (size_t)b * n_x, *xs = s.x_saved + (size_t)b * n_x;
  const double *len = s.link_len + (size_t)b * d;
  const double *obs = s.obstacles + (size_t)b * O * 3;
  const double *gs = s.gsave + (size_t)b * s.m_nl, *J = s.J + (size_t)b * s.m_nl * ds, *bm = s.bmod + (size_t)b * s.m_nl;
  const int status = q1.status[b], iters = q1.iters[b];
  // every scalar is read BEFORE the reduction's barriers; thread 0 rewrites them afterwards
  const double pen = sc.penalty, trust = sc.trust, merit = sc.merit, merit_viol0 = sc.merit_viol;
  const int qp_solves0 = sc.qp_solves;
  const bool ok = (status == 1 || status == 2);                 // prob.py:197
  const double *xq = ok ? (q1.x + (size_t)b * n) : xs;          // failed QP leaves the variables alone
  // Q3: blocks of the trial point that round onto an already-seen point reuse its f values
  __shared__ int ev_hit[260];
  const int H = s.H, NB = s.NB, RM = s.RM, m_nl = s.m_nl;
  const RowCtx rc{len, obs, s.target + (size_t)b * 2, s.point_link, s.point_frac, ds, O, s.point,
                  s.qQ + (size_t)b * O * ds * ds, s.qa + (size_t)b * O * ds, s.qc + (size_t)b * O,
                  s.pw, s.pptr, s.pconst, s.ppar + (size_t)b * s.n_par * (s.par_step ? s.T : 1), s.par_step, s.R1};
  double *hkey = s.hkey + (size_t)b * NB * H * ds, *hval = s.hval + (size_t)b * NB * H * RM;
  int *hn = s.hn + (size_t)b * NB;
  for (int t = tid; t < NB; t += SCO_BLOCK)
    ev_hit[t] = p.memo ? memo_find(hkey + (size_t)t * H * ds, hn[t], ds, xq + (t < NBt ? t : T - 1) * d) : -1;
  __syncthreads();
  // model violation uses the FULL Jacobian (prob.py:627-628), new violation f at the new point (prob.py:575-577)
  // v: quadratic objective, model violation, new violation, objective MODELS at the new point (prob.py:625-626),
  //    objective terms at the new point (prob.py:571-573), max violation at the saved point
  double v[6] = {traj_obj_partial(xq, d, T, tid, s.objw ? s.objw + (size_t)b * d : nullptr, s.accw ? s.accw + (size_t)b * d : nullptr), 0.0, 0.0, 0.0, 0.0, 0.0};
  if (s.cost) {
    const double *oH = s.oH + (size_t)b * T * d * d, *oA = s.oA + (size_t)b * T * d, *ob = s.ob + (size_t)b * T;
    for (int t = tid; t < T; t += SCO_BLOCK) {
      const double *th = xq + t * d, *Ht = oH + (size_t)t * d * d;
      double xHx = 0.0, ax = 0.0;
      for (int i = 0; i < d; i++) {
        double acc = 0.0;
        for (int j = 0; j < d; j++) acc += Ht[i * d + j] * th[j];
        xHx += th[i] * acc; ax += oA[t * d + i] * th[i];
      }
      v[3] += (0.5 * xHx + ax) + ob[t];                        // QuadExpr.eval (expr.py:205-206)
      double f;
      if (ev_hit[t] >= 0) f = hval[((size_t)t * H + ev_hit[t]) * RM + (RM - 1)];
      else {
        f = obj_value(ObjCtx{s.cost, len, d, s.ctgt[(size_t)b * 2], s.ctgt[(size_t)b * 2 + 1], s.cw[b]}, rc, t, th, -1, 0.0, -1, 0.0);
        if (p.memo && hn[t] < H) hval[((size_t)t * H + hn[t]) * RM + (RM - 1)] = f;
      }
      v[4] += f;
    }
  }
  for (int e = tid; e < m_nl; e += SCO_BLOCK) {
    const RowRef q = row_ref(e, L);
    const double rhs = row_rhs(rc, q);
    const int nj = q.eq == 1 ? d : ds;
    double acc = 0.0;
    for (int j = 0; j < nj; j++) acc += J[(size_t)e * ds + j] * xq[q.t * d + j];
    v[1] += row_viol(q, acc + bm[e]);
    double g;
    if (ev_hit[q.blk] >= 0) g = hval[((size_t)q.blk * H + ev_hit[q.blk]) * RM + q.r];
    else {
      g = row_value(rc, q, xq + q.t * d, -1, 0.0);
      if (p.memo && hn[q.blk] < H) hval[((size_t)q.blk * H + hn[q.blk]) * RM + q.r] = g;
    }
    v[2] += row_viol(q, g - rhs);
    v[5] = fmax(v[5], row_viol(q, gs[e] - rhs));                 // max violation at the SAVED point
  }
  __syncthreads();
  if (p.memo) {
    for (int t = tid; t < NB; t += SCO_BLOCK) {
      if (ev_hit[t] < 0 && hn[t] < H) {
        const double *xb = xq + (t < NBt ? t : T - 1) * d;
        for (int j = 0; j < ds; j++) hkey[((size_t)t * H + hn[t]) * ds + j] = rint(xb[j] * 1e6);
        hn[t] += 1;
      } else if (ev_hit[t] < 0) {
        atomicOr(&sc.flags, SCO_SQP_FLAG_MEMO_FULL);
      }
    }
  }
  block_reduce_sm<5, 1>(v, red);
  // constraint groups: which violated groups stopped improving, and do their overlapping groups too
  // (solver.py:155-161, 209-235)
  __shared__ unsigned int g_stalled, g_report;
  if (s.G > 0) {
    __shared__ double bsum[260], gimp[32];
    for (int blk = tid; blk < NB; blk += SCO_BLOCK) {
      const int e0 = blk < NBt ? blk * R : NBt * R, cnt = blk < NBt ? R : s.NE;
      double acc = 0.0;
      for (int e = e0; e < e0 + cnt; e++) {
        const RowRef q = row_ref(e, L);
        const int nj = q.eq == 1 ? d : ds;
        double ax = 0.0;
        for (int j = 0; j < nj; j++) ax += J[(size_t)e * ds + j] * xq[q.t * d + j];
        acc += row_viol(q, ax + bm[e]);
      }
      bsum[blk] = acc;
    }
    __syncthreads();
    for (int g = tid; g < s.G; g += SCO_BLOCK) {
      double acc = 0.0;
      for (int blk = 0; blk < NB; blk++) if ((s.gmask[blk] >> g) & 1u) acc += bsum[blk];
      gimp[g] = s.mvec[(size_t)b * 32 + g] - acc;         // approx_improve_vec
    }
    __syncthreads();
    if (tid == 0) {
      unsigned int stalled = 0, report = 0;
      for (int g = 0; g < s.G; g++) {
        const bool viol_g = s.mvec[(size_t)b * 32 + g] > p.cnt_tolerance;
        if (!(viol_g && gimp[g] < p.min_approx_improve)) continue;
        report |= 1u << g;
        bool overlap_improve = false;
        for (int h = 0; h < s.G; h++)
          if (((s.goverlap[g] >> h) & 1u) && gimp[h] > p.min_approx_improve) overlap_improve = true;
        if (!overlap_improve) stalled |= 1u << g;
      }
      g_stalled = stalled; g_report = report;
    }
    __syncthreads();
  }
  const double model_merit = (v[0] + v[3]) + pen * v[1], new_merit = (v[0] + v[4]) + pen * v[2];
  double approx = merit - model_merit;
  if (approx == 0.0) approx += 1e-12;                           // solver.py:152-153
  const double exact = merit - new_merit;
  const double ratio = exact / approx;
  const double approx_vec = merit_viol0 - v[1];               // single group "all" (prob.py:135-136)
  const bool violated = merit_viol0 > p.cnt_tolerance;
  int kind, ret = -1;   // ret: -1 continue, 0/1 = _min_merit_fn returned False/True
  double new_trust = trust;
  if (approx < -1e-5) { kind = STEP_BAD; ret = 0; }                               // solver.py:185-198
  else if (approx < p.min_approx_improve) { kind = STEP_YCONV; ret = 1; }         // solver.py:200-204
  else if (s.G > 0 ? g_stalled != 0u : (violated && approx_vec < p.min_approx_improve)) { kind = STEP_GROUP; ret = 1; }   // solver.py:209-235
  else if (exact < 0.0 || ratio < p.improve_ratio_threshold) {                   // solver.py:237-241
    kind = STEP_SHRINK; new_trust = trust * p.trust_shrink_ratio;
    if (new_trust < p.min_trust_region_size) { kind = STEP_XCONV; ret = 1; }      // solver.py:248-251
  } else { kind = STEP_ACCEPT; new_trust = trust * p.trust_expand_ratio; }        // solver.py:242-246
  const int qp_solves = qp_solves0 + 1;
  const bool capped = (ret < 0) && (qp_solves >= p.max_qp_solves);
  if (kind == STEP_ACCEPT) for (int i = tid; i < n_x; i += SCO_BLOCK) x[i] = xq[i];
  // every other outcome restores the saved point (solver.py:197, 203, 229, 238); x == xs already
  if (tid == 0) {
    // prob.nonconverged_groups is rewritten by every trust-region trial that gets past the
    // y-convergence test (solver.py:209, 233-235)
    if (kind != STEP_BAD && kind != STEP_YCONV)
    {
      s.nonconv[b] = (kind == STEP_GROUP) ? (s.G > 0 ? g_report : 1u) : 0u;
      s.stalled[b] = (kind == STEP_GROUP) ? (s.G > 0 ? g_stalled : 1u) : 0u;
    }
    sc.qp_solves = qp_solves; sc.admm_iters += iters;
    if (capped) atomicOr(&sc.flags, SCO_SQP_FLAG_CAPPED);
    if (sc.n_trace >= s.trace_cap) atomicOr(&sc.flags, SCO_SQP_FLAG_TRACE_FULL);
    trace_row(s, b, sc, kind, merit, model_merit, new_merit, trust, pen, status, iters);
    sc.trust = new_trust;
    if (ret < 0 && !capped) {
      sc.state = (kind == STEP_ACCEPT) ? ST_CONVEXIFY : ST_TRIAL;
      atomicAdd(s.n_active, 1);
    } else {
      // _min_merit_fn returned: outer loop of _penalty_sqp (solver.py:84-105)
      const double max_viol = v[5];
      sc.escalations += 1;
      if (!capped && max_viol > p.cnt_tolerance && sc.escalations < p.max_merit_coeff_increases) {
        sc.penalty = pen * p.merit_coeff_increase_ratio;
        sc.trust = p.initial_trust_region_size;
        sc.state = ST_CONVEXIFY;
        atomicAdd(s.n_active, 1);
      } else {
        sc.success = (!capped && max_viol <= p.cnt_tolerance) ? ret : 0;
        sc.state = ST_DONE; s.active[b] = 0;
      }
    }
  }
}

// final merit / max violation at the returned point (prob.py:571-603)
__global__ __launch_bounds__(SCO_BLOCK) void sqp_final_kernel(SqpDev s, double *merit_out, double *viol_out) {
  const int b = blockIdx.x, tid = threadIdx.x;
  __shared__ double red[NWAVE * 4];
  const int n_x = s.n_x, d = s.d, T = s.T, R = s.R, O = s.O;
  const double *x = s.x + (size_t)b * n_x;
  const double *len = s.link_len + (size_t)b * d;
  const double *obs = s.obstacles + (size_t)b * O * 3;
  const RowCtx rc{len, obs, s.target + (size_t)b * 2, s.point_link, s.point_frac, s.ds, O, s.point,
                  s.qQ + (size_t)b * O * s.ds * s.ds, s.qa + (size_t)b * O * s.ds, s.qc + (size_t)b * O,
                  s.pw, s.pptr, s.pconst, s.ppar + (size_t)b * s.n_par * (s.par_step ? s.T : 1), s.par_step, s.R1};
  const RowLay L{T, s.NBt, R, s.Req};
  double v[3] = {traj_obj_partial(x, d, T, tid, s.objw ? s.objw + (size_t)b * d : nullptr, s.accw ? s.accw + (size_t)b * d : nullptr), 0.0, 0.0};
  if (s.cost)
    for (int t = tid; t < T; t += SCO_BLOCK)
      v[0] += obj_value(ObjCtx{s.cost, len, d, s.ctgt[(size_t)b * 2], s.ctgt[(size_t)b * 2 + 1], s.cw[b]}, rc, t, x + t * d, -1, 0.0, -1, 0.0);
  for (int e = tid; e < s.m_nl; e += SCO_BLOCK) {
    const RowRef q = row_ref(e, L);
    const double g = row_viol(q, row_value(rc, q, x + q.t * d, -1, 0.0) - row_rhs(rc, q));
    v[1] += g; v[2] = fmax(v[2], g);
  }
  block_reduce_sm<2, 1>(v, red);
  if (tid == 0) { merit_out[b] = v[0] + s.sc[b].penalty * v[1]; viol_out[b] = v[2]; }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
extern "C" void sco_sqp_default_params(sco_sqp_params *p) {
  if (!p) return;
  p->improve_ratio_threshold = 0.25; p->min_trust_region_size = 1e-4; p->min_approx_improve = 1e-8;
  p->trust_shrink_ratio = 0.1; p->trust_expand_ratio = 1.5; p->cnt_tolerance = 1e-4;
  p->merit_coeff_increase_ratio = 10.0; p->initial_trust_region_size = 1.0; p->initial_penalty_coeff = 1e3;
  p->max_merit_coeff_increases = 1; p->compound_penalty = 1; p->duplicate_rows = 1; p->max_sqp_iters = 0;
  p->memoize_rounded = 1; p->warm_start_qps = 0; p->admm_slice = 0;
}

template <typename T>
static int sq_alloc(sco_sqp *h, size_t count, T **out) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  SCO_HIP(hipMalloc(&p, bytes));
  SCO_HIP(hipMemset(p, 0, bytes));
  h->allocs.push_back(p);
  *out = (T *)p;
  return SCO_OK;
}

static int sqp_create_impl(sco_sqp *h, int device, const sco_trajopt_desc *desc);

// r04: sco_sqp_create with n_rows GENERAL affine rows over the trajectory variables behind the built-in linear rows -- what a
// caller of the reference adds with prob.add_cnt_expr(BoundExpr(EqExpr / LEqExpr(AffExpr(A, b), val), traj)) (prob.py:126-131,
// 317-346): a shared CSR pattern (row_ptr[n_rows + 1], col_idx strictly increasing inside a row, columns = name-sorted
// trajectory atoms t * dof + j), row_is_eq[r] != 0: a x = rhs, else a x <= rhs; coefficients and right-hand sides per problem
// through sco_sqp_load_linear_rows.
extern "C" int sco_sqp_create_rows(int device, const sco_trajopt_desc *desc, int n_rows, const int *row_ptr, const int *col_idx,
                                   const int *row_is_eq, sco_sqp **out) {
  if (!desc || !out) { sco_set_error("sco_sqp_create: null pointer"); return SCO_ERR_ARG; }
  if (n_rows < 0 || n_rows > 65536 || (n_rows > 0 && (!row_ptr || !col_idx || !row_is_eq))) { sco_set_error("sco_sqp_create_rows: bad row pattern"); return SCO_ERR_ARG; }
  if (n_rows > 0) {
    if (row_ptr[0] != 0) { sco_set_error("sco_sqp_create_rows: bad row pattern"); return SCO_ERR_ARG; }
    const long long nx = (long long)desc->dof * desc->horizon;
    for (int r = 0; r < n_rows; r++) {
      if (row_ptr[r + 1] <= row_ptr[r] || row_ptr[r + 1] - row_ptr[r] > nx) { sco_set_error("sco_sqp_create_rows: bad row pattern (empty row or row_ptr not increasing)"); return SCO_ERR_ARG; }
      for (int k = row_ptr[r]; k < row_ptr[r + 1]; k++)
        if (col_idx[k] < 0 || col_idx[k] >= nx || (k > row_ptr[r] && col_idx[k] <= col_idx[k - 1])) {
          sco_set_error("sco_sqp_create_rows: column indices must lie in [0, dof * horizon) and increase inside a row"); return SCO_ERR_ARG;
        }
    }
  }
  const int fam = desc->family & 15, span = desc->span > 0 ? desc->span : 1;
  const bool statefam = fam == SCO_FAM_STATE_QUADRATIC || fam == SCO_FAM_STATE_PROGRAM;
  if (desc->batch <= 0 || desc->dof <= 0 || desc->horizon < 2 || desc->n_points <= 0 || desc->n_obstacles <= 0 ||
      desc->horizon > 256 || (desc->family & ~(15 | SCO_FAM_FLAG_VEL_LIMITS | SCO_FAM_FLAG_JOINT_LIMITS | SCO_FAM_FLAG_EE_COST | SCO_FAM_FLAG_OBJ_PROGRAM | SCO_FAM_FLAG_ACC_COST)) ||
      ((desc->family & SCO_FAM_FLAG_EE_COST) && desc->dof > OBJ_DMAX) || ((desc->family & SCO_FAM_FLAG_ACC_COST) && desc->horizon < 3) ||
      (fam != SCO_FAM_ARM_CIRCLES && fam != SCO_FAM_ARM_REACH && fam != SCO_FAM_POINT_CIRCLES && !statefam) ||
      (fam == SCO_FAM_POINT_CIRCLES && (desc->n_points != 1 || desc->dof < 2 || (desc->family & SCO_FAM_FLAG_EE_COST))) ||
      (statefam && (desc->n_points != 1 || (desc->family & SCO_FAM_FLAG_EE_COST))) ||
      (fam == SCO_FAM_STATE_QUADRATIC && desc->dof > OBJ_DMAX) ||
      // blocks of `span` timesteps, equality rows and objective programs: the state families' extensions (r03)
      desc->span < 0 || desc->span > 4 || desc->n_eq_rows < 0 || desc->n_eq_rows > desc->n_obstacles ||
      (span > 1 && fam != SCO_FAM_STATE_PROGRAM) || (desc->n_eq_rows > 0 && !statefam) ||
      (fam == SCO_FAM_STATE_PROGRAM && (span * desc->dof > SCO_STATE_MAX || span >= desc->horizon)) ||
      ((desc->family & SCO_FAM_FLAG_OBJ_PROGRAM) && (fam != SCO_FAM_STATE_PROGRAM || span != 1 || desc->dof > OBJ_DMAX))) {
    sco_set_error("sco_sqp_create: bad descriptor"); return SCO_ERR_ARG;
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    sco_set_error("sco_sqp_create: no HIP device visible (this library has no CPU fallback)");
    return SCO_ERR_NO_GPU;
  }
  if (device < 0 || device >= ndev) { sco_set_error("sco_sqp_create: bad device index"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(device);
  sco_sqp *h = new sco_sqp();
  h->device = device; h->desc = *desc;
  if (n_rows > 0) {
    h->gen_ptr.assign(row_ptr, row_ptr + n_rows + 1); h->gen_col.assign(col_idx, col_idx + row_ptr[n_rows]);
    h->gen_iseq.resize(n_rows);
    for (int r = 0; r < n_rows; r++) h->gen_iseq[r] = row_is_eq[r] ? 1 : 0;
  }
  const int rc = sqp_create_impl(h, device, desc);
  if (rc) { sco_sqp_destroy(h); return rc; }       // frees whatever had been allocated
  *out = h;
  return SCO_OK;
}

extern "C" int sco_sqp_create(int device, const sco_trajopt_desc *desc, sco_sqp **out) {
  return sco_sqp_create_rows(device, desc, 0, nullptr, nullptr, nullptr, out);
}

static int sqp_create_impl(sco_sqp *h, int device, const sco_trajopt_desc *desc) {
  SCO_HIP(hipStreamCreate(&h->stream));
  SCO_HIP(hipHostMalloc((void **)&h->host_active, SQP_MAX_GROUPS * SQP_DEPTH * sizeof(int)));
  {
    void *fb = nullptr;
    SCO_HIP(hipMalloc(&fb, 2 * (size_t)desc->batch * sizeof(double)));
    h->allocs.push_back(fb); h->fetch_buf = (double *)fb;
  }
  const int B = desc->batch, d = desc->dof, T = desc->horizon, K = desc->n_points, O = desc->n_obstacles;
  const bool reach = (desc->family & 15) == SCO_FAM_ARM_REACH, vel = (desc->family & SCO_FAM_FLAG_VEL_LIMITS) != 0;
  const bool jl = (desc->family & SCO_FAM_FLAG_JOINT_LIMITS) != 0;
  const bool cost = (desc->family & (SCO_FAM_FLAG_EE_COST | SCO_FAM_FLAG_OBJ_PROGRAM)) != 0;
  const bool acc = (desc->family & SCO_FAM_FLAG_ACC_COST) != 0;
  const int NE = reach ? 2 : 0;                  // equality rows (end-effector x, y) on the last timestep
  const int S = desc->span > 0 ? desc->span : 1, ds = S * d, NBt = T - S + 1, Req = desc->n_eq_rows;
  const int m_pin = reach ? d : 2 * d, dT1 = d * (T - 1), m_vel = vel ? 2 * dT1 : 0;
  const int m_jl = jl ? 2 * d * T : 0;
  // a hinge row has one slack, an equality row two (prob.py:258, 285-286); block-major, hinge rows of a block first
  // general affine rows (r04): column lists of the CSR pattern; entry k of the pattern = (row, column) in CSR order
  const int m_gen = (int)h->gen_iseq.size(), nnz_gen = m_gen ? h->gen_ptr[m_gen] : 0;
  std::vector<std::vector<std::pair<int, int>>> gen_of_col(d * T);       // (row, k), rows ascending
  for (int r = 0; r < m_gen; r++)
    for (int k = h->gen_ptr[r]; k < h->gen_ptr[r + 1]; k++) gen_of_col[h->gen_col[k]].push_back({r, k});
  std::vector<int> gpos0(std::max(nnz_gen, 1), 0), gpos1(std::max(nnz_gen, 1), 0);
  const int R = K * O, n_x = d * T, SB = R + Req, n_slack = NBt * SB + 2 * NE, n = n_x + n_slack, m_lin = m_pin + m_vel + m_jl + m_gen;
  const int m_nl = NBt * R + NE, m = m_lin + m_nl + n;
  // linear rows of column (t, j), ascending: pin, velocity rows "theta[t] - theta[t-1] <= vmax" (+1),
  // "theta[t+1] - theta[t] <= vmax" (-1), then the two negated rows
  auto linear_entries = [&](int t, int j, std::vector<int> &Ai, std::vector<double> &Av, std::vector<int> &gpos) {
    if (t == 0) { Ai.push_back(j); Av.push_back(1.0); }
    if (t == T - 1 && !reach) { Ai.push_back(d + j); Av.push_back(1.0); }
    if (vel) {
      if (t >= 1) { Ai.push_back(m_pin + (t - 1) * d + j); Av.push_back(1.0); }
      if (t <= T - 2) { Ai.push_back(m_pin + t * d + j); Av.push_back(-1.0); }
      if (t >= 1) { Ai.push_back(m_pin + dT1 + (t - 1) * d + j); Av.push_back(-1.0); }
      if (t <= T - 2) { Ai.push_back(m_pin + dT1 + t * d + j); Av.push_back(1.0); }
    }
    if (jl) {
      Ai.push_back(m_pin + m_vel + t * d + j); Av.push_back(1.0);
      Ai.push_back(m_pin + m_vel + d * T + t * d + j); Av.push_back(-1.0);
    }
    for (const auto &rk : gen_of_col[t * d + j]) {        // values per problem (sco_sqp_load_linear_rows): 0 in the constants
      gpos[rk.second] = (int)Ai.size();
      Ai.push_back(m_pin + m_vel + m_jl + rk.first); Av.push_back(0.0);
    }
  };
  std::vector<double> a0c, a1c;
  // ---- projection QP pattern: P = diag, A = [linear rows ; I]
  {
    std::vector<int> Pp(n_x + 1), Pi(n_x), Ap(n_x + 1), Ai;
    for (int i = 0; i < n_x; i++) { Pp[i] = i; Pi[i] = i; }
    Pp[n_x] = n_x;
    for (int col = 0; col < n_x; col++) {
      Ap[col] = (int)Ai.size();
      const int t = col / d, j = col % d;
      linear_entries(t, j, Ai, a0c, gpos0);
      Ai.push_back(m_lin + col); a0c.push_back(1.0);
    }
    Ap[n_x] = (int)Ai.size();
    int rc = sco_qp_create_on_stream(device, B, n_x, m_lin + n_x, Pp.data(), Pi.data(), Ap.data(), Ai.data(), h->stream, &h->qp0);
    if (rc) return rc;
  }
  // ---- penalty QP pattern (prob.py:251-278 rows, osqp_utils.py:185-189 bound rows)
  std::vector<int> jpos(n_x), epos(d, 0), ppos(n_x, 0);
  {
    std::vector<int> Pp(n + 1), Pi, Ap(n + 1), Ai;
    for (int col = 0; col < n; col++) {
      Pp[col] = (int)Pi.size();
      if (col < n_x) {
        if (acc && col / d > 1) Pi.push_back(col - 2 * d);       // acceleration term (r04): second super-diagonal block
        if (col / d > 0) Pi.push_back(col - d);
        ppos[col] = (int)Pi.size();
        // a non-quadratic objective term fills the upper triangle of its timestep's diagonal block
        if (cost) for (int i = 0; i < col % d; i++) Pi.push_back(col - col % d + i);
        else ppos[col] -= col % d;                 // entry (i, j) sits at ppos + i; only i == j exists
        Pi.push_back(col);
      }
    }
    Pp[n] = (int)Pi.size();
    for (int col = 0; col < n; col++) {
      Ap[col] = (int)Ai.size();
      if (col < n_x) {
        const int t = col / d, j = col % d;
        linear_entries(t, j, Ai, a1c, gpos1);
        jpos[col] = (int)Ai.size();
        // Jacobian slots: the R rows of every block that covers timestep t (blocks t - S + 1 .. t), in block order
        for (int blk = std::max(0, t - S + 1); blk <= std::min(t, NBt - 1); blk++)
          for (int r = 0; r < R; r++) { Ai.push_back(m_lin + blk * R + r); a1c.push_back(0.0); }
        if (t == T - 1 && reach) {
          epos[j] = (int)Ai.size();
          for (int r = 0; r < NE; r++) { Ai.push_back(m_lin + NBt * R + r); a1c.push_back(0.0); }
        }
        Ai.push_back(m_lin + m_nl + col); a1c.push_back(1.0);
      } else {
        // hinge slack i sits in hinge row i (-1); equality row r has p_r (slack T R + 2 r, -1) and n_r (+1)
        // (prob.py:265-275, 303-312); then the slack's bound row
        // slack columns block-major: per block its R - Req hinge slacks, then (p, n) of each of its Req equality rows;
        // behind the blocks (p, n) of the reach rows
        const int sidx = col - n_x;
        int row; double sgn = -1.0;
        if (sidx < NBt * SB) {
          const int blk = sidx / SB, k = sidx % SB, nh = R - Req;
          if (k < nh) row = blk * R + k;
          else { row = blk * R + nh + (k - nh) / 2; if ((k - nh) & 1) sgn = 1.0; }
        } else {
          const int ke = sidx - NBt * SB;
          row = NBt * R + ke / 2; if (ke & 1) sgn = 1.0;
        }
        Ai.push_back(m_lin + row); a1c.push_back(sgn);
        Ai.push_back(m_lin + m_nl + col); a1c.push_back(1.0);
      }
    }
    Ap[n] = (int)Ai.size();
    int rc = sco_qp_create_on_stream(device, B, n, m, Pp.data(), Pi.data(), Ap.data(), Ai.data(), h->stream, &h->qp1);
    if (rc) return rc;
  }
  SqpDev &s = h->d;
  s.batch = B; s.d = d; s.T = T; s.K = K; s.O = O; s.R = R; s.n_x = n_x; s.n_slack = n_slack; s.n = n;
  s.m_lin = m_lin; s.m_nl = m_nl; s.m = m; s.prox_count = desc->prox_count > 0 ? desc->prox_count : 1;
  s.analytic_jac = desc->analytic_jac; s.trace_cap = 64;
  s.point = (desc->family & 15) == SCO_FAM_POINT_CIRCLES ? 1 : (desc->family & 15) == SCO_FAM_STATE_QUADRATIC ? 2 :
            (desc->family & 15) == SCO_FAM_STATE_PROGRAM ? 3 : 0;
  s.NE = NE; s.NB = NBt + (reach ? 1 : 0); s.RM = std::max(R, NE) + (cost ? 1 : 0);     // + the objective term's value
  s.S = S; s.ds = ds; s.NBt = NBt; s.Req = Req;
  s.m_gen = m_gen; s.nnz_gen = nnz_gen;
  s.acc = acc ? 1 : 0;
  s.m_pin = m_pin; s.m_vel = m_vel; s.m_jl = m_jl; s.cost = (desc->family & SCO_FAM_FLAG_OBJ_PROGRAM) ? 2 : cost ? 1 : 0;
  int rc = 0;
#define AL(f, cnt) if ((rc = sq_alloc(h, (cnt), &s.f))) return rc;
  AL(x0, (size_t)B * n_x) AL(start, (size_t)B * d) AL(goal, (size_t)B * d) AL(link_len, (size_t)B * d)
  AL(obstacles, (size_t)B * O * 3) AL(target, (size_t)B * 2) AL(vmax, (size_t)B) AL(jlo, (size_t)B * d) AL(jhi, (size_t)B * d)
  {
    const bool quadf = (desc->family & 15) == SCO_FAM_STATE_QUADRATIC;
    AL(qQ, quadf ? (size_t)B * O * ds * ds : 1) AL(qa, quadf ? (size_t)B * O * ds : 1) AL(qc, quadf ? (size_t)B * O : 1)
  }
  AL(x, (size_t)B * n_x) AL(x_saved, (size_t)B * n_x) AL(gsave, (size_t)B * m_nl) AL(J, (size_t)B * m_nl * ds)
  AL(bmod, (size_t)B * m_nl) AL(trace, (size_t)B * s.trace_cap * TRACE_W) AL(mask, (size_t)B * m_nl * ds)
  AL(sc, (size_t)B) AL(active, (size_t)B) AL(n_active, SQP_MAX_GROUPS) AL(newqp, (size_t)B) AL(list_buf, (size_t)B)
  s.H = 40; s.HC = 24;
  AL(hkey, (size_t)B * s.NB * s.H * ds) AL(hval, (size_t)B * s.NB * s.H * s.RM) AL(ckey, (size_t)B * s.NB * s.HC * ds)
  AL(cJ, (size_t)B * s.NB * s.HC * s.RM * ds) AL(cb, (size_t)B * s.NB * s.HC * s.RM) AL(hn, (size_t)B * s.NB)
  AL(cn, (size_t)B * s.NB)
  AL(cw, (size_t)B) AL(ctgt, (size_t)B * 2)
  AL(oH, cost ? (size_t)B * T * d * d : 1) AL(oA, cost ? (size_t)B * T * d : 1) AL(ob, cost ? (size_t)B * T : 1)
#undef AL
  { int *p; if ((rc = sq_alloc(h, (size_t)n_x, &p))) return rc; s.ppos = p;
    SCO_HIP(hipMemcpy(p, ppos.data(), n_x * sizeof(int), hipMemcpyHostToDevice)); }
  { int *p; if ((rc = sq_alloc(h, (size_t)K, &p))) return rc; s.point_link = p; }
  { double *p; if ((rc = sq_alloc(h, (size_t)K, &p))) return rc; s.point_frac = p; }
  { int *p; if ((rc = sq_alloc(h, (size_t)n_x, &p))) return rc; s.jpos = p;
    SCO_HIP(hipMemcpy(p, jpos.data(), n_x * sizeof(int), hipMemcpyHostToDevice)); }
  { double *p; if ((rc = sq_alloc(h, a0c.size(), &p))) return rc; s.a0c = p;
    SCO_HIP(hipMemcpy(p, a0c.data(), a0c.size() * sizeof(double), hipMemcpyHostToDevice)); }
  { double *p; if ((rc = sq_alloc(h, a1c.size(), &p))) return rc; s.a1c = p;
    SCO_HIP(hipMemcpy(p, a1c.data(), a1c.size() * sizeof(double), hipMemcpyHostToDevice)); }
  { unsigned int *p; if ((rc = sq_alloc(h, (size_t)s.NB, &p))) return rc; s.gmask = p; }
  { unsigned int *p; if ((rc = sq_alloc(h, (size_t)32, &p))) return rc; s.goverlap = p; }
  if ((rc = sq_alloc(h, (size_t)B * 32, &s.mvec))) return rc;
  if ((rc = sq_alloc(h, (size_t)B, &s.nonconv))) return rc;
  if ((rc = sq_alloc(h, (size_t)B, &s.stalled))) return rc;
  s.G = 0;
  { int *p; if ((rc = sq_alloc(h, (size_t)d, &p))) return rc; s.epos = p;
    SCO_HIP(hipMemcpy(p, epos.data(), d * sizeof(int), hipMemcpyHostToDevice)); }
  s.bpos = nullptr;
  s.gen_eq = s.gpos0 = s.gpos1 = nullptr; s.gen_rhs = s.gen_val = nullptr;
  if (m_gen) {
    int *p; double *q;
    if ((rc = sq_alloc(h, (size_t)m_gen, &p))) return rc;
    SCO_HIP(hipMemcpy(p, h->gen_iseq.data(), m_gen * sizeof(int), hipMemcpyHostToDevice)); s.gen_eq = p;
    if ((rc = sq_alloc(h, (size_t)nnz_gen, &p))) return rc;
    SCO_HIP(hipMemcpy(p, gpos0.data(), nnz_gen * sizeof(int), hipMemcpyHostToDevice)); s.gpos0 = p;
    if ((rc = sq_alloc(h, (size_t)nnz_gen, &p))) return rc;
    SCO_HIP(hipMemcpy(p, gpos1.data(), nnz_gen * sizeof(int), hipMemcpyHostToDevice)); s.gpos1 = p;
    if ((rc = sq_alloc(h, (size_t)B * m_gen, &q))) return rc; s.gen_rhs = q;
    if ((rc = sq_alloc(h, (size_t)B * nnz_gen, &q))) return rc; s.gen_val = q;
  }
  return SCO_OK;
}

extern "C" int sco_sqp_destroy(sco_sqp *h) {
  if (!h) return SCO_OK;
  ScoDeviceGuard sco_guard_(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->qp0) sco_qp_destroy(h->qp0);
  if (h->qp1) sco_qp_destroy(h->qp1);
  for (void *p : h->allocs) (void)hipFree(p);
  for (void *p : h->prog_buf) if (p) (void)hipFree(p);
  if (h->objw_buf) (void)hipFree(h->objw_buf);
  if (h->accw_buf) (void)hipFree(h->accw_buf);
  for (auto e : h->events) (void)hipEventDestroy(e);
  for (auto &ge : h->gevents) for (auto e : ge) (void)hipEventDestroy(e);
  for (auto e : h->done) (void)hipEventDestroy(e);
  for (auto gs : h->gstream) if (gs) { (void)hipStreamSynchronize(gs); (void)hipStreamDestroy(gs); }
  if (h->host_active) (void)hipHostFree(h->host_active);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return SCO_OK;
}

extern "C" int sco_sqp_load(sco_sqp *h, const double *x0, const double *start, const double *goal,
                            const double *link_len, const int *point_link, const double *point_frac,
                            const double *obstacles) {
  if (!h || !x0 || !start || !goal || !link_len || !point_link || !point_frac || !obstacles) {
    sco_set_error("sco_sqp_load: null pointer"); return SCO_ERR_ARG;
  }
  const SqpDev &s = h->d;
  for (int k = 0; k < s.K; k++)
    if (point_link[k] < 0 || point_link[k] >= s.d) { sco_set_error("sco_sqp_load: point_link out of range"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  const size_t B = s.batch;
  SCO_HIP(hipMemcpyAsync(s.x0, x0, B * s.n_x * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(s.start, start, B * s.d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(s.goal, goal, B * s.d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(s.link_len, link_len, B * s.d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(s.obstacles, obstacles, B * s.O * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync((void *)s.point_link, point_link, s.K * sizeof(int), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync((void *)s.point_frac, point_frac, s.K * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipStreamSynchronize(h->stream));
  h->loaded = true; h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_load_ee_cost(sco_sqp *h, const double *weight, const double *target) {
  if (!h || !weight || !target) { sco_set_error("sco_sqp_load_ee_cost: null pointer"); return SCO_ERR_ARG; }
  if (!(h->desc.family & SCO_FAM_FLAG_EE_COST)) { sco_set_error("sco_sqp_load_ee_cost: family has no objective term"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_ee_cost: call sco_sqp_load first"); return SCO_ERR_STATE; }
  for (int b = 0; b < h->d.batch; b++)
    if (!(weight[b] >= 0.0)) { sco_set_error("sco_sqp_load_ee_cost: weight must be >= 0"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpyAsync(h->d.cw, weight, (size_t)h->d.batch * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(h->d.ctgt, target, (size_t)h->d.batch * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipStreamSynchronize(h->stream));
  h->cost_loaded = true; h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_load_target(sco_sqp *h, const double *target) {
  if (!h || !target) { sco_set_error("sco_sqp_load_target: null pointer"); return SCO_ERR_ARG; }
  if ((h->desc.family & 15) != SCO_FAM_ARM_REACH) { sco_set_error("sco_sqp_load_target: family has no target"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_target: call sco_sqp_load first"); return SCO_ERR_STATE; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpyAsync(h->d.target, target, (size_t)h->d.batch * 2 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipStreamSynchronize(h->stream));
  h->target_loaded = true; h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_load_vel_limit(sco_sqp *h, const double *vmax) {
  if (!h || !vmax) { sco_set_error("sco_sqp_load_vel_limit: null pointer"); return SCO_ERR_ARG; }
  if (!(h->desc.family & SCO_FAM_FLAG_VEL_LIMITS)) { sco_set_error("sco_sqp_load_vel_limit: family has no velocity limits"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_vel_limit: call sco_sqp_load first"); return SCO_ERR_STATE; }
  for (int b = 0; b < h->d.batch; b++)
    if (!(vmax[b] > 0.0)) { sco_set_error("sco_sqp_load_vel_limit: vmax must be positive"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpyAsync(h->d.vmax, vmax, (size_t)h->d.batch * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipStreamSynchronize(h->stream));
  h->vel_loaded = true; h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_load_joint_limits(sco_sqp *h, const double *lo, const double *hi) {
  if (!h || !lo || !hi) { sco_set_error("sco_sqp_load_joint_limits: null pointer"); return SCO_ERR_ARG; }
  if (!(h->desc.family & SCO_FAM_FLAG_JOINT_LIMITS)) { sco_set_error("sco_sqp_load_joint_limits: family has no joint limits"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_joint_limits: call sco_sqp_load first"); return SCO_ERR_STATE; }
  const size_t cnt = (size_t)h->d.batch * h->d.d;
  for (size_t k = 0; k < cnt; k++)
    if (!(lo[k] < hi[k])) { sco_set_error("sco_sqp_load_joint_limits: need lo < hi"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpyAsync(h->d.jlo, lo, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(h->d.jhi, hi, cnt * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipStreamSynchronize(h->stream));
  h->jl_loaded = true; h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_load_quadratic(sco_sqp *h, const double *Q, const double *a, const double *c) {
  if (!h || !Q || !a || !c) { sco_set_error("sco_sqp_load_quadratic: null pointer"); return SCO_ERR_ARG; }
  if ((h->desc.family & 15) != SCO_FAM_STATE_QUADRATIC) { sco_set_error("sco_sqp_load_quadratic: family has no quadratic rows"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_quadratic: call sco_sqp_load first"); return SCO_ERR_STATE; }
  const size_t B = h->d.batch, O = h->d.O, d = h->d.d;
  for (size_t k = 0; k < B * O; k++)
    for (size_t i = 0; i < d; i++)
      for (size_t j = 0; j < i; j++)
        if (Q[k * d * d + i * d + j] != Q[k * d * d + j * d + i]) { sco_set_error("sco_sqp_load_quadratic: Q must be symmetric"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpyAsync(h->d.qQ, Q, B * O * d * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(h->d.qa, a, B * O * d * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipMemcpyAsync(h->d.qc, c, B * O * sizeof(double), hipMemcpyHostToDevice, h->stream));
  SCO_HIP(hipStreamSynchronize(h->stream));
  h->quad_loaded = true; h->solved = false;
  return SCO_OK;
}

static int load_program_impl(sco_sqp *h, int n_words, const int *words, const int *row_ptr, int n_consts, const double *consts,
                             int n_params, const double *params, bool per_step) {
  if (!h || !words || !row_ptr || (n_consts > 0 && !consts) || (n_params > 0 && !params)) {
    sco_set_error("sco_sqp_load_program: null pointer"); return SCO_ERR_ARG;
  }
  if ((h->desc.family & 15) != SCO_FAM_STATE_PROGRAM) { sco_set_error("sco_sqp_load_program: family has no row programs"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_program: call sco_sqp_load first"); return SCO_ERR_STATE; }
  // the rows of a block, then (SCO_FAM_FLAG_OBJ_PROGRAM) the objective term of a timestep
  const int R = h->d.O - h->d.R1 + (h->d.cost == 2 ? 1 : 0), ds = h->d.ds;       // (r04: R1 leading rows of a block are circle rows)
  if (n_words <= 0 || n_consts < 0 || n_params < 0) { sco_set_error("sco_sqp_load_program: bad program layout"); return SCO_ERR_ARG; }
  // the whole of row_ptr is checked BEFORE any word is read through it: 0 = first entry, strictly increasing, last = n_words
  if (row_ptr[0] != 0) { sco_set_error("sco_sqp_load_program: bad program layout"); return SCO_ERR_ARG; }
  for (int r = 0; r < R; r++)
    if (row_ptr[r + 1] <= row_ptr[r] || row_ptr[r + 1] > n_words) { sco_set_error("sco_sqp_load_program: bad program layout"); return SCO_ERR_ARG; }
  if (row_ptr[R] != n_words) { sco_set_error("sco_sqp_load_program: bad program layout"); return SCO_ERR_ARG; }
  // every row's program is run on the host once, symbolically: stack depth and operand indices
  for (int r = 0; r < R; r++) {
    if (words[2 * (row_ptr[r + 1] - 1)] != SCO_OP_END) {
      sco_set_error("sco_sqp_load_program: a row's program must end with SCO_OP_END"); return SCO_ERR_ARG;
    }
    const int nx = r < h->d.O - h->d.R1 ? ds : h->d.d;      // the objective term sees one timestep
    int sp = 0;
    for (int w = row_ptr[r]; w < row_ptr[r + 1] - 1; w++) {
      const int op = words[2 * w], arg = words[2 * w + 1];
      bool ok = true;
      if (op == SCO_OP_X) { ok = arg >= 0 && arg < nx; sp++; }
      else if (op == SCO_OP_P) { ok = arg >= 0 && arg < n_params; sp++; }
      else if (op == SCO_OP_C) { ok = arg >= 0 && arg < n_consts; sp++; }
      else if (op >= SCO_OP_ADD && op <= SCO_OP_DIV) { ok = sp >= 2; sp--; }
      else if (op >= SCO_OP_NEG && op <= SCO_OP_SQUARE) ok = sp >= 1;
      else ok = false;
      if (!ok || sp > SCO_PROGRAM_STACK) { sco_set_error("sco_sqp_load_program: malformed program (operand index, stack depth or opcode)"); return SCO_ERR_ARG; }
    }
    if (sp != 1) { sco_set_error("sco_sqp_load_program: a row's program must leave exactly one value"); return SCO_ERR_ARG; }
  }
  SCO_ON_DEVICE(h->device);
  SqpDev &s = h->d;
  // the four buffers belong to the handle: a reload of the same sizes (new parameters per solve) reuses them, another
  // size frees the old ones first
  const size_t need[4] = {(size_t)2 * n_words * sizeof(int), ((size_t)R + 1) * sizeof(int), (size_t)std::max(n_consts, 1) * sizeof(double),
                          std::max<size_t>((size_t)s.batch * (per_step ? s.T : 1) * n_params, 1) * sizeof(double)};
  // From here until every buffer is in place and filled the handle has NO program: a failed hipMalloc / hipMemcpy below
  // returns early, and the kernels of a later sco_sqp_solve must not read freed or half-written buffers (solve refuses
  // while prog_loaded is false).
  h->prog_loaded = false; h->solved = false;
  s.pw = nullptr; s.pptr = nullptr; s.pconst = nullptr; s.ppar = nullptr; s.n_par = 0; s.par_step = 0;
  for (int k = 0; k < 4; k++)
    if (h->prog_bytes[k] != need[k]) {
      if (h->prog_buf[k]) { (void)hipFree(h->prog_buf[k]); h->prog_buf[k] = nullptr; h->prog_bytes[k] = 0; }
      SCO_HIP(hipMalloc(&h->prog_buf[k], need[k]));
      h->prog_bytes[k] = need[k];
    }
  SCO_HIP(hipMemcpy(h->prog_buf[0], words, (size_t)2 * n_words * sizeof(int), hipMemcpyHostToDevice));
  SCO_HIP(hipMemcpy(h->prog_buf[1], row_ptr, ((size_t)R + 1) * sizeof(int), hipMemcpyHostToDevice));
  if (n_consts) SCO_HIP(hipMemcpy(h->prog_buf[2], consts, (size_t)n_consts * sizeof(double), hipMemcpyHostToDevice));
  if (n_params) SCO_HIP(hipMemcpy(h->prog_buf[3], params, (size_t)s.batch * (per_step ? s.T : 1) * n_params * sizeof(double), hipMemcpyHostToDevice));
  s.pw = (const int *)h->prog_buf[0]; s.pptr = (const int *)h->prog_buf[1];
  s.pconst = (const double *)h->prog_buf[2]; s.ppar = (const double *)h->prog_buf[3]; s.n_par = n_params;
  s.par_step = per_step ? n_params : 0;
  h->prog_loaded = true; h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_load_program(sco_sqp *h, int n_words, const int *words, const int *row_ptr, int n_consts, const double *consts,
                                    int n_params, const double *params) {
  return load_program_impl(h, n_words, words, row_ptr, n_consts, consts, n_params, params, false);
}
// r04: parameters per problem AND timestep, params[batch][horizon][n_params]: block t (the Variable of timesteps t .. t + span - 1)
// and the objective term of timestep t are evaluated with params[problem][t] -- what a caller of the reference gets by closing
// each timestep's Expr over its own data (moving obstacles, time-varying references; expr.py:22-41, prob.py:112-144)
extern "C" int sco_sqp_load_program_steps(sco_sqp *h, int n_words, const int *words, const int *row_ptr, int n_consts,
                                          const double *consts, int n_params, const double *params) {
  return load_program_impl(h, n_words, words, row_ptr, n_consts, consts, n_params, params, true);
}

// r04 (SCO_FAM_FLAG_ACC_COST): weights a[batch][dof] >= 0 of the acceleration term of the quadratic objective,
//     sum_t sum_j a_j (theta[t+2][j] - 2 theta[t+1][j] + theta[t][j])^2,
// next to the (weighted) velocity term -- the QuadExpr a caller builds from first AND second difference matrices (prob.py:88-104,
// 348-367).  P gets its second super-diagonal block (pattern fixed at sco_sqp_create by the flag).  After sco_sqp_load; nullptr = 0.
extern "C" int sco_sqp_load_acc_weights(sco_sqp *h, const double *a) {
  if (!h) { sco_set_error("sco_sqp_load_acc_weights: null pointer"); return SCO_ERR_ARG; }
  if (!h->d.acc) { sco_set_error("sco_sqp_load_acc_weights: the handle was created without SCO_FAM_FLAG_ACC_COST"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_acc_weights: call sco_sqp_load first"); return SCO_ERR_STATE; }
  SqpDev &s = h->d;
  if (!a) { s.accw = nullptr; h->solved = false; return SCO_OK; }
  for (size_t i = 0; i < (size_t)s.batch * s.d; i++)
    if (!(a[i] >= 0.0) || !(a[i] < 1e30)) { sco_set_error("sco_sqp_load_acc_weights: weights must be finite and >= 0"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  s.accw = nullptr;
  if (!h->accw_buf) SCO_HIP(hipMalloc(&h->accw_buf, (size_t)s.batch * s.d * sizeof(double)));
  SCO_HIP(hipMemcpy(h->accw_buf, a, (size_t)s.batch * s.d * sizeof(double), hipMemcpyHostToDevice));
  s.accw = (const double *)h->accw_buf;
  h->solved = false;
  return SCO_OK;
}

// r04: two kinds of non-linear rows in one problem.  SCO_FAM_STATE_PROGRAM, span 1, before sco_sqp_load_program: the first n_rows
// rows of every block are keep-out rows r_o - || x[0:2] - c_o || of the point (x[0], x[1]) against obstacles[b][0 .. n_rows) of
// sco_sqp_load (SCO_FAM_POINT_CIRCLES' rows); the program supplies the remaining n_obstacles - n_rows rows.  In the reference: two
// BoundExprs on the same timestep Variable, the circle Expr added first (prob.py:112-144).  0 restores the plain program family.
extern "C" int sco_sqp_set_circle_rows(sco_sqp *h, int n_rows) {
  if (!h) { sco_set_error("sco_sqp_set_circle_rows: null pointer"); return SCO_ERR_ARG; }
  if ((h->desc.family & 15) != SCO_FAM_STATE_PROGRAM || h->d.S != 1 || h->d.d < 2) {
    sco_set_error("sco_sqp_set_circle_rows: program family on single timesteps with dof >= 2 only"); return SCO_ERR_ARG;
  }
  if (n_rows < 0 || n_rows >= h->d.O || n_rows > h->d.O - h->d.Req - 1 + 1) { sco_set_error("sco_sqp_set_circle_rows: 0 <= n_rows < rows per block, equality rows belong to the program"); return SCO_ERR_ARG; }
  if (n_rows != h->d.R1) { h->prog_loaded = false; h->solved = false; }       // the program's row count changes with it
  h->d.R1 = n_rows;
  return SCO_OK;
}

// r04: coefficients vals[batch][nnz] (CSR order of the pattern given to sco_sqp_create_rows) and right-hand sides
// rhs[batch][n_rows] of the general affine rows; after sco_sqp_load, before sco_sqp_solve; may be called again.
extern "C" int sco_sqp_load_linear_rows(sco_sqp *h, const double *vals, const double *rhs) {
  if (!h || !vals || !rhs) { sco_set_error("sco_sqp_load_linear_rows: null pointer"); return SCO_ERR_ARG; }
  SqpDev &s = h->d;
  if (!s.m_gen) { sco_set_error("sco_sqp_load_linear_rows: the handle was created without general affine rows"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_linear_rows: call sco_sqp_load first"); return SCO_ERR_STATE; }
  for (size_t i = 0; i < (size_t)s.batch * s.nnz_gen; i++)
    if (!(fabs(vals[i]) < 1e30)) { sco_set_error("sco_sqp_load_linear_rows: coefficients must be finite"); return SCO_ERR_ARG; }
  for (size_t i = 0; i < (size_t)s.batch * s.m_gen; i++)
    if (!(fabs(rhs[i]) < 1e30)) { sco_set_error("sco_sqp_load_linear_rows: right-hand sides must be finite"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpy((void *)s.gen_val, vals, (size_t)s.batch * s.nnz_gen * sizeof(double), hipMemcpyHostToDevice));
  SCO_HIP(hipMemcpy((void *)s.gen_rhs, rhs, (size_t)s.batch * s.m_gen * sizeof(double), hipMemcpyHostToDevice));
  h->gen_loaded = true; h->solved = false;
  return SCO_OK;
}

// r04: weights of the smoothing objective, w[batch][dof] >= 0: sum_t sum_j w_j (theta[t+1][j] - theta[t][j])^2 -- the QuadExpr a
// caller of the reference builds with a weighted difference matrix (prob.py:88-104, 348-367).  After sco_sqp_load; nullptr
// restores the unweighted objective.
extern "C" int sco_sqp_load_obj_weights(sco_sqp *h, const double *w) {
  if (!h) { sco_set_error("sco_sqp_load_obj_weights: null pointer"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_load_obj_weights: call sco_sqp_load first"); return SCO_ERR_STATE; }
  SqpDev &s = h->d;
  if (!w) { s.objw = nullptr; h->solved = false; return SCO_OK; }
  for (size_t i = 0; i < (size_t)s.batch * s.d; i++)
    if (!(w[i] >= 0.0) || !(w[i] < 1e30)) { sco_set_error("sco_sqp_load_obj_weights: weights must be finite and >= 0"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(h->device);
  s.objw = nullptr;
  if (!h->objw_buf) SCO_HIP(hipMalloc(&h->objw_buf, (size_t)s.batch * s.d * sizeof(double)));
  SCO_HIP(hipMemcpy(h->objw_buf, w, (size_t)s.batch * s.d * sizeof(double), hipMemcpyHostToDevice));
  s.objw = (const double *)h->objw_buf;
  h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_set_groups(sco_sqp *h, int n_groups, const unsigned int *block_mask) {
  if (!h || !block_mask) { sco_set_error("sco_sqp_set_groups: null pointer"); return SCO_ERR_ARG; }
  if (n_groups < 1 || n_groups > 32) { sco_set_error("sco_sqp_set_groups: 1..32 groups"); return SCO_ERR_ARG; }
  SqpDev &s = h->d;
  const unsigned int all = n_groups == 32 ? 0xffffffffu : ((1u << n_groups) - 1u);
  unsigned int overlap[32] = {0};
  for (int blk = 0; blk < s.NB; blk++) {
    const unsigned int mk = block_mask[blk];
    if (mk == 0u || (mk & ~all)) { sco_set_error("sco_sqp_set_groups: every block needs a group < n_groups"); return SCO_ERR_ARG; }
    for (int g = 0; g < n_groups; g++) if ((mk >> g) & 1u) overlap[g] |= mk & ~(1u << g);    // prob.py:139-142
  }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpy((void *)s.gmask, block_mask, (size_t)s.NB * sizeof(unsigned int), hipMemcpyHostToDevice));
  SCO_HIP(hipMemcpy((void *)s.goverlap, overlap, sizeof overlap, hipMemcpyHostToDevice));
  s.G = n_groups;
  h->solved = false;
  return SCO_OK;
}

extern "C" int sco_sqp_fetch_flags(sco_sqp *h, int *flags) {
  if (!h || !flags) return SCO_ERR_ARG;
  if (!h->solved) { sco_set_error("sco_sqp_fetch_flags: call sco_sqp_solve first"); return SCO_ERR_STATE; }
  SCO_ON_DEVICE(h->device);
  const size_t B = h->d.batch;
  std::vector<SqpScalars> sc(B);
  SCO_HIP(hipMemcpy(sc.data(), h->d.sc, B * sizeof(SqpScalars), hipMemcpyDeviceToHost));
  for (size_t b = 0; b < B; b++) flags[b] = sc[b].flags;
  return SCO_OK;
}

extern "C" int sco_sqp_fetch_groups(sco_sqp *h, unsigned int *nonconverged) {
  if (!h || !nonconverged) return SCO_ERR_ARG;
  if (!h->solved) { sco_set_error("sco_sqp_fetch_groups: call sco_sqp_solve first"); return SCO_ERR_STATE; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpy(nonconverged, h->d.nonconv, (size_t)h->d.batch * sizeof(unsigned int), hipMemcpyDeviceToHost));
  return SCO_OK;
}

extern "C" int sco_sqp_fetch_stalled_groups(sco_sqp *h, unsigned int *stalled) {
  if (!h || !stalled) return SCO_ERR_ARG;
  if (!h->solved) { sco_set_error("sco_sqp_fetch_stalled_groups: call sco_sqp_solve first"); return SCO_ERR_STATE; }
  SCO_ON_DEVICE(h->device);
  SCO_HIP(hipMemcpy(stalled, h->d.stalled, (size_t)h->d.batch * sizeof(unsigned int), hipMemcpyDeviceToHost));
  return SCO_OK;
}

static hipEvent_t next_event(sco_sqp *h, size_t &cursor) {
  if (cursor == h->events.size()) {
    hipEvent_t e; (void)hipEventCreate(&e); h->events.push_back(e);
  }
  return h->events[cursor++];
}

// Which problems take part in the next round.  A workgroup of the ADMM kernels fills a CU and the hardware deals the
// workgroups of a launch to XCDs and shader engines in a fixed rotation, waiting for the engine whose turn it is: a
// workgroup that finds its problem inactive and exits at once still takes its turn, so a launch of 1024 workgroups of
// which 768 have work costs four passes of the chip, not three (measured, profiles/r02_rounds.txt), and
// once problems start to finish the last pass of every lock-step round is partly empty.  With more active problems
// than CUs a round therefore runs P = floor(active / CUs) whole passes as a COMPACT launch: this kernel lists the
// P x CUs problems with most in front of them as far as the device can tell -- the slices their current QP has left if
// it runs to max_iter, one more QP for a problem still in its first penalty QP (a step is normally followed by at
// least one more), ties by index -- and workgroup g of the round's kernels takes problem list[g].  The host sizes the
// launch from the active count it read back SQP_DEPTH rounds earlier (`cap`, never too small: the count only falls);
// surplus workgroups find -1 and exit.  The others are not visited: per problem the sequence of kernels and every
// result is unchanged, only the round it happens in moves (model on the measured chains, 1024 problems at 7x20:
// 667 -> 636 ms per step; choosing by the true remaining work would give 629, profiles/r02_slice_model.txt).
// n_active starts the round at the number of live problems left out (sqp_post_kernel adds those that ran and go on).
#define SEL_T 1024
#define SEL_BUCKETS 1024
// (r03: the kernel works on the problems [s.b0, s.b0 + nb) of a stream group and fills that group's part of the list, so
// selection and stream groups combine: SCO_SQP_GROUPS)
__global__ __launch_bounds__(SEL_T) void sqp_select_kernel(SqpDev s, QpDev q1, int cus, int slice, int max_iter, int cap, int nb) {
  __shared__ int hist[SEL_BUCKETS];
  __shared__ int part[SEL_T], part2[SEL_T];
  __shared__ int s_active, s_thr, s_take;
  const int tid = threadIdx.x, B = nb, gb0 = s.b0;
  int *list = s.list_buf + gb0;
  for (int k = tid; k < SEL_BUCKETS; k += SEL_T) hist[k] = 0;
  if (tid == 0) s_active = 0;
  __syncthreads();
  const int per_qp = slice > 0 ? (max_iter + slice - 1) / slice : 1;
  // bucket 0 = most in front of it; -1 = finished
  auto key_of = [&](int b) -> int {
    const SqpScalars &sc = s.sc[b];
    if (sc.state == ST_DONE) return -1;
    const int done = (q1.prog && slice > 0) ? q1.prog[b] / slice : 0;
    const int est = max(per_qp - done, 0) + (sc.qp_solves <= 1 ? per_qp : 0);
    return max(SEL_BUCKETS - 1 - est, 0);
  };
  const int per = (B + SEL_T - 1) / SEL_T, b0 = gb0 + min(B, tid * per), b1 = min(gb0 + B, b0 + per);
  int mine = 0;
  for (int b = b0; b < b1; b++) {
    const int k = key_of(b);
    if (k >= 0) { atomicAdd(&hist[k], 1); mine++; }
  }
  if (mine) atomicAdd(&s_active, mine);
  __syncthreads();
  const int A = s_active;
  const int quota = min(cap, (cus <= 0 || A <= cus) ? A : (A / cus) * cus);
  if (tid == 0) {
    int acc = 0, k = 0;
    for (; k < SEL_BUCKETS; k++) { if (acc + hist[k] > quota) break; acc += hist[k]; }
    s_thr = k; s_take = quota - acc;           // every bucket below s_thr runs; of bucket s_thr the first s_take by index
    *s.n_active = A - quota;
  }
  __syncthreads();
  const int thr = s_thr, take = s_take;
  // two exclusive scans over the threads' consecutive chunks: members of bucket thr (ties), then chosen problems
  int cnt = 0;
  for (int b = b0; b < b1; b++) cnt += key_of(b) == thr ? 1 : 0;
  part[tid] = cnt;
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int t = 0; t < SEL_T; t++) { const int v = part[t]; part[t] = acc; acc += v; } }
  __syncthreads();
  int seen = part[tid], chosen = 0;
  for (int b = b0; b < b1; b++) {
    const int k = key_of(b);
    if (k >= 0 && k < thr) chosen++;
    else if (k == thr) { if (seen < take) chosen++; seen++; }
  }
  part2[tid] = chosen;
  __syncthreads();
  if (tid == 0) { int acc = 0; for (int t = 0; t < SEL_T; t++) { const int v = part2[t]; part2[t] = acc; acc += v; } }
  __syncthreads();
  int pos = part2[tid];
  seen = part[tid];
  for (int b = b0; b < b1; b++) {
    const int k = key_of(b);
    bool r = false;
    if (k >= 0 && k < thr) r = true;
    else if (k == thr) { r = seen < take; seen++; }
    if (r) list[pos++] = b;
  }
  for (int g = quota + tid; g < cap; g += SEL_T) list[g] = -1;
}

// problems the host loop's launch cap left unfinished: failed + SCO_SQP_FLAG_CAPPED
__global__ void sqp_cap_kernel(SqpDev s) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= s.batch) return;
  SqpScalars &sc = s.sc[b];
  if (sc.state != ST_DONE) { sc.state = ST_DONE; sc.success = 0; sc.flags |= SCO_SQP_FLAG_CAPPED; s.active[b] = 0; }
}

extern "C" int sco_sqp_solve(sco_sqp *h, const sco_sqp_params *params, const sco_qp_settings *qs) {
  if (!h || !params || !qs) { sco_set_error("sco_sqp_solve: null pointer"); return SCO_ERR_ARG; }
  if (!h->loaded) { sco_set_error("sco_sqp_solve: call sco_sqp_load first"); return SCO_ERR_STATE; }
  if ((h->desc.family & 15) == SCO_FAM_ARM_REACH && !h->target_loaded) {
    sco_set_error("sco_sqp_solve: call sco_sqp_load_target first"); return SCO_ERR_STATE;
  }
  if ((h->desc.family & SCO_FAM_FLAG_VEL_LIMITS) && !h->vel_loaded) {
    sco_set_error("sco_sqp_solve: call sco_sqp_load_vel_limit first"); return SCO_ERR_STATE;
  }
  if ((h->desc.family & SCO_FAM_FLAG_EE_COST) && !h->cost_loaded) {
    sco_set_error("sco_sqp_solve: call sco_sqp_load_ee_cost first"); return SCO_ERR_STATE;
  }
  if ((h->desc.family & SCO_FAM_FLAG_JOINT_LIMITS) && !h->jl_loaded) {
    sco_set_error("sco_sqp_solve: call sco_sqp_load_joint_limits first"); return SCO_ERR_STATE;
  }
  if (h->d.m_gen && !h->gen_loaded) {
    sco_set_error("sco_sqp_solve: call sco_sqp_load_linear_rows first"); return SCO_ERR_STATE;
  }
  if ((h->desc.family & 15) == SCO_FAM_STATE_PROGRAM && !h->prog_loaded) {
    sco_set_error("sco_sqp_solve: call sco_sqp_load_program first"); return SCO_ERR_STATE;
  }
  if ((h->desc.family & 15) == SCO_FAM_STATE_QUADRATIC && !h->quad_loaded) {
    sco_set_error("sco_sqp_solve: call sco_sqp_load_quadratic first"); return SCO_ERR_STATE;
  }
  h->solved = false;            // a failed call must not leave an older result readable through fetch / trace
  // settings are checked before the first launch: nothing on the device is touched by a call that is going to fail
  if (qs->max_iter <= 0 || !(qs->rho > 0) || !(qs->sigma > 0) || qs->scaling < 0 || !(qs->alpha > 0) || !(qs->alpha < 2)) {
    sco_set_error("sco_sqp_solve: bad QP settings (max_iter, rho, sigma > 0; 0 < alpha < 2; scaling >= 0)"); return SCO_ERR_ARG;
  }
  if (!(params->initial_trust_region_size > 0) || !(params->initial_penalty_coeff > 0) ||
      !(params->trust_shrink_ratio > 0) || !(params->trust_expand_ratio > 0) || !(params->merit_coeff_increase_ratio > 0) ||
      params->max_merit_coeff_increases < 0) {
    sco_set_error("sco_sqp_solve: bad SQP parameters"); return SCO_ERR_ARG;
  }
  if (qs->adaptive_rho) {
    if (!(qs->adaptive_rho_tolerance > 1.0) || qs->check_termination <= 0) {
      sco_set_error("sco_sqp_solve: adaptive_rho needs adaptive_rho_tolerance > 1 and check_termination > 0"); return SCO_ERR_ARG;
    }
    if (!sco_qp_can_adapt(h->qp1)) {
      sco_set_error("sco_sqp_solve: adaptive_rho: the dense form of the global-memory tier cannot park a solve");
      return SCO_ERR_CAPACITY;
    }
  }
  SCO_ON_DEVICE(h->device);
  SqpDev &s = h->d;
  SqpParamsDev p{params->improve_ratio_threshold, params->min_trust_region_size, params->min_approx_improve,
                 params->trust_shrink_ratio, params->trust_expand_ratio, params->cnt_tolerance,
                 params->merit_coeff_increase_ratio, params->initial_trust_region_size, params->initial_penalty_coeff,
                 params->max_merit_coeff_increases, params->compound_penalty, params->duplicate_rows,
                 params->max_sqp_iters > 0 ? params->max_sqp_iters : 10000, params->memoize_rounded ? 1 : 0};
  const dim3 grid(s.batch), block(SCO_BLOCK);
  size_t ec = 0;
  std::vector<int> stage;   // stage id of the interval that ENDS at event i
  auto mark = [&](int st) { hipEvent_t e = next_event(h, ec); (void)hipEventRecord(e, h->stream); stage.push_back(st); };
  mark(-1);
  // ---- round 0: projection onto the linear constraints, DEFAULT QP settings (Q7)
  SCO_HIP(hipMemsetAsync(s.n_active, 0, sizeof(int), h->stream));
  hipLaunchKernelGGL(sqp_proj_assemble_kernel, grid, block, 0, h->stream, s, h->qp0->d);
  SCO_HIP(hipGetLastError());
  mark(0);
  sco_qp_settings q0s; sco_qp_default_settings(&q0s);
  hipEvent_t mid = next_event(h, ec); stage.push_back(1);
  int rc = sco_qp_launch(h->qp0, &q0s, nullptr, mid);
  if (rc) return rc;
  mark(2);
  hipLaunchKernelGGL(sqp_proj_post_kernel, grid, block, 0, h->stream, s, h->qp0->d, h->qp1->d, p);
  SCO_HIP(hipGetLastError());
  mark(3);
  int n_active = 0;
  SCO_HIP(hipMemcpyAsync(&n_active, s.n_active, sizeof(int), hipMemcpyDeviceToHost, h->stream));
  SCO_HIP(hipStreamSynchronize(h->stream));
  h->rounds = 1;
  // ---- rounds: one QP solve per active problem
  sco_qp_settings qsl = *qs;
  if (params->warm_start_qps) {
    // beyond parity: every penalty QP starts from the previous one's solution; the first one from zero
    qsl.warm_start = 1;
    const QpDev &q1 = h->qp1->d;
    SCO_HIP(hipMemsetAsync(q1.x, 0, (size_t)q1.batch * q1.n * sizeof(double), h->stream));
    SCO_HIP(hipMemsetAsync(q1.y, 0, (size_t)q1.batch * q1.m * sizeof(double), h->stream));
    h->qp1->solved_once = true;
  }
  // time slicing (scheduling only): every launch advances each active QP by at most `slice` ADMM iterations; a
  // problem whose QP ended goes through post / pre / setup and joins the next launch with its next QP
  // default slice: 6250 iterations (7-DOF x 20: 964 ms per 1024-batch step against 1106 unsliced); with adaptive rho
  // the QPs are short and every rho change costs its problem a relaunch, so the slice is shorter (scripts/gpu_adaptive_slice_sweep.py)
  int slice_req = params->admm_slice < 0 ? 0 : (params->admm_slice > 0 ? params->admm_slice : (qs->adaptive_rho ? 2000 : 6250));
  if (params->admm_slice == 0) { const char *se = getenv("SCO_SQP_SLICE"); if (se && atoi(se) > 0) slice_req = atoi(se); }   // tuning aid: the default slice
  if (params->admm_slice == 0) {
    // with at most one problem per CU there is nobody to hand a CU to: slicing would only add relaunches
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && s.batch <= cus) slice_req = 0;
  }
  SCO_HIP(hipMemsetAsync(h->qp1->d.prog, 0, (size_t)s.batch * sizeof(int), h->stream));
  sco_qp_wv_iters_reset(h->qp1, h->stream);
  long long slices_per_qp = slice_req > 0 ? (qsl.max_iter + slice_req - 1) / slice_req : 1;
  if (qsl.adaptive_rho) slices_per_qp += qsl.max_iter / sco_qp_adaptive_interval(&qsl) + 1;    // a launch per rho change at most
  // (a round with selection runs at least half of the active problems, hence the factor 2)
  const long long round_cap = 2 * ((long long)p.max_qp_solves + 8) * slices_per_qp;
  // ---- scheduling of the rounds (results never depend on it).  A lock-step round costs ceil(active / CUs) passes of
  // workgroups, so once problems start to finish the last pass of every round is partly empty
  // (profiles/r02_launches.txt: 737 ms per step against 630 ms of work).  Default: ROUND SELECTION -- with more active
  // problems than CUs a round runs a whole number of passes, the problems with most in front of them first
  // (sqp_select_kernel), and SQP_DEPTH rounds are enqueued ahead so the device never waits for the host.
  // Opt-in on top (SCO_SQP_GROUPS = 2..4): the batch is cut into contiguous STREAM GROUPS that run their rounds
  // independently on streams of their own -- each with its own selection since r03 -- so that one group's launch fills the
  // CUs the end of another's leaves free.  (Measured on the 1024-problem 7x20 step: DESIGN.md 3.3.)
  int G = 1, cus = 0;
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device);
  {
    const char *ge = getenv("SCO_SQP_GROUPS");
    const int want = ge ? atoi(ge) : 1;
    if (slice_req > 0 && sco_qp_supports_groups(h->qp1, &qsl) && cus > 0 && s.batch >= 2 * cus)
      G = std::max(1, std::min(std::min(want, SQP_MAX_GROUPS), s.batch / cus));
  }
  const char *sel_env = getenv("SCO_SQP_SELECT");
  const bool select = slice_req > 0 && cus > 0 && s.batch > cus && sco_qp_supports_groups(h->qp1, &qsl) &&
                      !(sel_env && sel_env[0] == '0');
  if (getenv("SCO_SQP_TRACE_ROUNDS")) fprintf(stderr, "sco_sqp_solve: %d CUs, %d stream group(s), round selection %s, slice %d\n", cus, G, select ? "on" : "off", slice_req);
  // Tier of a round (handles whose penalty QP has the wavefront tier, parity-mode settings): with at least wv_min live
  // problems the round runs on the wavefront tier -- every live problem at once, four per CU -- below that on the row-local
  // kernel, one problem per CU in whole passes.  The first is the higher THROUGHPUT while the batch is alive (1024 / 3.1 us
  // against 256 / 0.95 us per iteration), the second the lower LATENCY for the tail of a step; a problem's QP changes kernel
  // at a slice boundary (the parked state is common).  Both kernels agree to rounding (1e-14), not bit for bit: for batches
  // that ever have wv_min live problems the last bits of a result depend on the schedule (SCO_WV_MIN_PER_CU=1e9: never).
  const bool has_wv = select && sco_qp_has_wv(h->qp1, &qsl);
  const int wv_min = sco_wv_min_live(cus);
  h->groups_used = G;
  for (int g = 1; g < G; g++)
    if (!h->gstream[g - 1]) SCO_HIP(hipStreamCreate(&h->gstream[g - 1]));
  struct Group {
    int b0 = 0, nb = 0; hipStream_t st = nullptr;
    int issued = 0, retired = 0, last_active = 1;      // rounds enqueued / read back; active count of the last one read
    size_t ec = 0; std::vector<int> stage;              // event cursor and stage ids (as for the main stream)
    double ms[5] = {0, 0, 0, 0, 0};
  } grp[SQP_MAX_GROUPS];
  for (int g = 0; g < G; g++) {
    grp[g].b0 = (int)((long long)s.batch * g / G); grp[g].nb = (int)((long long)s.batch * (g + 1) / G) - grp[g].b0;
    grp[g].st = g == 0 ? h->stream : h->gstream[g - 1];
    grp[g].last_active = G == 1 ? n_active : grp[g].nb;       // upper bound of the group's live problems
  }
  auto gevent = [&](int g) -> hipEvent_t {
    Group &r = grp[g];
    if (r.ec == h->gevents[g].size()) { hipEvent_t e; (void)hipEventCreate(&e); h->gevents[g].push_back(e); }
    return h->gevents[g][r.ec++];
  };
  auto gmark = [&](int g, int st) { hipEvent_t e = gevent(g); (void)hipEventRecord(e, grp[g].st); grp[g].stage.push_back(st); };
  std::vector<hipEvent_t> &done = h->done;          // per (group, slot): the round's read-back has landed
  while (done.size() < (size_t)G * SQP_DEPTH) { hipEvent_t e; SCO_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); done.push_back(e); }
  int wv_rounds = 0;
  auto enqueue_round = [&](int g) -> int {
    Group &r = grp[g];
    SqpDev sg = s; sg.b0 = r.b0; sg.n_active = s.n_active + g;
    if (r.issued == 0) gmark(g, -1);
    if (!select) SCO_HIP(hipMemsetAsync(sg.n_active, 0, sizeof(int), r.st));
    sg.list = nullptr;
    int nwg = r.nb;                              // workgroups of this round's kernels
    const bool wv_round = has_wv && r.last_active >= wv_min;
    if (select) {
      // compact launch: as many workgroups as the selection can let run, sized from the newest active count the host has
      // (a pass of the chip = one problem per CU, four on the wavefront tier)
      const int pass = wv_round ? 4 * cus : cus;
      nwg = std::max(1, r.last_active <= pass ? r.last_active : (r.last_active / pass) * pass);
      hipLaunchKernelGGL(sqp_select_kernel, dim3(1), dim3(SEL_T), 0, r.st, sg, h->qp1->d, pass, slice_req, qsl.max_iter, nwg, r.nb);
      SCO_HIP(hipGetLastError());
      sg.list = s.list_buf;
    }
    hipLaunchKernelGGL(sqp_pre_kernel, dim3(nwg), block, 0, r.st, sg, h->qp1->d, p);
    SCO_HIP(hipGetLastError());
    gmark(g, 0);
    hipEvent_t gm = gevent(g); r.stage.push_back(1);
    const QpGroup win{r.b0, nwg, r.st, sg.list, has_wv ? (wv_round ? 2 : 1) : 0};
    if (wv_round) wv_rounds++;
    const int rc_ = sco_qp_launch_sliced(h->qp1, &qsl, s.newqp, s.active, slice_req, gm, nullptr, (G > 1 || select) ? &win : nullptr);
    if (rc_) return rc_;
    gmark(g, wv_round ? 5 : 2);
    hipLaunchKernelGGL(sqp_post_kernel, dim3(nwg), block, 0, r.st, sg, h->qp1->d, p);
    SCO_HIP(hipGetLastError());
    gmark(g, 3);
    const int slot = r.issued % SQP_DEPTH;
    SCO_HIP(hipMemcpyAsync(h->host_active + g * SQP_DEPTH + slot, sg.n_active, sizeof(int), hipMemcpyDeviceToHost, r.st));
    SCO_HIP(hipEventRecord(done[(size_t)g * SQP_DEPTH + slot], r.st));
    r.issued++;
    return SCO_OK;
  };
  // Rounds kept enqueued ahead of the host: SQP_DEPTH where a round's read-back would otherwise leave the device idle
  // (selection, stream groups, time slices); ONE for the plain unsliced loop over a batch that fits the CUs (the B = 1
  // latency case): there a second round in flight would only be a trailing all-inactive launch sequence per solve.
  const int depth = (!select && G == 1 && slice_req == 0) ? 1 : SQP_DEPTH;
  bool capped = false;
  if (G > 1) {
    // the other groups' streams start behind the memsets queued on the main stream above (prog reset, warm-start zeroing)
    hipEvent_t ready = next_event(h, ec); stage.push_back(-2);
    SCO_HIP(hipEventRecord(ready, h->stream));
    for (int g = 1; g < G; g++) SCO_HIP(hipStreamWaitEvent(grp[g].st, ready, 0));
  }
  if (n_active > 0) {
    for (int k = 0; k < depth; k++)
      for (int g = 0; g < G; g++)
        if ((rc = enqueue_round(g))) return rc;
    for (bool busy = true; busy;) {
      busy = false;
      for (int g = 0; g < G; g++) {
        Group &r = grp[g];
        if (r.retired == r.issued) continue;
        busy = true;
        const int slot = r.retired % SQP_DEPTH;
        SCO_HIP(hipEventSynchronize(done[(size_t)g * SQP_DEPTH + slot]));
        r.last_active = h->host_active[g * SQP_DEPTH + slot];
        r.retired++;
        if (getenv("SCO_SQP_TRACE_ROUNDS") && (r.retired < 40 || r.retired % 100 == 0))
          fprintf(stderr, "sco_sqp_solve: group %d round %d, %d problems active\n", g, r.retired, r.last_active);
        if (r.last_active > 0) {
          if (r.issued + 1 < round_cap) { if ((rc = enqueue_round(g))) return rc; }
          else if (r.retired == r.issued) capped = true;
        }
      }
    }
  }
  {
    int total_rounds = 0;
    for (int g = 0; g < G; g++) total_rounds = std::max(total_rounds, grp[g].retired);
    h->rounds = 1 + total_rounds;
    h->launches = 0;
    for (int g = 0; g < G; g++) h->launches += grp[g].retired;
    h->wv_rounds = wv_rounds;
  }
  if (capped) {
    // the launch cap ended the loop with problems still running (it is sized so that this cannot happen while every
    // problem respects max_sqp_iters): they are reported as capped failures, never silently as finished
    hipLaunchKernelGGL(sqp_cap_kernel, dim3((s.batch + SCO_BLOCK - 1) / SCO_BLOCK), block, 0, h->stream, s);
    SCO_HIP(hipGetLastError());
    SCO_HIP(hipStreamSynchronize(h->stream));
  }
  // ---- timing: sum the event intervals by stage.  The rounds of different stream groups overlap, so their intervals
  // are laid on one time axis (milliseconds after the first event of the call) and every instant is charged to ONE
  // stage: the ADMM launch if any group is inside one, else QP setup, else convexify, else the decisions -- the ADMM
  // figure is then the wall time during which at least one ADMM launch was resident or queued behind another group's,
  // and the stages add up to the wall time of the call.
  double ms[5] = {0, 0, 0, 0, 0}, wv_ms = 0.0;
  for (size_t i = 1; i < ec; i++) {
    float t = 0; (void)hipEventElapsedTime(&t, h->events[i - 1], h->events[i]);
    if (stage[i] >= 0) ms[stage[i]] += t;
    ms[4] += t;
  }
  if (G == 1) {
    Group &r = grp[0];
    for (size_t i = 1; i < r.ec; i++) {
      float t = 0; (void)hipEventElapsedTime(&t, h->gevents[0][i - 1], h->gevents[0][i]);
      if (r.stage[i] == 5) { ms[2] += t; wv_ms += t; }
      else if (r.stage[i] >= 0) ms[r.stage[i]] += t;
      ms[4] += t;
    }
  } else {
    struct Edge { double t; int stage, d; };
    std::vector<Edge> edges;
    for (int g = 0; g < G; g++) {
      Group &r = grp[g];
      double prev = 0.0;
      for (size_t i = 0; i < r.ec; i++) {
        float t = 0; (void)hipEventElapsedTime(&t, h->events[0], h->gevents[g][i]);
        const int stg = r.stage[i] == 5 ? 2 : r.stage[i];
        if (i > 0 && r.stage[i] == 5) wv_ms += t - prev;
        if (i > 0 && stg >= 0 && t > prev) { edges.push_back({prev, stg, 1}); edges.push_back({(double)t, stg, -1}); }
        prev = t;
      }
    }
    std::sort(edges.begin(), edges.end(), [](const Edge &x, const Edge &y) { return x.t < y.t; });
    int open_[4] = {0, 0, 0, 0};
    const int prio[4] = {2, 1, 0, 3};
    for (size_t i = 0; i + 1 <= edges.size(); i++) {
      if (i > 0) {
        const double dt = edges[i].t - edges[i - 1].t;
        for (int k : prio) if (open_[k] > 0) { ms[k] += dt; ms[4] += dt; break; }
      }
      open_[edges[i].stage] += edges[i].d;
    }
  }
  memcpy(h->last_ms, ms, sizeof ms);
  h->wv_ms = wv_ms;
  h->solved = true;
  return SCO_OK;
}

extern "C" int sco_sqp_fetch(sco_sqp *h, double *x, int *success, int *sqp_iters, int *qp_solves,
                             long long *admm_iters, double *merit, double *max_violation) {
  if (!h) return SCO_ERR_ARG;
  if (!h->solved) { sco_set_error("sco_sqp_fetch: call sco_sqp_solve first"); return SCO_ERR_STATE; }
  SCO_ON_DEVICE(h->device);
  const SqpDev &s = h->d; const size_t B = s.batch;
  if (x) SCO_HIP(hipMemcpy(x, s.x, B * s.n_x * sizeof(double), hipMemcpyDeviceToHost));
  std::vector<SqpScalars> sc(B);
  SCO_HIP(hipMemcpy(sc.data(), s.sc, B * sizeof(SqpScalars), hipMemcpyDeviceToHost));
  for (size_t b = 0; b < B; b++) {
    if (success) success[b] = sc[b].success;
    if (sqp_iters) sqp_iters[b] = sc[b].sqp_iters;
    if (qp_solves) qp_solves[b] = sc[b].qp_solves;
    if (admm_iters) admm_iters[b] = sc[b].admm_iters;
  }
  if (merit || max_violation) {
    // result buffers of the handle (an allocation per call cost a device-wide synchronisation and now and then 70 ms)
    double *dm = h->fetch_buf, *dv = h->fetch_buf + B;
    hipLaunchKernelGGL(sqp_final_kernel, dim3(s.batch), dim3(SCO_BLOCK), 0, h->stream, s, dm, dv);
    SCO_HIP(hipGetLastError());
    SCO_HIP(hipStreamSynchronize(h->stream));
    if (merit) SCO_HIP(hipMemcpy(merit, dm, B * sizeof(double), hipMemcpyDeviceToHost));
    if (max_violation) SCO_HIP(hipMemcpy(max_violation, dv, B * sizeof(double), hipMemcpyDeviceToHost));
  }
  return SCO_OK;
}

extern "C" int sco_sqp_trace(sco_sqp *h, int cap, double *trace, int *n_entries) {
  if (!h || !trace || !n_entries || cap <= 0) return SCO_ERR_ARG;
  if (!h->solved) { sco_set_error("sco_sqp_trace: call sco_sqp_solve first"); return SCO_ERR_STATE; }
  SCO_ON_DEVICE(h->device);
  const SqpDev &s = h->d; const size_t B = s.batch;
  std::vector<double> tr(B * s.trace_cap * TRACE_W);
  std::vector<SqpScalars> sc(B);
  SCO_HIP(hipMemcpy(tr.data(), s.trace, tr.size() * sizeof(double), hipMemcpyDeviceToHost));
  SCO_HIP(hipMemcpy(sc.data(), s.sc, B * sizeof(SqpScalars), hipMemcpyDeviceToHost));
  for (size_t b = 0; b < B; b++) {
    n_entries[b] = sc[b].n_trace;
    const int rows = std::min(std::min(sc[b].n_trace, s.trace_cap), cap);
    for (int r = 0; r < rows; r++)
      memcpy(trace + (b * cap + r) * TRACE_W, tr.data() + (b * s.trace_cap + r) * TRACE_W, TRACE_W * sizeof(double));
  }
  return SCO_OK;
}

extern "C" int sco_sqp_last_rounds(const sco_sqp *h, int *rounds) {
  if (!h || !rounds) return SCO_ERR_ARG;
  *rounds = h->rounds;
  return SCO_OK;
}

extern "C" int sco_debug_sqp_wv_rounds(const sco_sqp *h) { return h ? h->wv_rounds : -1; }
extern "C" int sco_sqp_last_tiers(const sco_sqp *h, double ms[2], long long iters[2], int launches[2]) {
  if (!h || !ms || !iters || !launches) return SCO_ERR_ARG;
  SCO_ON_DEVICE(h->device);
  unsigned long long wv_it = 0;
  if (sco_qp_wv_iters(h->qp1, &wv_it)) return SCO_ERR_DEVICE;
  ms[0] = h->wv_ms; ms[1] = h->last_ms[2] - h->wv_ms;
  iters[0] = (long long)wv_it; iters[1] = -1;          // the other kernels': total (sco_sqp_fetch: admm_iters) minus these
  launches[0] = h->wv_rounds; launches[1] = h->launches - h->wv_rounds;
  return SCO_OK;
}
extern "C" int sco_sqp_last_launches(const sco_sqp *h, int *launches, int *groups) {
  if (!h) return SCO_ERR_ARG;
  if (launches) *launches = h->launches;
  if (groups) *groups = h->groups_used;
  return SCO_OK;
}

extern "C" int sco_sqp_last_timing(const sco_sqp *h, double ms[5]) {
  if (!h || !ms) return SCO_ERR_ARG;
  memcpy(ms, h->last_ms, 5 * sizeof(double));
  return SCO_OK;
}
