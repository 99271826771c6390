// sco_internal.h -- shared between the QP layer (sco_qp.hip) and the SQP layer
// (sco_sqp.hip).  Not part of the public ABI (that is include/sco_hip.h).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../include/sco_hip.h"
#include "qp_plan.h"

#define SCO_INFTY 1e30
#define SCO_MIN_SCALING 1e-4
#define SCO_MAX_SCALING 1e4
#define SCO_RHO_MIN 1e-6
#define SCO_RHO_TOL 1e-4
#define SCO_RHO_EQ_OVER_RHO_INEQ 1e3

#define SCO_BLOCK 256   // threads per workgroup (4 wavefronts of 64)

void sco_set_error(const std::string &msg);
int sco_hip_fail(hipError_t e, const char *what);
#define SCO_HIP(call)                                                         \
  do {                                                                        \
    hipError_t _e = (call);                                                   \
    if (_e != hipSuccess) return sco_hip_fail(_e, #call);                     \
  } while (0)

// Every ABI entry point runs on its handle's device and puts the caller's current device back on return
// (a process that shares HIP with PyTorch or another library on a different device must not find its
// current device changed by a call into this one).
struct ScoDeviceGuard {
  int prev = -1; bool ok = false;
  explicit ScoDeviceGuard(int dev) {
    if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); }
    ok = hipSetDevice(dev) == hipSuccess;
  }
  ~ScoDeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  ScoDeviceGuard(const ScoDeviceGuard &) = delete;
  ScoDeviceGuard &operator=(const ScoDeviceGuard &) = delete;
};
#define SCO_ON_DEVICE(dev)                                                    \
  ScoDeviceGuard sco_guard_(dev);                                             \
  if (!sco_guard_.ok) return sco_hip_fail(hipGetLastError(), "hipSetDevice")

// Device-side view of one batched QP: plans (shared) + per-problem value arrays.
struct QpDev {
  int n, m, nnzP, nnzA, n_e, n_c, ncpl, nS, batch;
  // launch window (stream groups of the SQP loop): the kernels that honour it work on problem blockIdx.x + b0 and are
  // launched with nb workgroups (nb = 0: the whole batch from problem 0)
  int b0, nb;
  const int *list;   // or null: workgroup g works on problem list[g + b0] (< 0: none) instead of g + b0
  // shared index plans
  const int *Ap, *Ai, *Rp, *Rj, *Rpos, *Fp, *Fi, *Fpos, *Pdiag;
  const int *elim_var, *core_var, *elim_of, *core_of;
  const int *e_ptr, *pair_core, *pair_elim, *cp_ptr, *cp_row, *cp_pa, *cp_pe, *a_ptr, *a_pair;
  const int *s_a, *s_b, *s_ppos, *sa_ptr, *sa_row, *sa_pa, *sa_pb, *ss_ptr, *ss_k1, *ss_k2, *ss_e;
  // per-problem inputs (unscaled), problem-major
  double *Pval, *q, *Aval, *l, *u;
  int *w;
  // per-problem scaled data and factor, written by the setup kernel
  double *Ps, *As, *qs, *ls, *us, *D, *E, *cscale, *rho, *kee_inv, *cpl, *W;
  // per-problem outputs
  double *x, *y, *resid;
  int *status, *iters;
  // optional per-problem activity mask (NULL = all active); inactive problems
  // are skipped by both kernels
  const int *active;
  // parked solves of the row-local tier (time slicing): iterations done, scaled x / z / y / t' / g_e
  int *prog;
  double *sx, *sz, *sy, *st, *sg;
  // adaptive rho (opt-in): rho in use per problem, "rho changed: re-derive the cached right-hand sides on
  // resume" flag, "run setup" mask, number of updates
  double *rho_b;
  int *rflag, *smask, *nupd, *amask;   // amask: problems still parked (sco_qp_solve's own loop)
};

// ---- fast ADMM path (sco_admm_fast.hip) ------------------------------------
// Sliced-ELL image of a sparse operator: slices of 64 items, entry k of item
// (slice s, lane l) at base[s] + 64 k + l; idx = 16-bit index of the gathered
// vector element, src = position of the value in the flat per-problem array.
struct SellHost {
  int nitems = 0, total = 0;
  std::vector<int> base, width, src;
  std::vector<unsigned short> idx;
};
struct SellDev { const int *base, *width, *src; const unsigned short *idx; int total; };
struct FastHost { SellHost Ac, Ar, Ca, Ce; int TR = 1, TC = 2; bool capped = false; size_t lds_doubles = 0, lds_bytes = 0; };
struct FastDev { SellDev Ac, Ar, Ca, Ce; };

struct AdmmArgs {
  QpDev d;
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, check;
  int warm;     // start from the previous solution in d.x / d.y (row-local tier only)
  int slice;    // > 0: at most this many ADMM iterations per launch (row-local and generic kernels), see RlArgs
  int adaptive; // adaptive rho: every ad_interval iterations the kernel estimates rho; when it must change the solve
                // parks with the new rho in d.rho_b[b] and d.smask[b] = d.rflag[b] = 1
  int ad_interval;
  double ad_tol;
  const int *skip;   // or null: problems with skip[b] != 0 are left alone (the wavefront tier has taken them)
};

bool fast_plan_build(const QpPlan &pl, FastHost &fh);
int fast_upload(const FastHost &fh, std::vector<void *> &allocs, FastDev &fd);
int fast_launch(const AdmmArgs &a, const FastHost &fh, const FastDev &fd, hipStream_t st);

// ---- register-resident ADMM path (sco_admm_reg.hip) -------------------------
// Per-thread programs: slot s of thread t at [s * 512 + t]; slots = CW column
// entries, 2 x RW row entries, PX coupling entries.
struct RegHost {
  int TR = 1, TC = 2, CW = 0, RW = 0, PX = 0;
  SellHost Ac, Ar, Ca, Ce;
  size_t lds_bytes = 0;
  std::vector<int> role;
  std::vector<unsigned short> off;
};
struct RegDev { const unsigned short *off; const int *role; const int *srcAc, *srcAr, *srcCa, *srcCe; };
int reg_caps_for(const QpPlan &pl, int *CW, int *RW, int *PX);
bool reg_plan_build(const QpPlan &pl, int CW, int RW, int PX, RegHost &rh);
int reg_upload(const RegHost &rh, std::vector<void *> &allocs, RegDev &rd);
int reg_launch(const AdmmArgs &a, const RegHost &rh, const RegDev &rd, hipStream_t st);

// ---- row-local ADMM path (sco_admm_rl.hip): the coupling block never materialises
struct RlHost {
  int TR = 1, TC = 2, CW = 12;     // CW: slots for a core variable's column entries (12 or 16)
  int pcw = 0;                     // most P entries in a core variable's column (<= 4: cached in LDS for the termination test)
  int zpos = 0;                    // LDS position of the always-zero pair of the row vectors
  bool merged = false;             // closed thread assignment: phases (Y) and (1) share a wavefront, one barrier less per iteration
  bool aligned = false;            // closed assignment + the W rows of a wavefront's core columns in that wavefront: phases
                                   // (3), (Y), (1) are wavefront-local, ONE barrier per iteration (double-buffered right-hand side)
  bool split = false;              // a core column's entries in two neighbouring lanes (owner + helper), phase (1) on twice the lanes
  bool lay8 = false;               // 8 column groups of the W tile (3 x 18) with the open assignment
  int NS = 2;                      // row slots per thread (3: patterns with more than 1024 rows)
  SellHost Ac, Ar[3];
  size_t lds_bytes = 0;
  std::vector<unsigned short> off;
  std::vector<int> role, pc_ptr, pc_pos, pc_core;
};
struct RlDev { const unsigned short *off; const int *role, *srcAc, *srcAr[3], *pc_ptr, *pc_pos, *pc_core; };
bool rl_plan_build(const QpPlan &pl, RlHost &rh);
int rl_upload(const RlHost &rh, std::vector<void *> &allocs, RlDev &rd);
int rl_launch(const AdmmArgs &a, const RlHost &rh, const RlDev &rd, hipStream_t st);

// ---- wavefront tier (sco_admm_wv.hip): one wavefront per problem, block-tridiagonal core solve, four problems per CU
struct WvHost {
  int bs = 0, nb = 0, mid = 0, lpb = 0, n_extra = 0;    // block order, blocks, middle block, lanes per block, second single rows
  int BS = 8, NS = 1, NV = 1, NSTEP = 4;                 // instantiation: mat-vec width, hinge-row / variable slots per lane, sweep steps
  int cst_slots = 0;
  size_t g_doubles = 0, lds_doubles = 0, lds_bytes = 0;
  std::vector<int> tab;                                  // lane tables (wv_tab_layout)
};
// G: [batch][g_doubles] factor (pivot inverses, couplings); cst: [batch][cst_slots][64] constants of the termination test;
// wc: [batch] common weight of the hinge rows; ok: [batch] 1 = the problem's values have the penalty-QP structure (else the
// row-local kernel solves it: rl_need = its setup mask); scr: [batch][m + n] delta_y / delta_x of the last checked iteration
// w_ready: [batch] 1 = the problem's W buffer holds the dense inverse (the row-local kernel can run it), 0 = it still
// holds S (the wavefront tier factored this QP; qp_sweep_kernel has to run before the row-local kernel takes over)
// it_count: [1] ADMM iterations the tier's kernel has run since the last reset (diagnostics: per-tier rates of bench.py)
struct WvDev { const int *tab; double *G, *cst, *wc, *scr; int *ok, *rl_need, *w_ready, *pflag; unsigned long long *it_count; };
bool wv_plan_build(const QpPlan &pl, WvHost &wh);
int wv_upload(const WvHost &wh, int batch, int n, int m, std::vector<void *> &allocs, WvDev &wd);
int wv_launch_factor(const AdmmArgs &a, const int *setup_mask, const WvHost &wh, const WvDev &wd, hipStream_t st);
int wv_launch(const AdmmArgs &a, const WvHost &wh, const WvDev &wd, hipStream_t st);
// row-local round on a handle that also holds the wavefront tier: need[b] = setup_mask[b] or (active and W not ready)
int wv_launch_need(const AdmmArgs &a, const int *setup_mask, const WvDev &wd, hipStream_t st);

// ---- big tier (sco_qp_big.hip): everything in HBM/L2, 1024 threads per problem
struct BigHost {
  std::vector<int> row_elim, row_epos, er_ptr, er_row, free_rows, pc_ptr, pc_pos, pc_core;
  size_t ws_doubles = 0;
};
struct BigDev { const int *row_elim, *row_epos, *er_ptr, *er_row, *free_rows, *pc_ptr, *pc_pos, *pc_core; double *ws; };
bool big_plan_build(const QpPlan &pl, BigHost &bh);
int big_upload(const BigHost &bh, int batch, std::vector<void *> &allocs, BigDev &bd);
// structured variant of the big tier: block-tridiagonal core solve (factors in LDS) and
// dense row blocks addressed without index arrays
struct BtHost {
  int bs = 0, nb = 0, nchunks = 0, npart = 0;
  bool use_part = false;          // dense chunks reduce A' t in the wavefront (no product round trip)
  std::vector<int> cent;          // [n_c padded][4] column contributions: >= 0 partial index, <= -2 product position, -1 none
  size_t blk_doubles = 0, ws_doubles = 0, lds_bytes = 0;
  std::vector<int> ch_desc, it;     // it: 8 ints per chunk slot (e, j, r0, r1, ep0, ep1, core idx, core pos)
  // r04 (N1): block normal matrices on the matrix cores.  For blocks of order >= 12 whose hinge rows are dense, the sum over
  // those rows inside every diagonal-block entry of S is formed by v_mfma_f64_16x16x4 (one wavefront per block) between the
  // terms that come before them in the entry's list and the ones after: mf_id [nb][256] entry id of tile position (i, j),
  // i >= j (-1: none); mf_h0 / mf_h1 [nS] the run of hinge-row terms inside the entry's A' R A list (h0 = h1: none)
  bool use_mfma = false;
  int mf_why = 0;                 // 0 used, 1 switched off, 2 block order outside 12 .. 16, 3 hinge terms not one run, 4 rows differ between entries, 5 hole in a block, 6 no hinge rows
  std::vector<int> mf_id, mf_h0, mf_h1;
};
// park_part: [batch][npart] of a parked solve; cflag: [batch][nchunks] flags of the per-row constants that are known values
// for all rows of a dense chunk, ccon: [batch] the common row weight they refer to (written by qp_setup_big_kernel)
struct BtDev { const int *ch_desc, *it, *cent; double *blk, *ws, *park_part; int *cflag; double *ccon; const int *mf_id, *mf_h0, *mf_h1; };
bool bt_plan_build(const QpPlan &pl, const BigHost &bh, BtHost &th);
int bt_upload(const BtHost &th, int batch, std::vector<void *> &allocs, BtDev &td);
// th/td null = dense route
// setup_mask: the problems whose setup + factorisation run (a.d.active: the ones whose ADMM runs); the structured
// form honours a.slice / a.adaptive (park and resume), the dense form runs every solve to the end
int big_launch(const AdmmArgs &a, const int *setup_mask, int scaling, const int *Pp, const int *Pi, const BigHost &bh,
               const BigDev &bd, const BtHost *th, const BtDev *td, hipStream_t st, hipEvent_t ev_mid, hipEvent_t mid2);

struct sco_qp {
  int device = 0;
  QpPlan plan;
  BigHost big;
  BigDev bigd{};
  bool use_big = false;
  BtHost bt;
  BtDev btd{};
  bool use_bt = false;
  RlHost rl;
  RlDev rld{};
  bool use_rl = false;
  WvHost wv;
  WvDev wvd{};
  bool use_wv = false;
  int cus = 0;                         // compute units of the device (tier choice by batch size)
  RegHost reg;
  RegDev regd{};
  bool use_reg = false;
  FastHost fast;
  FastDev fastd{};
  bool use_fast = false;
  QpDev d{};
  const int *Pp_dev = nullptr, *Pi_dev = nullptr;   // device copies of the P triu pattern
  hipStream_t stream = nullptr;
  bool own_stream = false;
  hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
  std::vector<void *> allocs;
  bool loaded = false, solved_once = false;
  bool factor_cholesky = false;        // SCO_QP_FACTOR_CHOLESKY=1: W by Cholesky + triangular inverse (cross-check)
  size_t lds_setup = 0, lds_admm = 0;
  double last_ms[2] = {0, 0};
};

// Internal (device-pointer) entry points used by the SQP layer.
int sco_qp_create_on_stream(int device, int batch, int n, int m, const int *Pp, const int *Pi,
                            const int *Ap, const int *Ai, hipStream_t stream, sco_qp **out);
// Launch setup + ADMM on the handle's stream for the problems whose active flag
// is non-zero; no host synchronisation.
// `mid` (may be null) is recorded between the two kernels so callers can split the time.
int sco_qp_launch(sco_qp *qp, const sco_qp_settings *st, const int *active_dev, hipEvent_t mid);
// Time-sliced variant for the SQP loop: setup runs for the problems flagged in `setup_mask` (those that start
// a new QP), ADMM for the ones in `active_dev` for at most `slice` iterations (0 = to the end); a problem whose
// solve is not finished keeps status 0 and is resumed by the next call.  Returns the slice actually used
// through *sliced (0 if the handle's tier cannot park a solve).
// With st->adaptive_rho the slice is the rho-update interval and `setup_mask` names the problems that START a
// QP (device array, or SCO_MASK_ALL / SCO_MASK_NONE); setup then also runs for parked problems whose rho changed.
#define SCO_MASK_ALL ((const int *)(uintptr_t)1)
#define SCO_MASK_NONE ((const int *)(uintptr_t)2)
// `grp` (may be null): run only problems [b0, b0 + nb) and on grp->stream instead of the handle's own (stream groups of
// the SQP loop; row-local tier with the Gauss-Jordan inversion only, see sco_qp_supports_groups)
// tier: 0 = the handle decides (wavefront tier when the batch has at least SCO_WV_MIN_PER_CU problems per CU), 1 = row-local
// kernel, 2 = wavefront tier (handles that hold it; others ignore the field)
struct QpGroup { int b0, nb; hipStream_t stream; const int *list; int tier; };       // list: see QpDev (null = none)
int sco_qp_launch_sliced(sco_qp *qp, const sco_qp_settings *st, const int *setup_mask, const int *active_dev,
                         int slice, hipEvent_t mid, int *sliced, const QpGroup *grp = nullptr);
bool sco_qp_supports_groups(const sco_qp *qp, const sco_qp_settings *st);
int sco_qp_adaptive_interval(const sco_qp_settings *st);
int sco_wv_min_live(int cus);      // fewest live problems for which a round runs on the wavefront tier
bool sco_qp_has_wv(const sco_qp *qp, const sco_qp_settings *st);
int sco_qp_wv_iters(const sco_qp *qp, unsigned long long *out);      // iterations run by the wavefront kernel since the reset
void sco_qp_wv_iters_reset(sco_qp *qp, hipStream_t st);   // the handle holds the wavefront tier and these settings can use it
bool sco_qp_can_adapt(const sco_qp *qp);   // false: this handle sits on the dense global-memory tier, which cannot park a solve
