// sco_qp.hip -- batched OSQP-style ADMM for gfx950 (MI355X), QP layer of libsco_hip.
//
// Replaces the third-party OSQP solve the reference performs once per QP at
// /root/reference/sco_py/sco_osqp/osqp_utils.py:195-216.  Algorithm (scaling,
// rho selection, ADMM recurrences, termination and infeasibility tests) follows
// the published OSQP method with the 0.6-series defaults; see oracle/osqp_ref.c
// for the CPU restatement every result here is tested against.
//
// Mapping to the hardware
//   * one workgroup (256 threads = 4 wavefronts) per problem, grid = batch;
//   * qp_setup_kernel: Ruiz equilibration, rho vector, reduced matrix assembly
//     (scaled data in LDS, the core Schur complement S into the W buffer);
//   * qp_sweep_kernel (1024 threads): W = S^-1 by Gauss-Jordan sweeps with S in registers
//     (qp_factor_kernel: the same by Cholesky + triangular inverse in LDS, cross-check);
//   * qp_admm_kernel: the whole ADMM loop of one problem inside one launch with
//     every iterate (x, z, y), the scaled A values and the coupling block held
//     in LDS; HBM is touched once to load the problem, once per iteration for
//     the dense core inverse W (L2 resident), and once to store the answer;
//   * residual norms use wavefront shuffles + one LDS hop across the 4 waves;
//   * no atomics, fixed summation orders: results are run-to-run deterministic.
#include "sco_internal.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

// --------------------------------------------------------------------------
// error plumbing
// --------------------------------------------------------------------------
static thread_local std::string g_err;
void sco_set_error(const std::string &msg) { g_err = msg; }
int sco_hip_fail(hipError_t e, const char *what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return SCO_ERR_DEVICE;
}
extern "C" const char *sco_last_error(void) { return g_err.c_str(); }
extern "C" int sco_version(void) { return 100; }

extern "C" int sco_device_count(int *count) {
  if (!count) return SCO_ERR_ARG;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { *count = 0; (void)hipGetLastError(); return SCO_OK; }
  *count = n;
  return SCO_OK;
}

extern "C" void sco_qp_default_settings(sco_qp_settings *s) {
  if (!s) return;
  s->rho = 0.1; s->sigma = 5e-10; s->alpha = 1.6;
  s->eps_abs = 1e-6; s->eps_rel = 1e-9;
  s->eps_prim_inf = 1e-4; s->eps_dual_inf = 1e-4;
  s->max_iter = 100000; s->check_termination = 25; s->scaling = 10; s->warm_start = 0;
  s->adaptive_rho = 0; s->adaptive_rho_interval = 0; s->adaptive_rho_tolerance = 5.0;
}

// --------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------
#define NWAVE (SCO_BLOCK / 64)

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Reduce NR per-thread values across the workgroup; every thread gets the
// result.  `red` is NWAVE*NR doubles of LDS.  Fixed tree => deterministic.
template <int NR, bool IS_MAX>
__device__ __forceinline__ void block_reduce(double (&v)[NR], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = IS_MAX ? wave_max(v[k]) : wave_sum(v[k]);
  __syncthreads();   // protect `red` against a previous use
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NR; k++) red[wv * NR + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NR; k++) {
    double r = red[k];
#pragma unroll
    for (int w = 1; w < NWAVE; w++) r = IS_MAX ? fmax(r, red[w * NR + k]) : r + red[w * NR + k];
    v[k] = r;
  }
}

__device__ __forceinline__ double limit_scaling(double v) {
  v = v < SCO_MIN_SCALING ? 1.0 : v;
  return v > SCO_MAX_SCALING ? SCO_MAX_SCALING : v;
}

__host__ __device__ __forceinline__ size_t tri_idx(int i, int j) {   // packed lower, j <= i
  return (size_t)i * (i + 1) / 2 + j;
}

// --------------------------------------------------------------------------
// setup kernel: scale, rho, reduced matrix, factor, inverse
// --------------------------------------------------------------------------
struct SetupArgs {
  QpDev d;
  const int *Pp, *Pi;
  double rho, sigma;
  int scaling;
  int per_problem_rho;      // adaptive rho: the problem's current rho is d.rho_b[b]
};

#ifdef SCO_STAMP
// diagnostic build only: cycles per phase of the setup of problem 0 (never compiled into the product)
__device__ double g_setup_stamp[16];
extern "C" int sco_debug_setup_stamps(double *out) {
  return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_setup_stamp), 16 * sizeof(double)) == hipSuccess ? 0 : -2;
}
#define SSTAMP(k) { if (b == 0 && tid == 0) { const long long now_ = __builtin_readcyclecounter(); g_setup_stamp[k] += (double)(now_ - sst_t); sst_t = now_; } }
#else
#define SSTAMP(k)
#endif

__global__ __launch_bounds__(SCO_BLOCK) void qp_setup_kernel(SetupArgs a) {
  const QpDev &d = a.d;
  const int b = d.list ? d.list[blockIdx.x + d.b0] : (int)blockIdx.x + d.b0, tid = threadIdx.x;
  if (b < 0 || (d.active && !d.active[b])) return;
#ifdef SCO_STAMP
  long long sst_t = __builtin_readcyclecounter();
  if (b == 0 && tid == 0) g_setup_stamp[15] += 1.0;
#endif
  const double rho0 = a.per_problem_rho ? d.rho_b[b] : a.rho;
  const int n = d.n, m = d.m, nnzP = d.nnzP, nnzA = d.nnzA, n_e = d.n_e, n_c = d.n_c, ncpl = d.ncpl;

  extern __shared__ double lds[];
  double *Ps = lds;                 // nnzP
  double *As = Ps + nnzP;           // nnzA
  double *qs = As + nnzA;           // n
  double *D = qs + n;               // n
  double *E = D + n;                // m
  double *Dt = E + m;               // n   (later: kinv[n_e])
  double *Et = Dt + n;              // m   (later: rw[m])
  double *cpl = Et + m;             // ncpl
  double *red = cpl + ncpl;         // NWAVE * 2

  const double *Pval = d.Pval + (size_t)b * nnzP;
  const double *Aval = d.Aval + (size_t)b * nnzA;
  for (int t = tid; t < nnzP; t += SCO_BLOCK) Ps[t] = Pval[t];
  for (int t = tid; t < nnzA; t += SCO_BLOCK) As[t] = Aval[t];
  for (int j = tid; j < n; j += SCO_BLOCK) { qs[j] = d.q[(size_t)b * n + j]; D[j] = 1.0; }
  for (int i = tid; i < m; i += SCO_BLOCK) E[i] = 1.0;
  double c = 1.0;
  __syncthreads();
  SSTAMP(0)

  for (int it = 0; it < a.scaling; it++) {
    // inf-norms of the columns of [[P, A'], [A, 0]]
    for (int j = tid; j < n; j += SCO_BLOCK) {
      double v = 0.0;
      for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) v = fmax(v, fabs(Ps[d.Fpos[t]]));
      for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) v = fmax(v, fabs(As[t]));
      Dt[j] = 1.0 / sqrt(limit_scaling(v));
    }
    for (int i = tid; i < m; i += SCO_BLOCK) {
      double v = 0.0;
      for (int t = d.Rp[i]; t < d.Rp[i + 1]; t++) v = fmax(v, fabs(As[d.Rpos[t]]));
      Et[i] = 1.0 / sqrt(limit_scaling(v));
    }
    __syncthreads();
    // P <- Dt P Dt, A <- Et A Dt, q <- Dt q
    for (int j = tid; j < n; j += SCO_BLOCK) {
      const double dj = Dt[j];
      for (int t = a.Pp[j]; t < a.Pp[j + 1]; t++) Ps[t] = (Ps[t] * Dt[a.Pi[t]]) * dj;
      for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) As[t] = (As[t] * Et[d.Ai[t]]) * dj;
      qs[j] *= dj; D[j] *= dj;
    }
    for (int i = tid; i < m; i += SCO_BLOCK) E[i] *= Et[i];
    __syncthreads();
    // cost normalisation: c = 1 / max(mean col-norm of P, ||q||_inf)
    double r2[2] = {0.0, 0.0};
    for (int j = tid; j < n; j += SCO_BLOCK) {
      double v = 0.0;
      for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) v = fmax(v, fabs(Ps[d.Fpos[t]]));
      r2[0] += v;
    }
    {
      double s1[1] = {r2[0]};
      block_reduce<1, false>(s1, red);
      r2[0] = s1[0];
      double mq = 0.0;
      for (int j = tid; j < n; j += SCO_BLOCK) mq = fmax(mq, fabs(qs[j]));
      double s2[1] = {mq};
      block_reduce<1, true>(s2, red);
      r2[1] = s2[0];
    }
    double ct = n > 0 ? r2[0] / (double)n : 0.0;
    ct = fmax(ct, limit_scaling(r2[1]));
    ct = 1.0 / limit_scaling(ct);
    for (int t = tid; t < nnzP; t += SCO_BLOCK) Ps[t] *= ct;
    for (int j = tid; j < n; j += SCO_BLOCK) qs[j] *= ct;
    c *= ct;
    __syncthreads();
  }

  SSTAMP(1)
  // scaled bounds, rho vector; Et becomes rw = w * rho
  {
    double *ls = d.ls + (size_t)b * m, *us = d.us + (size_t)b * m, *rho = d.rho + (size_t)b * m;
    const double *l = d.l + (size_t)b * m, *u = d.u + (size_t)b * m;
    const int *w = d.w + (size_t)b * m;
    for (int i = tid; i < m; i += SCO_BLOCK) {
      double li = fmax(l[i], -SCO_INFTY) * E[i], ui = fmin(u[i], SCO_INFTY) * E[i];
      ls[i] = li; us[i] = ui;
      double r;
      if (li < -SCO_INFTY * SCO_MIN_SCALING && ui > SCO_INFTY * SCO_MIN_SCALING) r = SCO_RHO_MIN;
      else if (ui - li < SCO_RHO_TOL) r = SCO_RHO_EQ_OVER_RHO_INEQ * rho0;
      else r = rho0;
      rho[i] = r;
      Et[i] = r * (double)w[i];
    }
  }
  // publish the scaled problem
  {
    double *gPs = d.Ps + (size_t)b * nnzP, *gAs = d.As + (size_t)b * nnzA;
    for (int t = tid; t < nnzP; t += SCO_BLOCK) gPs[t] = Ps[t];
    for (int t = tid; t < nnzA; t += SCO_BLOCK) gAs[t] = As[t];
    for (int j = tid; j < n; j += SCO_BLOCK) { d.qs[(size_t)b * n + j] = qs[j]; d.D[(size_t)b * n + j] = D[j]; }
    for (int i = tid; i < m; i += SCO_BLOCK) d.E[(size_t)b * m + i] = E[i];
    if (tid == 0) d.cscale[b] = c;
  }
  __syncthreads();

  SSTAMP(2)
  // ---- K_EE^-1 (diagonal) and the coupling block K_CE ----------------------
  double *kinv = Dt;
  for (int e = tid; e < n_e; e += SCO_BLOCK) {
    const int ve = d.elim_var[e];
    double v = a.sigma;
    if (d.Pdiag[ve] >= 0) v += Ps[d.Pdiag[ve]];
    for (int t = d.Ap[ve]; t < d.Ap[ve + 1]; t++) v += Et[d.Ai[t]] * As[t] * As[t];
    v = 1.0 / v;
    kinv[e] = v; d.kee_inv[(size_t)b * n_e + e] = v;
  }
  for (int k = tid; k < ncpl; k += SCO_BLOCK) {
    double v = 0.0;
    for (int t = d.cp_ptr[k]; t < d.cp_ptr[k + 1]; t++) v += Et[d.cp_row[t]] * As[d.cp_pa[t]] * As[d.cp_pe[t]];
    cpl[k] = v; d.cpl[(size_t)b * ncpl + k] = v;
  }
  // ---- S = K_CC - K_CE K_EE^-1 K_EC, packed lower triangle, into the problem's W buffer (qp_factor_kernel
  //      turns it into W = S^-1 in place) ---------------------------------------------------------------
  const size_t ntri = (size_t)n_c * (n_c + 1) / 2;
  double *Sg = d.W + (size_t)b * n_c * n_c;
  for (size_t t = tid; t < ntri; t += SCO_BLOCK) Sg[t] = 0.0;
  __syncthreads();
  SSTAMP(3)
  for (int id = tid; id < d.nS; id += SCO_BLOCK) {
    const int sa = d.s_a[id], sb = d.s_b[id];
    double v = (sa == sb) ? a.sigma : 0.0;
    if (d.s_ppos[id] >= 0) v += Ps[d.s_ppos[id]];
    for (int t = d.sa_ptr[id]; t < d.sa_ptr[id + 1]; t++) v += Et[d.sa_row[t]] * As[d.sa_pa[t]] * As[d.sa_pb[t]];
    for (int t = d.ss_ptr[id]; t < d.ss_ptr[id + 1]; t++) v -= cpl[d.ss_k1[t]] * cpl[d.ss_k2[t]] * kinv[d.ss_e[t]];
    Sg[tri_idx(sa, sb)] = v;
  }
  SSTAMP(4)
}

// --------------------------------------------------------------------------
// factor kernel: W = S^-1 of the dense core (order n_c <= 256), one workgroup of 1024 threads per problem.
// S arrives as a packed lower triangle in the problem's W buffer and lives in LDS from then on:
//   Cholesky S = L L' (left-looking), M = L^-1 in place (last column first), W = M' M -> global.
// Both triangular loops are LDS-latency bound (one dependent load-multiply-add chain per row), so a row's
// k-range is split over LPR adjacent lanes (LPR = the power of two that fills the workgroup: few rows are left
// exactly when the chains are long) and added with lane shuffles in a fixed order; 16 wavefronts hide the rest.
// (The 256-thread single-chain version took 2.9 M cycles per 140 x 140 core, profiles/r01_setup_stamps.txt.)
// --------------------------------------------------------------------------
#define SCO_FACTOR_BLOCK 1024

__device__ __forceinline__ int lanes_per_row(int rows) {
  int l = 64;
  while (l > 1 && l * rows > SCO_FACTOR_BLOCK) l >>= 1;
  return l;
}

__global__ __launch_bounds__(SCO_FACTOR_BLOCK) void qp_factor_kernel(QpDev d) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (d.active && !d.active[b]) return;
  const int n_c = d.n_c;
  if (n_c == 0) return;
#ifdef SCO_STAMP
  long long sst_t = __builtin_readcyclecounter();
#endif
  extern __shared__ double lds[];
  double *Lp = lds;                                   // n_c (n_c + 1) / 2
  double *dg = Lp + (size_t)n_c * (n_c + 1) / 2;      // n_c: diagonal of L (Lp keeps S_jj there until the inverse)
  const size_t ntri = (size_t)n_c * (n_c + 1) / 2;
  double *W = d.W + (size_t)b * n_c * n_c;
  for (size_t t = tid; t < ntri; t += SCO_FACTOR_BLOCK) Lp[t] = W[t];
  __syncthreads();

  // ---- Cholesky: one barrier per column (a row's owner lane reads S_ij and writes L_ij itself; the pivot goes to dg)
  for (int j = 0; j < n_c; j++) {
    const int rows = n_c - j, lpr = lanes_per_row(rows);
    const int r = tid / lpr, part = tid - r * lpr, i = j + r;
    const bool act = r < rows;
    const double *rj = Lp + tri_idx(j, 0);
    const double *ri = Lp + tri_idx(act ? i : j, 0);
    double p0 = 0.0, p1 = 0.0, s0 = 0.0, s1 = 0.0;
    int k = part;
    for (; k + lpr < j; k += 2 * lpr) {
      const double a0 = rj[k], a1 = rj[k + lpr], b0 = ri[k], b1 = ri[k + lpr];
      p0 += a0 * a0; p1 += a1 * a1; s0 += b0 * a0; s1 += b1 * a1;
    }
    if (k < j) { const double a0 = rj[k]; p0 += a0 * a0; s0 += ri[k] * a0; }
    double p = p0 + p1, sm = s0 + s1;
    for (int o = lpr >> 1; o > 0; o >>= 1) { p += __shfl_xor(p, o); sm += __shfl_xor(sm, o); }
    if (act && part == 0) {
      const double piv = sqrt(rj[j] - p);
      if (i == j) dg[j] = piv; else Lp[tri_idx(i, j)] = (ri[j] - sm) / piv;
    }
    __syncthreads();
  }
  SSTAMP(5)

  // ---- M = L^-1 in place, last column first:  M_ij = -(sum_{k=j+1..i} M_ik L_kj) / L_jj
  for (int j = n_c - 1; j >= 0; j--) {
    const int rows = n_c - 1 - j, lpr = lanes_per_row(rows > 0 ? rows : 1);
    const int r = tid / lpr, part = tid - r * lpr, i = j + 1 + r;
    const bool act = r < rows;
    const double ljj = dg[j];
    double s0 = 0.0, s1 = 0.0;
    if (act) {
      const double *ri = Lp + tri_idx(i, 0);
      int k = j + 1 + part;
      for (; k + lpr <= i; k += 2 * lpr) {
        s0 += ri[k] * Lp[tri_idx(k, j)]; s1 += ri[k + lpr] * Lp[tri_idx(k + lpr, j)];
      }
      if (k <= i) s0 += ri[k] * Lp[tri_idx(k, j)];
    }
    double sm = s0 + s1;
    for (int o = lpr >> 1; o > 0; o >>= 1) sm += __shfl_xor(sm, o);
    __syncthreads();                              // every read of column j of L is done
    if (act && part == 0) Lp[tri_idx(i, j)] = -sm / ljj;
    if (tid == 0) Lp[tri_idx(j, j)] = 1.0 / ljj;
    __syncthreads();
  }
  SSTAMP(6)

  // ---- W = M' M (dense, symmetric) -> global
  for (size_t p = tid; p < ntri; p += SCO_FACTOR_BLOCK) {
    int ia = (int)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (tri_idx(ia + 1, 0) <= p) ia++;
    while (tri_idx(ia, 0) > p) ia--;
    const int ib = (int)(p - tri_idx(ia, 0));
    double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
    int k = ia;
    size_t c = tri_idx(k, 0);                     // row k of M
    for (; k + 3 < n_c; k += 4) {
      const size_t c1 = c + k + 1, c2 = c1 + k + 2, c3 = c2 + k + 3;
      s0 += Lp[c + ia] * Lp[c + ib]; s1 += Lp[c1 + ia] * Lp[c1 + ib];
      s2 += Lp[c2 + ia] * Lp[c2 + ib]; s3 += Lp[c3 + ia] * Lp[c3 + ib];
      c = c3 + k + 4;
    }
    for (; k < n_c; k++) { s0 += Lp[c + ia] * Lp[c + ib]; c += k + 1; }
    const double s = (s0 + s1) + (s2 + s3);
    W[(size_t)ia * n_c + ib] = s;
    W[(size_t)ib * n_c + ia] = s;
  }
  SSTAMP(7)
}

// --------------------------------------------------------------------------
// invert kernel (default): W = S^-1 by symmetric Gauss-Jordan sweeps with the matrix in REGISTERS.
// Every thread of the 1024 owns up to three 4 x 4 tiles of the lower triangle for the whole run (loaded straight
// from the packed S in the W buffer); step k publishes column k (n values, double-buffered in LDS: one barrier per
// step), every thread reads its eight entries of it and updates its tiles:
//     a_kk <- -1 / a_kk,   a_ik <- a_ik / a_kk,   a_ij <- a_ij - a_ik a_jk / a_kk
// after n sweeps the registers hold -S^-1 (pivots are the Schur complements of an SPD matrix: no pivoting needed).
// No dependent load-multiply-add chains, no LDS traffic for the matrix: ~100 k cycles for a 140 x 140 core against
// 1.0 M for the Cholesky route below (profiles/r01_setup_stamps.txt), which stays as a cross-check
// (SCO_QP_FACTOR_CHOLESKY=1 at handle creation).
// --------------------------------------------------------------------------
// a_ik for i >= k from the tiles in block column kb, a_kj = a_jk for j < k from the tiles in block row kb
template <int NT, int KK>
__device__ __forceinline__ void sweep_publish(const double (&t)[NT][4][4], const int (&bi)[NT], const int (&bj)[NT],
                                              int kb, int k, int n, double *cur) {
#pragma unroll
  for (int u = 0; u < NT; u++) {
    if (bi[u] < 0) continue;
    if (bj[u] == kb) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int i = 4 * bi[u] + r;
        if (i < n && i >= k) cur[i] = t[u][r][KK];
      }
    }
    if (bi[u] == kb) {
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int j = 4 * bj[u] + c;
        if (j < k) cur[j] = t[u][KK][c];
      }
    }
  }
}

// the sweep itself: every entry a_ij -= a_ik a_jk / a_kk, then the entries of column / row k and the pivot are
// overwritten with their own rule (entries above the diagonal of a diagonal tile are scratch values nobody reads)
template <int NT, int KK>
__device__ __forceinline__ void sweep_update(double (&t)[NT][4][4], const int (&bi)[NT], const int (&bj)[NT],
                                             int kb, int k, int n, const double *cur) {
  const double dinv = 1.0 / cur[k];
#pragma unroll
  for (int u = 0; u < NT; u++) {
    if (bi[u] < 0) continue;
    double ai[4], aj[4];
#pragma unroll
    for (int r = 0; r < 4; r++) { const int i = 4 * bi[u] + r; ai[r] = cur[i < n ? i : n - 1] * dinv; }
#pragma unroll
    for (int c = 0; c < 4; c++) aj[c] = cur[4 * bj[u] + c];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) t[u][r][c] -= ai[r] * aj[c];
    if (bj[u] == kb) {
#pragma unroll
      for (int r = 0; r < 4; r++) t[u][r][KK] = ai[r];
    }
    if (bi[u] == kb) {
#pragma unroll
      for (int c = 0; c < 4; c++) t[u][KK][c] = aj[c] * dinv;
      if (bj[u] == kb) t[u][KK][KK] = -dinv;
    }
  }
}

// SCO_SWEEP_NT tiles per thread (template): 1 up to order 176, 2 up to 252, 3 up to 256 (16 wavefronts leave 128
// registers per thread, a tile takes 32)
template <int SCO_SWEEP_NT>
__global__ __launch_bounds__(SCO_FACTOR_BLOCK) void qp_sweep_kernel(QpDev d) {
  const int b = d.list ? d.list[blockIdx.x + d.b0] : (int)blockIdx.x + d.b0, tid = threadIdx.x;
  if (b < 0 || (d.active && !d.active[b])) return;
  const int n = d.n_c;
  if (n == 0) return;
#ifdef SCO_STAMP
  long long sst_t = __builtin_readcyclecounter();
#endif
  __shared__ double ck[2][264];
  double *W = d.W + (size_t)b * n * n;
  const int nb = (n + 3) >> 2, ntiles = nb * (nb + 1) / 2;
  int bi[SCO_SWEEP_NT], bj[SCO_SWEEP_NT];
  double t[SCO_SWEEP_NT][4][4];
#pragma unroll
  for (int u = 0; u < SCO_SWEEP_NT; u++) {
    const int T = tid + u * SCO_FACTOR_BLOCK;
    bi[u] = -1; bj[u] = 0;
    if (T < ntiles) {
      int r = (int)((sqrt(8.0 * (double)T + 1.0) - 1.0) * 0.5);
      while ((r + 1) * (r + 2) / 2 <= T) r++;
      while (r * (r + 1) / 2 > T) r--;
      bi[u] = r; bj[u] = T - r * (r + 1) / 2;
    }
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int i = 4 * bi[u] + r, j = 4 * bj[u] + c;
        t[u][r][c] = (bi[u] >= 0 && j <= i && i < n) ? W[tri_idx(i, j)] : 0.0;
      }
  }
  __syncthreads();
  SSTAMP(5)
  for (int k = 0; k < n; k++) {
    double *cur = ck[k & 1];
    const int kb = k >> 2, kk = k & 3;
    // kk = k mod 4 selects a compile-time register row / column (a run-time index would send the tiles to scratch)
    switch (kk) {
      case 0: sweep_publish<SCO_SWEEP_NT, 0>(t, bi, bj, kb, k, n, cur); break;
      case 1: sweep_publish<SCO_SWEEP_NT, 1>(t, bi, bj, kb, k, n, cur); break;
      case 2: sweep_publish<SCO_SWEEP_NT, 2>(t, bi, bj, kb, k, n, cur); break;
      default: sweep_publish<SCO_SWEEP_NT, 3>(t, bi, bj, kb, k, n, cur); break;
    }
    __syncthreads();
    switch (kk) {
      case 0: sweep_update<SCO_SWEEP_NT, 0>(t, bi, bj, kb, k, n, cur); break;
      case 1: sweep_update<SCO_SWEEP_NT, 1>(t, bi, bj, kb, k, n, cur); break;
      case 2: sweep_update<SCO_SWEEP_NT, 2>(t, bi, bj, kb, k, n, cur); break;
      default: sweep_update<SCO_SWEEP_NT, 3>(t, bi, bj, kb, k, n, cur); break;
    }
    // the next step publishes into the other buffer; the barrier of step k + 1 separates this step's reads of
    // `cur` from step k + 2's writes to it
  }
  SSTAMP(6)
#pragma unroll
  for (int u = 0; u < SCO_SWEEP_NT; u++) {
    if (bi[u] < 0) continue;
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
      for (int c = 0; c < 4; c++) {
        const int i = 4 * bi[u] + r, j = 4 * bj[u] + c;
        if (j <= i && i < n) {
          const double v = -t[u][r][c];
          W[(size_t)i * n + j] = v;
          W[(size_t)j * n + i] = v;
        }
      }
  }
  SSTAMP(7)
}

// --------------------------------------------------------------------------
// ADMM kernel
// --------------------------------------------------------------------------
struct AdmmLds {
  double *As, *qs, *ls, *us, *rho, *x, *z, *y, *t, *xt, *ge, *r, *xc, *kinv, *cpl, *sdy, *sdx, *red;
  int *w;
};

__host__ __device__ inline size_t admm_lds_doubles(int n, int m, int nnzA, int n_e, int n_c, int ncpl) {
  return (size_t)nnzA + n + 2 * (size_t)m + m + n + 2 * (size_t)m + m + n + n_e + 2 * (size_t)n_c + n_e + ncpl + m + n + NWAVE * 8;
}

// Termination test of one ADMM iterate (all threads of the workgroup take part
// and return the same value): 0 = keep iterating, otherwise an SCO_QP_* status.
__device__ int admm_check(const AdmmArgs &a, const AdmmLds &s, int b, int approximate,
                          double cscale, double *pri_out, double *dua_out) {
  const QpDev &d = a.d;
  const int tid = threadIdx.x, n = d.n, m = d.m;
  const double *Ps = d.Ps + (size_t)b * d.nnzP;
  const double *Dg = d.D + (size_t)b * n, *Eg = d.E + (size_t)b * m;
  const double cinv = 1.0 / cscale;
  double ea = a.eps_abs, er = a.eps_rel, epi = a.eps_prim_inf, edi = a.eps_dual_inf;
  if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }

  // rows: primal residual and its scale
  double v[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = tid; i < m; i += SCO_BLOCK) {
    double ax = 0.0;
    for (int t = d.Rp[i]; t < d.Rp[i + 1]; t++) ax += s.As[d.Rpos[t]] * s.x[d.Rj[t]];
    const double ei = 1.0 / Eg[i];
    v[0] = fmax(v[0], fabs(ei * (ax - s.z[i])));
    v[1] = fmax(v[1], fabs(ei * s.z[i]));
    v[2] = fmax(v[2], fabs(ei * ax));
  }
  // columns: dual residual and its scale
  for (int j = tid; j < n; j += SCO_BLOCK) {
    double px = 0.0, aty = 0.0;
    for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * s.x[d.Fi[t]];
    for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) { const int i = d.Ai[t]; aty += s.As[t] * s.y[i] * (double)s.w[i]; }
    const double dj = 1.0 / Dg[j];
    v[3] = fmax(v[3], fabs(dj * (s.qs[j] + px + aty)));
    v[4] = fmax(v[4], fabs(dj * s.qs[j]));
    v[5] = fmax(v[5], fabs(dj * aty));
    v[6] = fmax(v[6], fabs(dj * px));
  }
  block_reduce<7, true>(v, s.red);
  const double pri = v[0], dua = cinv * v[3];
  *pri_out = pri; *dua_out = dua;
  if (!(pri <= SCO_INFTY) || !(dua <= SCO_INFTY)) return SCO_QP_NON_CVX;
  const double eps_p = ea + er * fmax(v[1], v[2]);
  const double eps_d = ea + er * cinv * fmax(v[4], fmax(v[5], v[6]));
  const bool prim_ok = (m == 0) || (pri < eps_p);
  const bool dual_ok = dua < eps_d;
  if (prim_ok && dual_ok) return approximate ? SCO_QP_SOLVED_INACCURATE : SCO_QP_SOLVED;

  // ---- primal infeasibility certificate from delta_y --------------------------
  if (!prim_ok) {
    double r2[1] = {0.0};
    for (int i = tid; i < m; i += SCO_BLOCK) {
      double dy = s.sdy[i];
      const double li = s.ls[i], ui = s.us[i];
      if (ui > SCO_INFTY * SCO_MIN_SCALING) {
        if (li < -SCO_INFTY * SCO_MIN_SCALING) dy = 0.0; else dy = fmin(dy, 0.0);
      } else if (li < -SCO_INFTY * SCO_MIN_SCALING) dy = fmax(dy, 0.0);
      s.sdy[i] = dy;
      r2[0] = fmax(r2[0], fabs(Eg[i] * dy));
    }
    block_reduce<1, true>(r2, s.red);
    const double ndy = r2[0];
    if (ndy > epi) {
      double lhs[1] = {0.0};
      for (int i = tid; i < m; i += SCO_BLOCK) {
        const double dy = s.sdy[i];
        lhs[0] += (double)s.w[i] * (s.us[i] * fmax(dy, 0.0) + s.ls[i] * fmin(dy, 0.0));
      }
      block_reduce<1, false>(lhs, s.red);
      if (lhs[0] < -epi * ndy) {
        double nat[1] = {0.0};
        for (int j = tid; j < n; j += SCO_BLOCK) {
          double aty = 0.0;
          for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) { const int i = d.Ai[t]; aty += s.As[t] * s.sdy[i] * (double)s.w[i]; }
          nat[0] = fmax(nat[0], fabs(aty / Dg[j]));
        }
        block_reduce<1, true>(nat, s.red);
        if (nat[0] < epi * ndy) return approximate ? SCO_QP_PRIMAL_INFEASIBLE_INACCURATE : SCO_QP_PRIMAL_INFEASIBLE;
      }
    }
  }
  // ---- dual infeasibility certificate from delta_x ------------------------------
  if (!dual_ok) {
    double r1[1] = {0.0};
    for (int j = tid; j < n; j += SCO_BLOCK) r1[0] = fmax(r1[0], fabs(Dg[j] * s.sdx[j]));
    block_reduce<1, true>(r1, s.red);
    const double ndx = r1[0];
    if (ndx > edi) {
      double qdx[1] = {0.0};
      for (int j = tid; j < n; j += SCO_BLOCK) qdx[0] += s.qs[j] * s.sdx[j];
      block_reduce<1, false>(qdx, s.red);
      if (qdx[0] < -cscale * edi * ndx) {
        double npx[1] = {0.0};
        for (int j = tid; j < n; j += SCO_BLOCK) {
          double px = 0.0;
          for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * s.sdx[d.Fi[t]];
          npx[0] = fmax(npx[0], fabs(px / Dg[j]));
        }
        block_reduce<1, true>(npx, s.red);
        if (npx[0] < cscale * edi * ndx) {
          double bad[1] = {0.0};
          for (int i = tid; i < m; i += SCO_BLOCK) {
            double adx = 0.0;
            for (int t = d.Rp[i]; t < d.Rp[i + 1]; t++) adx += s.As[d.Rpos[t]] * s.sdx[d.Rj[t]];
            adx /= Eg[i];
            if ((s.us[i] < SCO_INFTY * SCO_MIN_SCALING && adx > edi * ndx) ||
                (s.ls[i] > -SCO_INFTY * SCO_MIN_SCALING && adx < -edi * ndx)) bad[0] = 1.0;
          }
          block_reduce<1, true>(bad, s.red);
          if (bad[0] == 0.0) return approximate ? SCO_QP_DUAL_INFEASIBLE_INACCURATE : SCO_QP_DUAL_INFEASIBLE;
        }
      }
    }
  }
  return 0;
}

// OSQP's rho estimate from the SCALED iterates in LDS (osqp 0.6 auxil.c compute_rho_estimate, as recalled; the
// library is not available here, see oracle/osqp_ref.c):
//   rho sqrt( (|Ax - z| / (max(|z|, |Ax|) + 1e-10)) / (|Px + q + A'y| / (max(|q|, |A'y|, |Px|) + 1e-10) + 1e-10) )
// clipped to [1e-6, 1e6].  Fixed summation order: the value does not depend on scheduling.
__device__ double admm_rho_estimate(const AdmmArgs &a, const AdmmLds &s, int b, double rho) {
  const QpDev &d = a.d;
  const int tid = threadIdx.x, n = d.n, m = d.m;
  const double *Ps = d.Ps + (size_t)b * d.nnzP;
  double v[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = tid; i < m; i += SCO_BLOCK) {
    double ax = 0.0;
    for (int t = d.Rp[i]; t < d.Rp[i + 1]; t++) ax += s.As[d.Rpos[t]] * s.x[d.Rj[t]];
    v[0] = fmax(v[0], fabs(ax - s.z[i])); v[1] = fmax(v[1], fabs(s.z[i])); v[2] = fmax(v[2], fabs(ax));
  }
  for (int j = tid; j < n; j += SCO_BLOCK) {
    double px = 0.0, aty = 0.0;
    for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * s.x[d.Fi[t]];
    for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) aty += s.As[t] * ((double)s.w[d.Ai[t]] * s.y[d.Ai[t]]);
    v[3] = fmax(v[3], fabs(px + s.qs[j] + aty)); v[4] = fmax(v[4], fabs(s.qs[j]));
    v[5] = fmax(v[5], fabs(aty)); v[6] = fmax(v[6], fabs(px));
  }
  block_reduce<7, true>(v, s.red);
  const double pri = v[0] / (fmax(v[1], v[2]) + 1e-10);
  const double dua = v[3] / (fmax(v[4], fmax(v[5], v[6])) + 1e-10);
  return fmin(fmax(rho * sqrt(pri / (dua + 1e-10)), SCO_RHO_MIN), 1e6);
}

__global__ __launch_bounds__(SCO_BLOCK) void qp_admm_kernel(AdmmArgs a) {
  const QpDev &d = a.d;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (d.active && !d.active[b]) return;
  const int n = d.n, m = d.m, nnzA = d.nnzA, n_e = d.n_e, n_c = d.n_c, ncpl = d.ncpl;
  const int it0 = a.slice > 0 ? d.prog[b] : 0;

  extern __shared__ double lds[];
  AdmmLds s;
  s.As = lds;            s.qs = s.As + nnzA;   s.ls = s.qs + n;     s.us = s.ls + m;
  s.rho = s.us + m;      s.x = s.rho + m;      s.z = s.x + n;       s.y = s.z + m;
  s.t = s.y + m;         s.xt = s.t + m;       s.ge = s.xt + n;     s.r = s.ge + n_e;
  s.xc = s.r + n_c;      s.kinv = s.xc + n_c;  s.cpl = s.kinv + n_e; s.sdy = s.cpl + ncpl;
  s.sdx = s.sdy + m;     s.red = s.sdx + n;
  s.w = (int *)(s.red + NWAVE * 8);

  {
    const double *gAs = d.As + (size_t)b * nnzA;
    for (int t = tid; t < nnzA; t += SCO_BLOCK) s.As[t] = gAs[t];
    // a parked solve (time slicing / adaptive rho) comes back from its scaled x, z, y; t = w (rho z - y) is
    // rebuilt with the rho in force now
    for (int j = tid; j < n; j += SCO_BLOCK) {
      s.qs[j] = d.qs[(size_t)b * n + j]; s.x[j] = it0 > 0 ? d.sx[(size_t)b * n + j] : 0.0; s.sdx[j] = 0.0;
    }
    for (int i = tid; i < m; i += SCO_BLOCK) {
      s.ls[i] = d.ls[(size_t)b * m + i]; s.us[i] = d.us[(size_t)b * m + i];
      s.rho[i] = d.rho[(size_t)b * m + i]; s.w[i] = d.w[(size_t)b * m + i];
      const double z0 = it0 > 0 ? d.sz[(size_t)b * m + i] : 0.0, y0 = it0 > 0 ? d.sy[(size_t)b * m + i] : 0.0;
      s.z[i] = z0; s.y[i] = y0; s.t[i] = (double)s.w[i] * (s.rho[i] * z0 - y0); s.sdy[i] = 0.0;
    }
    for (int e = tid; e < n_e; e += SCO_BLOCK) s.kinv[e] = d.kee_inv[(size_t)b * n_e + e];
    for (int k = tid; k < ncpl; k += SCO_BLOCK) s.cpl[k] = d.cpl[(size_t)b * ncpl + k];
  }
  const double *W = d.W + (size_t)b * n_c * n_c;
  const double cscale = d.cscale[b];
  const double alpha = a.alpha, sigma = a.sigma;
  if (a.adaptive && tid == 0) { d.smask[b] = 0; d.rflag[b] = 0; }
  __syncthreads();

  int status = 0, iter = 0;
  double pri = 0.0, dua = 0.0;
  for (iter = it0 + 1; iter <= a.max_iter; iter++) {
    const bool chk = (a.check > 0 && iter % a.check == 0) || iter == a.max_iter;
    // (1) right-hand side  sigma x - q + A' t,  t = w (rho z - y); eliminated part scaled by K_ee^-1
    for (int j = tid; j < n; j += SCO_BLOCK) {
      double v = 0.0;
      for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) v += s.As[t] * s.t[d.Ai[t]];
      v += sigma * s.x[j] - s.qs[j];
      s.xt[j] = v;
      const int e = d.elim_of[j];
      if (e >= 0) s.ge[e] = v * s.kinv[e];
    }
    __syncthreads();
    // (2) core right-hand side  r = rhs_C - K_CE K_EE^-1 rhs_E
    for (int c = tid; c < n_c; c += SCO_BLOCK) {
      double v = s.xt[d.core_var[c]];
      for (int t = d.a_ptr[c]; t < d.a_ptr[c + 1]; t++) { const int k = d.a_pair[t]; v -= s.cpl[k] * s.ge[d.pair_elim[k]]; }
      s.r[c] = v;
    }
    __syncthreads();
    // (3) x~_C = W r   (W symmetric: thread c walks column c = row c, coalesced across lanes)
    for (int c = tid; c < n_c; c += SCO_BLOCK) {
      double v = 0.0;
      for (int k = 0; k < n_c; k++) v += W[(size_t)k * n_c + c] * s.r[k];
      s.xc[c] = v;
    }
    __syncthreads();
    // (4) back-substitute the eliminated variables, scatter x~ into xt
    for (int j = tid; j < n; j += SCO_BLOCK) {
      const int e = d.elim_of[j];
      if (e >= 0) {
        double v = 0.0;
        for (int k = d.e_ptr[e]; k < d.e_ptr[e + 1]; k++) v += s.cpl[k] * s.xc[d.pair_core[k]];
        s.xt[j] = s.ge[e] - s.kinv[e] * v;
      } else {
        s.xt[j] = s.xc[d.core_of[j]];
      }
    }
    __syncthreads();
    // (5) z~ = A x~, then the z / y / x updates and next iteration's t
    for (int i = tid; i < m; i += SCO_BLOCK) {
      double zt = 0.0;
      for (int t = d.Rp[i]; t < d.Rp[i + 1]; t++) zt += s.As[d.Rpos[t]] * s.xt[d.Rj[t]];
      const double rho = s.rho[i], rinv = 1.0 / rho;
      const double zr = alpha * zt + (1.0 - alpha) * s.z[i];
      double zn = zr + rinv * s.y[i];
      zn = fmin(fmax(zn, s.ls[i]), s.us[i]);
      const double dy = rho * (zr - zn);
      const double yn = s.y[i] + dy;
      s.z[i] = zn; s.y[i] = yn;
      s.t[i] = (double)s.w[i] * (rho * zn - yn);
      if (chk) s.sdy[i] = dy;
    }
    for (int j = tid; j < n; j += SCO_BLOCK) {
      const double xo = s.x[j];
      const double xn = alpha * s.xt[j] + (1.0 - alpha) * xo;
      if (chk) s.sdx[j] = xn - xo;
      s.x[j] = xn;
    }
    __syncthreads();
    if (chk) {
      status = admm_check(a, s, b, 0, cscale, &pri, &dua);
      if (status) break;
    }
    double rho_new = 0.0;
    if (a.adaptive && iter % a.ad_interval == 0 && iter < a.max_iter) {
      const double rho = d.rho_b[b], est = admm_rho_estimate(a, s, b, rho);
      if (est > rho * a.ad_tol || est < rho / a.ad_tol) rho_new = est;
    }
    if (rho_new > 0.0 || (a.slice > 0 && iter == it0 + a.slice && iter < a.max_iter)) {
      // rho must change (setup refactors, then the solve resumes), or the slice is used up (it ends on a
      // termination check): park the solve
      if (rho_new > 0.0 && tid == 0) { d.rho_b[b] = rho_new; d.rflag[b] = 1; d.smask[b] = 1; d.nupd[b] += 1; }
      for (int j = tid; j < n; j += SCO_BLOCK) d.sx[(size_t)b * n + j] = s.x[j];
      for (int i = tid; i < m; i += SCO_BLOCK) { d.sz[(size_t)b * m + i] = s.z[i]; d.sy[(size_t)b * m + i] = s.y[i]; }
      if (tid == 0) { d.prog[b] = iter; d.status[b] = 0; d.iters[b] = iter; }
      return;
    }
  }
  if (a.slice > 0 && tid == 0) d.prog[b] = 0;
  if (!status) {
    iter = a.max_iter;
    status = admm_check(a, s, b, 1, cscale, &pri, &dua);
    if (!status) status = SCO_QP_MAX_ITER_REACHED;
  }
  // ---- unscale and store -----------------------------------------------------
  {
    const double *Dg = d.D + (size_t)b * n, *Eg = d.E + (size_t)b * m;
    const double cinv = 1.0 / cscale;
    for (int j = tid; j < n; j += SCO_BLOCK) d.x[(size_t)b * n + j] = Dg[j] * s.x[j];
    for (int i = tid; i < m; i += SCO_BLOCK) d.y[(size_t)b * m + i] = cinv * Eg[i] * s.y[i] * (double)s.w[i];
    if (tid == 0) {
      d.status[b] = status; d.iters[b] = iter;
      d.resid[2 * (size_t)b] = pri; d.resid[2 * (size_t)b + 1] = dua;
    }
  }
}

// --------------------------------------------------------------------------
// adaptive rho (opt-in).  The ADMM kernels estimate rho themselves (every ad_interval iterations, after the
// termination test) and park the solve when it must change; this kernel only gives the problems that START a QP
// their initial rho and raises their setup flag.
// mode 0: problems flagged in `newqp` start a QP; 1: every active problem does.
// --------------------------------------------------------------------------
struct RhoInitArgs {
  QpDev d;
  const int *newqp;
  int mode;
  double rho0;
};

__global__ void qp_rho_init_kernel(RhoInitArgs a) {
  const QpDev &d = a.d;
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= d.batch) return;
  if (d.active && !d.active[b]) { d.smask[b] = 0; d.rflag[b] = 0; return; }
  if (a.mode == 1 || a.newqp[b]) { d.rho_b[b] = a.rho0; d.smask[b] = 1; d.rflag[b] = 0; d.nupd[b] = 0; }
}

// --------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------
template <typename T>
static int dev_upload(sco_qp *qp, const std::vector<T> &v, const T **out) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  SCO_HIP(hipMalloc(&p, bytes));
  qp->allocs.push_back(p);
  if (!v.empty()) SCO_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)p;
  return SCO_OK;
}
template <typename T>
static int dev_alloc(sco_qp *qp, size_t count, T **out) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  SCO_HIP(hipMalloc(&p, bytes));
  SCO_HIP(hipMemset(p, 0, bytes));
  qp->allocs.push_back(p);
  *out = (T *)p;
  return SCO_OK;
}

static size_t setup_lds_doubles(const QpPlan &pl) {
  return (size_t)pl.nnzP + pl.nnzA + 2 * (size_t)pl.n + pl.m + pl.n + pl.m + pl.ncpl + NWAVE * 2 +
         (size_t)pl.n_c * (pl.n_c + 1) / 2;
}

static int qp_create_impl(sco_qp *qp, int device, int batch, int n, int m, const int *Pp, const int *Pi,
                          const int *Ap, const int *Ai, hipStream_t stream);

int sco_qp_create_on_stream(int device, int batch, int n, int m, const int *Pp, const int *Pi,
                            const int *Ap, const int *Ai, hipStream_t stream, sco_qp **out) {
  if (!out || batch <= 0 || n <= 0 || m < 0 || !Pp || !Ap) { sco_set_error("sco_qp_create: bad argument"); return SCO_ERR_ARG; }
  if ((Pp[n] > 0 && !Pi) || (Ap[n] > 0 && !Ai)) { sco_set_error("sco_qp_create: null row-index array for a non-empty pattern"); return SCO_ERR_ARG; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    sco_set_error("sco_qp_create: no HIP device visible (this library has no CPU fallback)");
    return SCO_ERR_NO_GPU;
  }
  if (device < 0 || device >= ndev) { sco_set_error("sco_qp_create: bad device index"); return SCO_ERR_ARG; }
  SCO_ON_DEVICE(device);
  sco_qp *qp = new sco_qp();
  qp->device = device;
  const int rc = qp_create_impl(qp, device, batch, n, m, Pp, Pi, Ap, Ai, stream);
  if (rc) { sco_qp_destroy(qp); return rc; }      // frees whatever had been allocated
  *out = qp;
  return SCO_OK;
}

static int qp_create_impl(sco_qp *qp, int device, int batch, int n, int m, const int *Pp, const int *Pi,
                          const int *Ap, const int *Ai, hipStream_t stream) {
  { const char *fc = getenv("SCO_QP_FACTOR_CHOLESKY"); qp->factor_cholesky = fc && fc[0] == '1'; }
  const char *no_elim = getenv("SCO_QP_NO_ELIM");
  int rc = qp_plan_build(n, m, Pp, Pi, Ap, Ai, (no_elim && no_elim[0] == '1') ? 0 : 1, qp->plan);
  if (rc != 0) { sco_set_error("sco_qp_create: malformed sparsity pattern"); return SCO_ERR_ARG; }
  const QpPlan &pl = qp->plan;
  (void)hipDeviceGetAttribute(&qp->cus, hipDeviceAttributeMultiprocessorCount, device);
  qp->lds_setup = setup_lds_doubles(pl) * sizeof(double);
  qp->lds_admm = admm_lds_doubles(n, m, pl.nnzA, pl.n_e, pl.n_c, pl.ncpl) * sizeof(double) + (size_t)m * sizeof(int);
  const size_t lds_cap = 160 * 1024;
  const char *force_big = getenv("SCO_QP_FORCE_BIG");
  if (qp->lds_setup > lds_cap || qp->lds_admm > lds_cap || pl.n_c > SCO_BLOCK || (force_big && force_big[0] == '1')) {
    // working set beyond a CU's LDS: the global-memory tier (sco_qp_big.hip)
    bool ok_big = pl.n_c <= 1024 && big_plan_build(pl, qp->big);
    if (!ok_big && !(no_elim && no_elim[0] == '1')) {
      // the global-memory kernels keep an eliminated variable and its (at most two) rows in one thread:
      // send variables with more rows to the core and analyse again
      rc = qp_plan_build(n, m, Pp, Pi, Ap, Ai, 2, qp->plan);
      if (rc != 0) { sco_set_error("sco_qp_create: malformed sparsity pattern"); return SCO_ERR_ARG; }
      qp->big = BigHost();
      ok_big = pl.n_c <= 1024 && big_plan_build(pl, qp->big);
    }
    if (!ok_big) {
      char buf[256];
      snprintf(buf, sizeof buf, "sco_qp_create: pattern not supported (setup %zu B, admm %zu B of LDS, core %d)",
               qp->lds_setup, qp->lds_admm, pl.n_c);
      sco_set_error(buf); return SCO_ERR_CAPACITY;
    }
    qp->use_big = true;
    const char *no_bt = getenv("SCO_QP_NO_BT");
    qp->use_bt = !(no_bt && no_bt[0] == '1') && bt_plan_build(pl, qp->big, qp->bt);
  }
  if (stream) { qp->stream = stream; qp->own_stream = false; }
  else { SCO_HIP(hipStreamCreate(&qp->stream)); qp->own_stream = true; }
  for (auto &e : qp->ev) SCO_HIP(hipEventCreate(&e));

  QpDev &d = qp->d;
  d.n = n; d.m = m; d.nnzP = pl.nnzP; d.nnzA = pl.nnzA; d.n_e = pl.n_e; d.n_c = pl.n_c;
  d.ncpl = pl.ncpl; d.nS = pl.nS; d.batch = batch; d.active = nullptr;
#define UP(field, vec) { int r_ = dev_upload<int>(qp, pl.vec, &d.field); if (r_) return r_; }
  UP(Ap, Ap) UP(Ai, Ai) UP(Rp, Rp) UP(Rj, Rj) UP(Rpos, Rpos) UP(Fp, Fp) UP(Fi, Fi) UP(Fpos, Fpos) UP(Pdiag, Pdiag)
  UP(elim_var, elim_var) UP(core_var, core_var) UP(elim_of, elim_of) UP(core_of, core_of)
  UP(e_ptr, e_ptr) UP(pair_core, pair_core) UP(pair_elim, pair_elim) UP(cp_ptr, cp_ptr) UP(cp_row, cp_row)
  UP(cp_pa, cp_pa) UP(cp_pe, cp_pe) UP(a_ptr, a_ptr) UP(a_pair, a_pair)
  UP(s_a, s_a) UP(s_b, s_b) UP(s_ppos, s_ppos) UP(sa_ptr, sa_ptr) UP(sa_row, sa_row) UP(sa_pa, sa_pa) UP(sa_pb, sa_pb)
  UP(ss_ptr, ss_ptr) UP(ss_k1, ss_k1) UP(ss_k2, ss_k2) UP(ss_e, ss_e)
#undef UP
  { int r_ = dev_upload<int>(qp, pl.Pp, &qp->Pp_dev); if (r_) return r_; }
  { int r_ = dev_upload<int>(qp, pl.Pi, &qp->Pi_dev); if (r_) return r_; }
  const size_t B = (size_t)batch;
#define AL(field, count) { int r_ = dev_alloc(qp, (count), &d.field); if (r_) return r_; }
  AL(Pval, B * pl.nnzP) AL(q, B * n) AL(Aval, B * pl.nnzA) AL(l, B * m) AL(u, B * m) AL(w, B * m)
  AL(Ps, B * pl.nnzP) AL(As, B * pl.nnzA) AL(qs, B * n) AL(ls, B * m) AL(us, B * m) AL(D, B * n) AL(E, B * m)
  AL(cscale, B) AL(rho, B * m) AL(kee_inv, B * pl.n_e) AL(cpl, B * pl.ncpl) AL(W, (qp->use_bt ? 1 : B) * pl.n_c * pl.n_c)
  AL(x, B * n) AL(y, B * m) AL(resid, B * 2) AL(status, B) AL(iters, B)
  AL(prog, B) AL(sx, B * n) AL(sz, B * m) AL(sy, B * m) AL(st, B * m) AL(sg, B * pl.n_e)
  AL(rho_b, B) AL(rflag, B) AL(smask, B) AL(nupd, B) AL(amask, B)
#undef AL
  SCO_HIP(hipFuncSetAttribute((const void *)qp_setup_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
  SCO_HIP(hipFuncSetAttribute((const void *)qp_factor_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
  SCO_HIP(hipFuncSetAttribute((const void *)qp_admm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_cap));
  if (qp->use_big) {
    int r_ = big_upload(qp->big, qp->use_bt ? 1 : batch, qp->allocs, qp->bigd);   // dense workspace unused by bt
    if (r_) return r_;
    if (qp->use_bt && (r_ = bt_upload(qp->bt, batch, qp->allocs, qp->btd))) return r_;
    return SCO_OK;
  }
  {
    const char *no_rl = getenv("SCO_QP_NO_RL");
    if (!(no_rl && no_rl[0] == '1') && rl_plan_build(pl, qp->rl)) {
      int r_ = rl_upload(qp->rl, qp->allocs, qp->rld);
      if (r_) return r_;
      qp->use_rl = true;
    }
  }
  {
    // wavefront tier (one wavefront per problem, block-tridiagonal core solve): patterns of trajectory penalty QPs; it
    // needs the row-local tier beside it for problems whose values leave the penalty-QP structure, and for the opt-in
    // extensions (warm start, adaptive rho), which stay on the row-local kernel
    const char *no_wv = getenv("SCO_QP_NO_WV");
    const bool want_wv = !(no_wv && no_wv[0] == '1');
    if (qp->use_rl && !qp->factor_cholesky && want_wv && wv_plan_build(pl, qp->wv)) {
      int r_ = wv_upload(qp->wv, batch, n, m, qp->allocs, qp->wvd);
      if (r_) return r_;
      qp->use_wv = true;
    }
  }
  {
    const char *no_reg = getenv("SCO_QP_NO_REG");
    int CW = 0, RW = 0, PX = 0;
    if (!(no_reg && no_reg[0] == '1') && reg_caps_for(pl, &CW, &RW, &PX) && reg_plan_build(pl, CW, RW, PX, qp->reg)) {
      int r_ = reg_upload(qp->reg, qp->allocs, qp->regd);
      if (r_) return r_;
      qp->use_reg = true;
    }
  }
  {
    const char *no_fast = getenv("SCO_QP_NO_FAST");
    if (!(no_fast && no_fast[0] == '1') && fast_plan_build(pl, qp->fast)) {
      int r_ = fast_upload(qp->fast, qp->allocs, qp->fastd);
      if (r_) return r_;
      qp->use_fast = true;
    }
  }
  return SCO_OK;
}

extern "C" int sco_qp_create(int device, int batch, int n, int m, const int *Pp, const int *Pi,
                             const int *Ap, const int *Ai, sco_qp **out) {
  return sco_qp_create_on_stream(device, batch, n, m, Pp, Pi, Ap, Ai, nullptr, out);
}

extern "C" int sco_qp_destroy(sco_qp *qp) {
  if (!qp) return SCO_OK;
  ScoDeviceGuard sco_guard_(qp->device);
  if (qp->stream) (void)hipStreamSynchronize(qp->stream);
  for (void *p : qp->allocs) (void)hipFree(p);
  for (auto &e : qp->ev) if (e) (void)hipEventDestroy(e);
  if (qp->own_stream && qp->stream) (void)hipStreamDestroy(qp->stream);
  delete qp;
  return SCO_OK;
}

extern "C" int sco_qp_load(sco_qp *qp, const double *P_val, const double *q, const double *A_val,
                           const double *l, const double *u, const int *row_weight) {
  if (!qp || !q || (qp->d.nnzP && !P_val) || (qp->d.nnzA && !A_val) || (qp->d.m && (!l || !u))) {
    sco_set_error("sco_qp_load: null pointer"); return SCO_ERR_ARG;
  }
  SCO_ON_DEVICE(qp->device);
  const QpDev &d = qp->d; const size_t B = d.batch;
  if (d.nnzP) SCO_HIP(hipMemcpyAsync(d.Pval, P_val, B * d.nnzP * sizeof(double), hipMemcpyHostToDevice, qp->stream));
  SCO_HIP(hipMemcpyAsync(d.q, q, B * d.n * sizeof(double), hipMemcpyHostToDevice, qp->stream));
  if (d.nnzA) SCO_HIP(hipMemcpyAsync(d.Aval, A_val, B * d.nnzA * sizeof(double), hipMemcpyHostToDevice, qp->stream));
  if (d.m) {
    SCO_HIP(hipMemcpyAsync(d.l, l, B * d.m * sizeof(double), hipMemcpyHostToDevice, qp->stream));
    SCO_HIP(hipMemcpyAsync(d.u, u, B * d.m * sizeof(double), hipMemcpyHostToDevice, qp->stream));
    if (row_weight) SCO_HIP(hipMemcpyAsync(d.w, row_weight, B * d.m * sizeof(int), hipMemcpyHostToDevice, qp->stream));
    else {
      std::vector<int> ones(B * d.m, 1);
      SCO_HIP(hipMemcpyAsync(d.w, ones.data(), ones.size() * sizeof(int), hipMemcpyHostToDevice, qp->stream));
      SCO_HIP(hipStreamSynchronize(qp->stream));
    }
  }
  SCO_HIP(hipStreamSynchronize(qp->stream));
  qp->loaded = true;
  return SCO_OK;
}

extern "C" int sco_qp_set_bounds(sco_qp *qp, const double *l, const double *u) {
  if (!qp || !l || !u) { sco_set_error("sco_qp_set_bounds: null pointer"); return SCO_ERR_ARG; }
  if (!qp->loaded) { sco_set_error("sco_qp_set_bounds: call sco_qp_load first"); return SCO_ERR_STATE; }
  SCO_ON_DEVICE(qp->device);
  const QpDev &d = qp->d; const size_t B = d.batch;
  SCO_HIP(hipMemcpyAsync(d.l, l, B * d.m * sizeof(double), hipMemcpyHostToDevice, qp->stream));
  SCO_HIP(hipMemcpyAsync(d.u, u, B * d.m * sizeof(double), hipMemcpyHostToDevice, qp->stream));
  SCO_HIP(hipStreamSynchronize(qp->stream));
  return SCO_OK;
}

int sco_qp_launch(sco_qp *qp, const sco_qp_settings *st, const int *active_dev, hipEvent_t mid) {
  return sco_qp_launch_sliced(qp, st, active_dev, active_dev, 0, mid, nullptr);
}

// fewest problems of a launch for which the wavefront tier is the faster one (SCO_WV_MIN_PER_CU: problems per CU, default 3.0)
int sco_wv_min_live(int cus) {
  const char *e = getenv("SCO_WV_MIN_PER_CU");
  const double per = e ? atof(e) : 3.0;
  return (int)(per * (cus > 0 ? cus : 256));
}

int sco_qp_adaptive_interval(const sco_qp_settings *st) {
  if (st->adaptive_rho_interval > 0) {
    // park points sit on termination checks
    if (st->check_termination > 0)
      return std::max(1, (st->adaptive_rho_interval + st->check_termination / 2) / st->check_termination) * st->check_termination;
    return st->adaptive_rho_interval;
  }
  return st->check_termination > 0 ? 4 * st->check_termination : 100;
}

int sco_qp_wv_iters(const sco_qp *qp, unsigned long long *out) {
  *out = 0;
  if (!qp->use_wv) return SCO_OK;
  SCO_HIP(hipMemcpy(out, qp->wvd.it_count, sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SCO_OK;
}
void sco_qp_wv_iters_reset(sco_qp *qp, hipStream_t st) { if (qp->use_wv) (void)hipMemsetAsync(qp->wvd.it_count, 0, sizeof(unsigned long long), st); }
bool sco_qp_has_wv(const sco_qp *qp, const sco_qp_settings *st) { return qp->use_wv && !st->warm_start && !st->adaptive_rho; }
bool sco_qp_can_adapt(const sco_qp *qp) { return !(qp->use_big && !qp->use_bt); }

// Launch windows and index lists (QpGroup) exist for the paths the bench workloads take: row-local ADMM kernel with
// the Gauss-Jordan inversion, and the structured global-memory tier; fixed rho.
bool sco_qp_supports_groups(const sco_qp *qp, const sco_qp_settings *st) {
  if (st->adaptive_rho) return false;
  if (qp->use_big) return qp->use_bt;                      // structured global-memory tier
  return qp->use_rl && !qp->factor_cholesky;
}

int sco_qp_launch_sliced(sco_qp *qp, const sco_qp_settings *st, const int *setup_mask, const int *active_dev,
                         int slice, hipEvent_t mid, int *sliced, const QpGroup *grp) {
  const bool adaptive = st->adaptive_rho != 0;
  if (grp && (!sco_qp_supports_groups(qp, st) || grp->b0 < 0 || grp->nb <= 0 || grp->b0 + grp->nb > qp->d.batch)) {
    sco_set_error("sco_qp_launch_sliced: launch window not supported by this handle's tier"); return SCO_ERR_STATE;
  }
  hipStream_t stream = grp ? grp->stream : qp->stream;
  if (adaptive) {
    if (qp->use_big && !qp->use_bt) {
      sco_set_error("adaptive_rho: the dense form of the global-memory tier cannot park a solve");
      return SCO_ERR_CAPACITY;
    }
    if (!(st->adaptive_rho_tolerance > 1.0) || st->check_termination <= 0) {
      sco_set_error("adaptive_rho needs adaptive_rho_tolerance > 1 and check_termination > 0"); return SCO_ERR_ARG;
    }
  }
  // kernel of the on-chip tiers (every one of them parks and resumes a solve; the register-offset kernel since r03)
  enum { K_RL, K_REG, K_FAST, K_GENERIC };
  const int kern = qp->use_rl ? K_RL : qp->use_reg ? K_REG : qp->use_fast ? K_FAST : K_GENERIC;
  const bool can_park = qp->use_big ? qp->use_bt : true;
  if (!can_park || st->check_termination <= 0 || (slice <= 0 && !adaptive)) {
    slice = 0;                                  // one launch per solve (always so on the dense global-memory form)
  } else {
    // slices end on a termination check; adaptive rho without time slicing parks only when rho changes
    if (slice <= 0) slice = st->max_iter;
    slice = std::max(1, (slice + st->check_termination - 1) / st->check_termination) * st->check_termination;
  }
  if (sliced) *sliced = slice;
  QpDev d = qp->d; d.active = active_dev;
  QpDev dsetup = qp->d; dsetup.active = adaptive ? qp->d.smask : setup_mask;
  if (grp) { d.b0 = dsetup.b0 = grp->b0; d.nb = dsetup.nb = grp->nb; d.list = dsetup.list = grp->list; }
  const int nwg = grp ? grp->nb : d.batch;          // workgroups of the per-problem kernels
  SetupArgs sa{dsetup, qp->Pp_dev, qp->Pi_dev, st->rho, st->sigma, st->scaling, adaptive ? 1 : 0};
  AdmmArgs aa{d, st->rho, st->sigma, st->alpha, st->eps_abs, st->eps_rel, st->eps_prim_inf, st->eps_dual_inf,
              st->max_iter, st->check_termination, (st->warm_start && qp->solved_once) ? 1 : 0, slice,
              adaptive ? 1 : 0, adaptive ? sco_qp_adaptive_interval(st) : 0, st->adaptive_rho_tolerance, nullptr};
  // The wavefront tier runs cold-start, fixed-rho solves (parity mode); the opt-in extensions keep the row-local kernel.
  // It puts four problems on a CU at ~3.1 us per iteration each where the row-local kernel runs one at ~0.95 us: it is
  // the faster way through a launch only with more than ~3 problems per CU to run (profiles/r04_wv.txt).  The SQP loop
  // says per round which one it wants (QpGroup::tier, from its live count); a plain sco_qp_solve goes by the batch.
  const bool wv_can = qp->use_wv && !aa.warm && !adaptive && !st->warm_start;
  const int wv_min = sco_wv_min_live(qp->cus);
  const bool wv_now = wv_can && (grp && grp->tier ? grp->tier == 2 : (grp ? grp->nb : d.batch) >= wv_min);
  qp->solved_once = true;
  if (!grp) SCO_HIP(hipEventRecord(qp->ev[0], stream));
  if (adaptive && setup_mask != SCO_MASK_NONE) {
    // the problems that start a QP get the initial rho and their setup flag (the parked ones keep what the ADMM
    // kernel left when it parked them)
    RhoInitArgs ra{d, setup_mask == SCO_MASK_ALL ? nullptr : setup_mask, setup_mask == SCO_MASK_ALL ? 1 : 0, st->rho};
    hipLaunchKernelGGL(qp_rho_init_kernel, dim3((d.batch + 255) / 256), dim3(256), 0, stream, ra);
    SCO_HIP(hipGetLastError());
  }
  if (qp->use_big) {
    int r_ = big_launch(aa, dsetup.active, st->scaling, qp->Pp_dev, qp->Pi_dev, qp->big, qp->bigd,
                        qp->use_bt ? &qp->bt : nullptr, qp->use_bt ? &qp->btd : nullptr, stream, grp ? nullptr : qp->ev[1], mid);
    if (r_) return r_;
    if (!grp) SCO_HIP(hipEventRecord(qp->ev[2], stream));
    return SCO_OK;
  }
  {
    // scaling + reduced matrix (256 threads, S into the W buffer), then factor + inverse (1024 threads)
    const size_t ntri = (size_t)d.n_c * (d.n_c + 1) / 2;
    hipLaunchKernelGGL(qp_setup_kernel, dim3(nwg), dim3(SCO_BLOCK), qp->lds_setup - ntri * sizeof(double), stream, sa);
    SCO_HIP(hipGetLastError());
    if (wv_now) {
      // twisted block factorisation + the value test; the dense inverse below then runs for the problems that failed it only
      int r_ = wv_launch_factor(aa, dsetup.active, qp->wv, qp->wvd, stream);
      if (r_) return r_;
      dsetup.active = qp->wvd.rl_need;
    } else if (qp->use_wv) {
      // a row-local launch on a handle that also runs the wavefront tier: besides the problems that start a QP, the dense
      // inverse is formed for the active ones whose QP was factored by that tier (their W buffer still holds S)
      int r_ = wv_launch_need(aa, dsetup.active, qp->wvd, stream);
      if (r_) return r_;
      dsetup.active = qp->wvd.rl_need;
    }
    if (qp->factor_cholesky)
      hipLaunchKernelGGL(qp_factor_kernel, dim3(d.batch), dim3(SCO_FACTOR_BLOCK), (ntri + d.n_c) * sizeof(double), stream, dsetup);
    else {
      const int nb = (d.n_c + 3) / 4, ntiles = nb * (nb + 1) / 2;
      if (ntiles <= SCO_FACTOR_BLOCK) hipLaunchKernelGGL(qp_sweep_kernel<1>, dim3(nwg), dim3(SCO_FACTOR_BLOCK), 0, stream, dsetup);
      else if (ntiles <= 2 * SCO_FACTOR_BLOCK) hipLaunchKernelGGL(qp_sweep_kernel<2>, dim3(nwg), dim3(SCO_FACTOR_BLOCK), 0, stream, dsetup);
      else hipLaunchKernelGGL(qp_sweep_kernel<3>, dim3(nwg), dim3(SCO_FACTOR_BLOCK), 0, stream, dsetup);
    }
    SCO_HIP(hipGetLastError());
  }
  if (!grp) SCO_HIP(hipEventRecord(qp->ev[1], stream));
  if (mid) SCO_HIP(hipEventRecord(mid, stream));
  if (kern == K_RL) {
    if (wv_now) {
      int r_ = wv_launch(aa, qp->wv, qp->wvd, stream);
      if (r_) return r_;
      aa.skip = qp->wvd.ok;                      // the row-local kernel takes what the wavefront tier has left
    }
    int r_ = rl_launch(aa, qp->rl, qp->rld, stream);
    if (r_) return r_;
  } else if (kern == K_REG) {
    int r_ = reg_launch(aa, qp->reg, qp->regd, stream);
    if (r_) return r_;
  } else if (kern == K_FAST) {
    int r_ = fast_launch(aa, qp->fast, qp->fastd, stream);
    if (r_) return r_;
  } else {
    hipLaunchKernelGGL(qp_admm_kernel, dim3(d.batch), dim3(SCO_BLOCK), qp->lds_admm, stream, aa);
    SCO_HIP(hipGetLastError());
  }
  if (!grp) SCO_HIP(hipEventRecord(qp->ev[2], stream));
  return SCO_OK;
}

extern "C" int sco_qp_solve(sco_qp *qp, const sco_qp_settings *settings, double *x, double *y,
                            int *status, int *iters, double *resid) {
  if (!qp || !settings) { sco_set_error("sco_qp_solve: null pointer"); return SCO_ERR_ARG; }
  if (!qp->loaded) { sco_set_error("sco_qp_solve: call sco_qp_load first"); return SCO_ERR_STATE; }
  if (settings->max_iter <= 0 || !(settings->rho > 0) || !(settings->sigma > 0) || settings->scaling < 0) {
    sco_set_error("sco_qp_solve: bad settings"); return SCO_ERR_ARG;
  }
  SCO_ON_DEVICE(qp->device);
  const QpDev &d = qp->d; const size_t B = d.batch;
  int rc;
  if (settings->adaptive_rho) {
    // one launch per rho-update interval; a problem that has not ended stays parked and is resumed by the next one
    SCO_HIP(hipMemsetAsync(d.prog, 0, B * sizeof(int), qp->stream));
    std::vector<int> prog(B);
    const long long cap = (long long)settings->max_iter / sco_qp_adaptive_interval(settings) + 2;   // a launch per rho change at most
    float ms_setup = 0, ms_admm = 0;
    for (long long round = 0; round < cap; round++) {
      // from the second launch on only the parked problems run (a finished one would start over)
      rc = sco_qp_launch_sliced(qp, settings, round == 0 ? SCO_MASK_ALL : SCO_MASK_NONE, round == 0 ? nullptr : d.amask,
                                0, nullptr, nullptr);
      if (rc) return rc;
      SCO_HIP(hipMemcpyAsync(prog.data(), d.prog, B * sizeof(int), hipMemcpyDeviceToHost, qp->stream));
      SCO_HIP(hipStreamSynchronize(qp->stream));
      float t0 = 0, t1 = 0;
      SCO_HIP(hipEventElapsedTime(&t0, qp->ev[0], qp->ev[1]));
      SCO_HIP(hipEventElapsedTime(&t1, qp->ev[1], qp->ev[2]));
      ms_setup += t0; ms_admm += t1;
      bool parked = false;
      for (size_t b = 0; b < B; b++) { prog[b] = prog[b] > 0 ? 1 : 0; parked = parked || prog[b]; }
      if (!parked) break;
      SCO_HIP(hipMemcpyAsync(d.amask, prog.data(), B * sizeof(int), hipMemcpyHostToDevice, qp->stream));
    }
    if (x) SCO_HIP(hipMemcpyAsync(x, d.x, B * d.n * sizeof(double), hipMemcpyDeviceToHost, qp->stream));
    if (y && d.m) SCO_HIP(hipMemcpyAsync(y, d.y, B * d.m * sizeof(double), hipMemcpyDeviceToHost, qp->stream));
    if (status) SCO_HIP(hipMemcpyAsync(status, d.status, B * sizeof(int), hipMemcpyDeviceToHost, qp->stream));
    if (iters) SCO_HIP(hipMemcpyAsync(iters, d.iters, B * sizeof(int), hipMemcpyDeviceToHost, qp->stream));
    if (resid) SCO_HIP(hipMemcpyAsync(resid, d.resid, B * 2 * sizeof(double), hipMemcpyDeviceToHost, qp->stream));
    SCO_HIP(hipStreamSynchronize(qp->stream));
    qp->last_ms[0] = ms_setup; qp->last_ms[1] = ms_admm;
    return SCO_OK;
  }
  rc = sco_qp_launch(qp, settings, nullptr, nullptr);
  if (rc) return rc;
  if (x) SCO_HIP(hipMemcpyAsync(x, d.x, B * d.n * sizeof(double), hipMemcpyDeviceToHost, qp->stream));
  if (y && d.m) SCO_HIP(hipMemcpyAsync(y, d.y, B * d.m * sizeof(double), hipMemcpyDeviceToHost, qp->stream));
  if (status) SCO_HIP(hipMemcpyAsync(status, d.status, B * sizeof(int), hipMemcpyDeviceToHost, qp->stream));
  if (iters) SCO_HIP(hipMemcpyAsync(iters, d.iters, B * sizeof(int), hipMemcpyDeviceToHost, qp->stream));
  if (resid) SCO_HIP(hipMemcpyAsync(resid, d.resid, B * 2 * sizeof(double), hipMemcpyDeviceToHost, qp->stream));
  SCO_HIP(hipStreamSynchronize(qp->stream));
  float ms0 = 0, ms1 = 0;
  SCO_HIP(hipEventElapsedTime(&ms0, qp->ev[0], qp->ev[1]));
  SCO_HIP(hipEventElapsedTime(&ms1, qp->ev[1], qp->ev[2]));
  qp->last_ms[0] = ms0; qp->last_ms[1] = ms1;
  return SCO_OK;
}

extern "C" int sco_qp_adaptive_info(sco_qp *qp, double *rho, int *updates) {
  if (!qp) return SCO_ERR_ARG;
  SCO_ON_DEVICE(qp->device);
  const QpDev &d = qp->d;
  if (rho) SCO_HIP(hipMemcpy(rho, d.rho_b, (size_t)d.batch * sizeof(double), hipMemcpyDeviceToHost));
  if (updates) SCO_HIP(hipMemcpy(updates, d.nupd, (size_t)d.batch * sizeof(int), hipMemcpyDeviceToHost));
  return SCO_OK;
}

extern "C" int sco_qp_info(const sco_qp *qp, int info[4]) {
  if (!qp || !info) return SCO_ERR_ARG;
  info[0] = qp->plan.n_e; info[1] = qp->plan.n_c;
  info[2] = (int)(qp->use_big ? (qp->use_bt ? qp->bt.lds_bytes : 0) : qp->use_rl ? 30208 + qp->rl.lds_bytes : qp->use_reg ? 44032 + qp->reg.lds_bytes : (qp->use_fast ? qp->fast.lds_bytes : qp->lds_admm)); info[3] = qp->plan.ncpl;
  return SCO_OK;
}

extern "C" int sco_qp_last_timing(const sco_qp *qp, double ms[2]) {
  if (!qp || !ms) return SCO_ERR_ARG;
  ms[0] = qp->last_ms[0]; ms[1] = qp->last_ms[1];
  return SCO_OK;
}

// --------------------------------------------------------------------------
// Host-only debug view of the symbolic analysis (no HIP call): lets the CPU
// test-suite check the plans against a NumPy emulation of the kernels.
// sizes[16]: n_e, n_c, ncpl, nS, len(cp_row), len(sa_row), len(ss_k1), nnzFull
// --------------------------------------------------------------------------
static QpPlan g_dbg_plan;
extern "C" int sco_debug_plan_build(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                                    int allow_elim, int *sizes) {
  int rc = qp_plan_build(n, m, Pp, Pi, Ap, Ai, allow_elim, g_dbg_plan);
  if (rc) return SCO_ERR_ARG;
  const QpPlan &p = g_dbg_plan;
  sizes[0] = p.n_e; sizes[1] = p.n_c; sizes[2] = p.ncpl; sizes[3] = p.nS;
  sizes[4] = (int)p.cp_row.size(); sizes[5] = (int)p.sa_row.size(); sizes[6] = (int)p.ss_k1.size();
  sizes[7] = (int)p.Fi.size();
  return SCO_OK;
}
// Host-only: would the pattern last given to sco_debug_plan_build land on the row-local tier, and with what shape?
// info[0] fits, [1] CW (value slots per column = 2 x operand pairs), [2] TR, [3] TC, [4] dynamic LDS bytes, [5] closed plan,
// [6] 1 = aligned closed plan (one barrier per iteration), 2 = 8 column groups of the W tile with the open assignment,
// [7] row slots per thread
extern "C" int sco_debug_rl_plan(int *info) {
  RlHost rh;
  const bool ok = rl_plan_build(g_dbg_plan, rh);
  info[0] = ok ? 1 : 0; info[1] = rh.CW; info[2] = rh.TR; info[3] = rh.TC; info[4] = (int)rh.lds_bytes; info[5] = rh.merged ? 1 : 0; info[6] = rh.aligned ? 1 : rh.lay8 ? 2 : 0; info[7] = rh.NS;
  return SCO_OK;
}
// Host-only: would that pattern land on the wavefront tier?  info[0] fits, [1] block order, [2] blocks, [3] lanes per block,
// [4] instantiation BS, [5] NS, [6] NV, [7] NSTEP, [8] LDS bytes, [9] second single rows
extern "C" int sco_debug_wv_plan(int *info) {
  WvHost wh;
  const bool ok = wv_plan_build(g_dbg_plan, wh);
  info[0] = ok ? 1 : 0; info[1] = wh.bs; info[2] = wh.nb; info[3] = wh.lpb; info[4] = wh.BS; info[5] = wh.NS; info[6] = wh.NV;
  info[7] = wh.NSTEP; info[8] = (int)wh.lds_bytes; info[9] = wh.n_extra;
  return SCO_OK;
}
// Host-only: the structured global-memory plan of that pattern.  info[0] fits, [1] block order, [2] blocks, [3] block normal
// matrices on the matrix cores, [4] why not (BtHost::mf_why), [5] LDS bytes
extern "C" int sco_debug_bt_plan(int *info) {
  BigHost bh; BtHost th;
  const bool ok = g_dbg_plan.n_c <= 1024 && big_plan_build(g_dbg_plan, bh) && bt_plan_build(g_dbg_plan, bh, th);
  info[0] = ok ? 1 : 0; info[1] = th.bs; info[2] = th.nb; info[3] = th.use_mfma ? 1 : 0; info[4] = th.mf_why; info[5] = (int)th.lds_bytes;
  return SCO_OK;
}
// Which ADMM tiers a handle holds: bit 0 row-local, 1 register-offset, 2 sliced-ELL, 3 global memory, 4 its structured form,
// 5 wavefront tier, 6 the structured form builds its block normal matrices on the matrix cores (MFMA)
extern "C" int sco_debug_qp_tiers(const sco_qp *qp) {
  if (!qp) return SCO_ERR_ARG;
  return (qp->use_rl ? 1 : 0) | (qp->use_reg ? 2 : 0) | (qp->use_fast ? 4 : 0) | (qp->use_big ? 8 : 0) | (qp->use_bt ? 16 : 0) | (qp->use_wv ? 32 : 0) |
         ((qp->use_bt && qp->bt.use_mfma) ? 64 : 0);
}
extern "C" int sco_debug_plan_get(const char *name, int *out, int cap) {
  const QpPlan &p = g_dbg_plan;
  const std::vector<int> *v = nullptr;
#define F(x) if (!strcmp(name, #x)) v = &p.x;
  F(Rp) F(Rj) F(Rpos) F(Fp) F(Fi) F(Fpos) F(Pdiag) F(elim_var) F(core_var) F(elim_of) F(core_of)
  F(e_ptr) F(pair_core) F(pair_elim) F(cp_ptr) F(cp_row) F(cp_pa) F(cp_pe) F(a_ptr) F(a_pair)
  F(s_a) F(s_b) F(s_ppos) F(sa_ptr) F(sa_row) F(sa_pa) F(sa_pb) F(ss_ptr) F(ss_k1) F(ss_k2) F(ss_e)
#undef F
  if (!v) return SCO_ERR_ARG;
  if ((int)v->size() > cap) return SCO_ERR_CAPACITY;
  std::copy(v->begin(), v->end(), out);
  return (int)v->size();
}
