// sco_qp_big.hip -- QP layer tier for problems whose working set does not fit a
// CU's LDS (BASELINE config 5: 12-DOF x 50 steps, n = 5600, m = 10 624, dense core
// of order 600).  Same algorithm as the LDS tiers (OSQP's ADMM, the third-party call
// behind /root/reference/sco_py/sco_osqp/osqp_utils.py:195-216, with the two-level
// reduced solve of qp_plan.h and the row-local rewrite of sco_admm_rl.hip); every
// array lives in HBM/L2, one workgroup of 1024 threads per problem.  This tier is
// about coverage, not speed: it is latency/L2-bound (W alone is n_c^2 * 8 = 2.9 MB
// per problem and is streamed every iteration).
#include "sco_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#define BT 1024
#define BWV (BT / 64)

// --------------------------------------------------------------------------
// host plan
// --------------------------------------------------------------------------
bool big_plan_build(const QpPlan &pl, BigHost &bh) {
  const int m = pl.m;
  bh.row_elim.assign(m, -1); bh.row_epos.assign(m, -1);
  std::vector<std::vector<int>> erows(pl.n_e);
  for (int i = 0; i < m; i++)
    for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
      const int e = pl.elim_of[pl.Rj[s]];
      if (e >= 0) {
        if (bh.row_elim[i] >= 0) return false;
        bh.row_elim[i] = e; bh.row_epos[i] = pl.Rpos[s]; erows[e].push_back(i);
      }
    }
  bh.er_ptr.assign(pl.n_e + 1, 0);
  for (int e = 0; e < pl.n_e; e++) {
    if (erows[e].size() > 2) return false;
    for (int i : erows[e]) bh.er_row.push_back(i);
    bh.er_ptr[e + 1] = (int)bh.er_row.size();
  }
  for (int i = 0; i < m; i++) if (bh.row_elim[i] < 0) bh.free_rows.push_back(i);
  bh.pc_ptr.assign(pl.n_c + 1, 0);
  for (int c = 0; c < pl.n_c; c++) {
    const int j = pl.core_var[c];
    for (int p = pl.Fp[j]; p < pl.Fp[j + 1]; p++) {
      const int c2 = pl.core_of[pl.Fi[p]];
      if (c2 < 0) return false;
      bh.pc_pos.push_back(pl.Fpos[p]); bh.pc_core.push_back(c2);
    }
    bh.pc_ptr[c + 1] = (int)bh.pc_pos.size();
  }
  bh.ws_doubles = 3 * (size_t)m + pl.n_e + 2 * (size_t)pl.n_c + pl.n + (size_t)pl.n_c * pl.n_c;
  return true;
}

// --------------------------------------------------------------------------
// device helpers
// --------------------------------------------------------------------------
__device__ __forceinline__ double bwmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double bwsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <int NR, bool IS_MAX>
__device__ __forceinline__ void bblock_reduce(double (&v)[NR], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = IS_MAX ? bwmax(v[k]) : bwsum(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NR; k++) red[wv * NR + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NR; k++) {
    double r = red[k];
    for (int w = 1; w < BWV; w++) r = IS_MAX ? fmax(r, red[w * NR + k]) : r + red[w * NR + k];
    v[k] = r;
  }
}
__device__ __forceinline__ double blimit(double v) {
  v = v < SCO_MIN_SCALING ? 1.0 : v;
  return v > SCO_MAX_SCALING ? SCO_MAX_SCALING : v;
}

struct BigArgs {
  QpDev d;
  const int *Pp, *Pi;
  const int *row_elim, *row_epos, *er_ptr, *er_row, *free_rows, *pc_ptr, *pc_pos, *pc_core;
  int n_free;
  double *ws; size_t ws_stride;
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, check, scaling;
};

// --------------------------------------------------------------------------
// setup: scaling, rho, K_EE^-1, coupling, S, Cholesky, inverse  (global memory)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(BT) void qp_setup_big_kernel(BigArgs a) {
  const QpDev &d = a.d;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (d.active && !d.active[b]) return;
  const int n = d.n, m = d.m, nnzP = d.nnzP, nnzA = d.nnzA, n_e = d.n_e, n_c = d.n_c, ncpl = d.ncpl;
  __shared__ double red[BWV * 2];
  double *Ps = d.Ps + (size_t)b * nnzP, *As = d.As + (size_t)b * nnzA, *qs = d.qs + (size_t)b * n;
  double *D = d.D + (size_t)b * n, *E = d.E + (size_t)b * m;
  double *ws = a.ws + (size_t)b * a.ws_stride;
  double *Dt = ws;                 // n   (scratch during scaling)
  double *Et = ws + n;             // m   (later: rw)
  double *S = ws + (a.ws_stride - (size_t)n_c * n_c);   // n_c x n_c
  const double *Pval = d.Pval + (size_t)b * nnzP, *Aval = d.Aval + (size_t)b * nnzA;
  for (int t = tid; t < nnzP; t += BT) Ps[t] = Pval[t];
  for (int t = tid; t < nnzA; t += BT) As[t] = Aval[t];
  for (int j = tid; j < n; j += BT) { qs[j] = d.q[(size_t)b * n + j]; D[j] = 1.0; }
  for (int i = tid; i < m; i += BT) E[i] = 1.0;
  double c = 1.0;
  __syncthreads();
  for (int it = 0; it < a.scaling; it++) {
    for (int j = tid; j < n; j += BT) {
      double v = 0.0;
      for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) v = fmax(v, fabs(Ps[d.Fpos[t]]));
      for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) v = fmax(v, fabs(As[t]));
      Dt[j] = 1.0 / sqrt(blimit(v));
    }
    for (int i = tid; i < m; i += BT) {
      double v = 0.0;
      for (int t = d.Rp[i]; t < d.Rp[i + 1]; t++) v = fmax(v, fabs(As[d.Rpos[t]]));
      Et[i] = 1.0 / sqrt(blimit(v));
    }
    __syncthreads();
    for (int j = tid; j < n; j += BT) {
      const double dj = Dt[j];
      for (int t = a.Pp[j]; t < a.Pp[j + 1]; t++) Ps[t] = (Ps[t] * Dt[a.Pi[t]]) * dj;
      for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) As[t] = (As[t] * Et[d.Ai[t]]) * dj;
      qs[j] *= dj; D[j] *= dj;
    }
    for (int i = tid; i < m; i += BT) E[i] *= Et[i];
    __syncthreads();
    double s1[1] = {0.0}, s2[1] = {0.0};
    for (int j = tid; j < n; j += BT) {
      double v = 0.0;
      for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) v = fmax(v, fabs(Ps[d.Fpos[t]]));
      s1[0] += v; s2[0] = fmax(s2[0], fabs(qs[j]));
    }
    bblock_reduce<1, false>(s1, red);
    bblock_reduce<1, true>(s2, red);
    double ct = n > 0 ? s1[0] / (double)n : 0.0;
    ct = fmax(ct, blimit(s2[0]));
    ct = 1.0 / blimit(ct);
    for (int t = tid; t < nnzP; t += BT) Ps[t] *= ct;
    for (int j = tid; j < n; j += BT) qs[j] *= ct;
    c *= ct;
    __syncthreads();
  }
  {
    double *ls = d.ls + (size_t)b * m, *us = d.us + (size_t)b * m, *rho = d.rho + (size_t)b * m;
    const double *l = d.l + (size_t)b * m, *u = d.u + (size_t)b * m;
    const int *w = d.w + (size_t)b * m;
    for (int i = tid; i < m; i += BT) {
      double li = fmax(l[i], -SCO_INFTY) * E[i], ui = fmin(u[i], SCO_INFTY) * E[i];
      ls[i] = li; us[i] = ui;
      double r;
      if (li < -SCO_INFTY * SCO_MIN_SCALING && ui > SCO_INFTY * SCO_MIN_SCALING) r = SCO_RHO_MIN;
      else if (ui - li < SCO_RHO_TOL) r = SCO_RHO_EQ_OVER_RHO_INEQ * a.rho;
      else r = a.rho;
      rho[i] = r; Et[i] = r * (double)w[i];
    }
    if (tid == 0) d.cscale[b] = c;
  }
  __syncthreads();
  double *kinv = d.kee_inv + (size_t)b * n_e, *cpl = d.cpl + (size_t)b * ncpl;
  for (int e = tid; e < n_e; e += BT) {
    const int ve = d.elim_var[e];
    double v = a.sigma;
    if (d.Pdiag[ve] >= 0) v += Ps[d.Pdiag[ve]];
    for (int t = d.Ap[ve]; t < d.Ap[ve + 1]; t++) v += Et[d.Ai[t]] * As[t] * As[t];
    kinv[e] = 1.0 / v;
  }
  for (int k = tid; k < ncpl; k += BT) {
    double v = 0.0;
    for (int t = d.cp_ptr[k]; t < d.cp_ptr[k + 1]; t++) v += Et[d.cp_row[t]] * As[d.cp_pa[t]] * As[d.cp_pe[t]];
    cpl[k] = v;
  }
  for (size_t t = tid; t < (size_t)n_c * n_c; t += BT) S[t] = 0.0;
  __syncthreads();
  for (int id = tid; id < d.nS; id += BT) {
    const int sa = d.s_a[id], sb = d.s_b[id];
    double v = (sa == sb) ? a.sigma : 0.0;
    if (d.s_ppos[id] >= 0) v += Ps[d.s_ppos[id]];
    for (int t = d.sa_ptr[id]; t < d.sa_ptr[id + 1]; t++) v += Et[d.sa_row[t]] * As[d.sa_pa[t]] * As[d.sa_pb[t]];
    for (int t = d.ss_ptr[id]; t < d.ss_ptr[id + 1]; t++) v -= cpl[d.ss_k1[t]] * cpl[d.ss_k2[t]] * kinv[d.ss_e[t]];
    S[(size_t)sa * n_c + sb] = v;       // lower triangle (sa >= sb)
  }
  __syncthreads();
  // Cholesky S = L L' on the lower triangle (left-looking; every thread recomputes the pivot)
  for (int j = 0; j < n_c; j++) {
    const double *rj = S + (size_t)j * n_c;
    double piv = rj[j];
    for (int k = 0; k < j; k++) piv -= rj[k] * rj[k];
    piv = sqrt(piv);
    for (int i = j + 1 + tid; i < n_c; i += BT) {
      double *ri = S + (size_t)i * n_c;
      double s = ri[j];
      for (int k = 0; k < j; k++) s -= ri[k] * rj[k];
      ri[j] = s / piv;
    }
    __syncthreads();
    if (tid == 0) S[(size_t)j * n_c + j] = piv;
    __syncthreads();
  }
  // M = L^-1 in place, last column first
  for (int j = n_c - 1; j >= 0; j--) {
    const double ljj = S[(size_t)j * n_c + j];
    double acc[1];   // n_c <= BT: one row per thread
    const int i = j + 1 + tid;
    acc[0] = 0.0;
    if (i < n_c) {
      const double *ri = S + (size_t)i * n_c;
      double s = 0.0;
      for (int k = j + 1; k <= i; k++) s += ri[k] * S[(size_t)k * n_c + j];
      acc[0] = s;
    }
    __syncthreads();
    if (i < n_c) S[(size_t)i * n_c + j] = -acc[0] / ljj;
    if (tid == 0) S[(size_t)j * n_c + j] = 1.0 / ljj;
    __syncthreads();
  }
  // W = M' M
  double *W = d.W + (size_t)b * n_c * n_c;
  for (size_t p = tid; p < (size_t)n_c * n_c; p += BT) {
    const int ia = (int)(p / n_c), ib = (int)(p % n_c);
    if (ib > ia) continue;
    double s = 0.0;
    for (int k = ia; k < n_c; k++) s += S[(size_t)k * n_c + ia] * S[(size_t)k * n_c + ib];
    W[(size_t)ia * n_c + ib] = s; W[(size_t)ib * n_c + ia] = s;
  }
}

// --------------------------------------------------------------------------
// ADMM (row-local formulation, see sco_admm_rl.hip), all state in global memory
// --------------------------------------------------------------------------
__device__ __forceinline__ double big_row_core_dot(const QpDev &d, const double *As, const double *vec, int i) {
  double v = 0.0;
  for (int s = d.Rp[i]; s < d.Rp[i + 1]; s++) {
    const int c = d.core_of[d.Rj[s]];
    if (c >= 0) v += As[d.Rpos[s]] * vec[c];
  }
  return v;
}

__global__ __launch_bounds__(BT) void qp_admm_big_kernel(BigArgs a) {
  const QpDev &d = a.d;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (d.active && !d.active[b]) return;
  const int n = d.n, m = d.m, n_e = d.n_e, n_c = d.n_c;
  __shared__ double red[BWV * 8];
  const double *As = d.As + (size_t)b * d.nnzA, *Ps = d.Ps + (size_t)b * d.nnzP;
  const double *qs = d.qs + (size_t)b * n, *ls = d.ls + (size_t)b * m, *us = d.us + (size_t)b * m;
  const double *rho = d.rho + (size_t)b * m, *kinv = d.kee_inv + (size_t)b * n_e;
  const double *W = d.W + (size_t)b * n_c * n_c, *Dg = d.D + (size_t)b * n, *Eg = d.E + (size_t)b * m;
  const int *w = d.w + (size_t)b * m;
  double *x = d.x + (size_t)b * n, *y = d.y + (size_t)b * m;     // scaled iterates; unscaled at the end
  double *ws = a.ws + (size_t)b * a.ws_stride;
  double *z = ws, *tp = z + m, *sdy = tp + m, *ge = sdy + m, *rv = ge + n_e, *xc = rv + n_c, *sdx = xc + n_c;
  const double cscale = d.cscale[b], alpha = a.alpha, sigma = a.sigma;

  for (int j = tid; j < n; j += BT) { x[j] = 0.0; sdx[j] = 0.0; }
  for (int i = tid; i < m; i += BT) { z[i] = 0.0; y[i] = 0.0; tp[i] = 0.0; sdy[i] = 0.0; }
  __syncthreads();
  for (int e = tid; e < n_e; e += BT) {
    const double g = -qs[d.elim_var[e]] * kinv[e];
    ge[e] = g;
    for (int r = a.er_ptr[e]; r < a.er_ptr[e + 1]; r++) {
      const int i = a.er_row[r];
      tp[i] = -((double)w[i] * rho[i]) * As[a.row_epos[i]] * g;
    }
  }
  __syncthreads();

  int status = 0, iter = 0;
  double pri = 0.0, dua = 0.0;
  for (iter = 1; iter <= a.max_iter; iter++) {
    const bool chk = (a.check > 0 && iter % a.check == 0) || iter == a.max_iter;
    // (1) core right-hand side
    for (int c = tid; c < n_c; c += BT) {
      const int j = d.core_var[c];
      double v = 0.0;
      for (int p = d.Ap[j]; p < d.Ap[j + 1]; p++) v += As[p] * tp[d.Ai[p]];
      rv[c] = (sigma * x[j] - qs[j]) + v;
    }
    __syncthreads();
    // (3) x~_C = W r  (W symmetric: lane c walks column c, coalesced)
    for (int c = tid; c < n_c; c += BT) {
      double v = 0.0;
      for (int k = 0; k < n_c; k++) v += W[(size_t)k * n_c + c] * rv[k];
      xc[c] = v;
    }
    __syncthreads();
    // (Y) eliminated variables with their rows, then the remaining rows, then core x
    for (int e = tid; e < n_e; e += BT) {
      const int r0 = a.er_ptr[e], nr = a.er_ptr[e + 1] - r0, j = d.elim_var[e];
      double zc[2] = {0.0, 0.0}, ae[2] = {0.0, 0.0}, rw[2] = {0.0, 0.0};
      int ri[2] = {-1, -1};
      double acc = 0.0;
      for (int q = 0; q < nr; q++) {
        const int i = a.er_row[r0 + q];
        ri[q] = i; ae[q] = As[a.row_epos[i]]; rw[q] = (double)w[i] * rho[i];
        zc[q] = big_row_core_dot(d, As, xc, i);
        acc += rw[q] * ae[q] * zc[q];
      }
      const double xte = ge[e] - kinv[e] * acc;
      double tq[2] = {0.0, 0.0};
      for (int q = 0; q < nr; q++) {
        const int i = ri[q];
        const double zt = zc[q] + ae[q] * xte;
        const double zr = alpha * zt + (1.0 - alpha) * z[i];
        double zn = zr + (1.0 / rho[i]) * y[i];
        zn = fmin(fmax(zn, ls[i]), us[i]);
        const double dy = rho[i] * (zr - zn);
        y[i] += dy; z[i] = zn; sdy[i] = dy;
        tq[q] = (double)w[i] * (rho[i] * zn - y[i]);
      }
      const double xn = alpha * xte + (1.0 - alpha) * x[j];
      sdx[j] = xn - x[j]; x[j] = xn;
      double rhs_e = sigma * xn - qs[j];
      for (int q = 0; q < nr; q++) rhs_e += ae[q] * tq[q];
      const double g = rhs_e * kinv[e];
      ge[e] = g;
      for (int q = 0; q < nr; q++) tp[ri[q]] = tq[q] - rw[q] * ae[q] * g;
    }
    for (int f = tid; f < a.n_free; f += BT) {
      const int i = a.free_rows[f];
      const double zt = big_row_core_dot(d, As, xc, i);
      const double zr = alpha * zt + (1.0 - alpha) * z[i];
      double zn = zr + (1.0 / rho[i]) * y[i];
      zn = fmin(fmax(zn, ls[i]), us[i]);
      const double dy = rho[i] * (zr - zn);
      y[i] += dy; z[i] = zn; sdy[i] = dy;
      tp[i] = (double)w[i] * (rho[i] * zn - y[i]);
    }
    for (int c = tid; c < n_c; c += BT) {
      const int j = d.core_var[c];
      const double xn = alpha * xc[c] + (1.0 - alpha) * x[j];
      sdx[j] = xn - x[j]; x[j] = xn;
    }
    __syncthreads();
    if (!chk) continue;

    for (int approximate = 0; approximate < 2 && !status; approximate++) {
      if (approximate && iter < a.max_iter) break;
      const double cinv = 1.0 / cscale;
      double ea = a.eps_abs, er = a.eps_rel, epi = a.eps_prim_inf, edi = a.eps_dual_inf;
      if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
      double v[7] = {0, 0, 0, 0, 0, 0, 0};
      for (int i = tid; i < m; i += BT) {
        double ax = 0.0;
        for (int s = d.Rp[i]; s < d.Rp[i + 1]; s++) ax += As[d.Rpos[s]] * x[d.Rj[s]];
        const double ei = 1.0 / Eg[i];
        v[0] = fmax(v[0], fabs(ei * (ax - z[i]))); v[1] = fmax(v[1], fabs(ei * z[i])); v[2] = fmax(v[2], fabs(ei * ax));
      }
      for (int j = tid; j < n; j += BT) {
        double px = 0.0, aty = 0.0;
        for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * x[d.Fi[t]];
        for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) { const int i = d.Ai[t]; aty += As[t] * y[i] * (double)w[i]; }
        const double dj = 1.0 / Dg[j];
        v[3] = fmax(v[3], fabs(dj * (qs[j] + px + aty))); v[4] = fmax(v[4], fabs(dj * qs[j]));
        v[5] = fmax(v[5], fabs(dj * aty)); v[6] = fmax(v[6], fabs(dj * px));
      }
      bblock_reduce<7, true>(v, red);
      pri = v[0]; dua = cinv * v[3];
      if (!(pri <= SCO_INFTY) || !(dua <= SCO_INFTY)) { status = SCO_QP_NON_CVX; break; }
      const double eps_p = ea + er * fmax(v[1], v[2]);
      const double eps_d = ea + er * cinv * fmax(v[4], fmax(v[5], v[6]));
      const bool prim_ok = (m == 0) || (pri < eps_p), dual_ok = dua < eps_d;
      if (prim_ok && dual_ok) { status = approximate ? SCO_QP_SOLVED_INACCURATE : SCO_QP_SOLVED; break; }
      if (!prim_ok) {
        double r1[1] = {0.0};
        for (int i = tid; i < m; i += BT) {
          double dy = sdy[i];
          if (us[i] > SCO_INFTY * SCO_MIN_SCALING) {
            if (ls[i] < -SCO_INFTY * SCO_MIN_SCALING) dy = 0.0; else dy = fmin(dy, 0.0);
          } else if (ls[i] < -SCO_INFTY * SCO_MIN_SCALING) dy = fmax(dy, 0.0);
          sdy[i] = dy;
          r1[0] = fmax(r1[0], fabs(Eg[i] * dy));
        }
        bblock_reduce<1, true>(r1, red);
        const double ndy = r1[0];
        if (ndy > epi) {
          double lhs[1] = {0.0};
          for (int i = tid; i < m; i += BT) lhs[0] += (double)w[i] * (us[i] * fmax(sdy[i], 0.0) + ls[i] * fmin(sdy[i], 0.0));
          bblock_reduce<1, false>(lhs, red);
          if (lhs[0] < -epi * ndy) {
            double nat[1] = {0.0};
            for (int j = tid; j < n; j += BT) {
              double aty = 0.0;
              for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) { const int i = d.Ai[t]; aty += As[t] * sdy[i] * (double)w[i]; }
              nat[0] = fmax(nat[0], fabs(aty / Dg[j]));
            }
            bblock_reduce<1, true>(nat, red);
            if (nat[0] < epi * ndy) { status = approximate ? SCO_QP_PRIMAL_INFEASIBLE_INACCURATE : SCO_QP_PRIMAL_INFEASIBLE; break; }
          }
        }
      }
      if (!dual_ok) {
        double r1[1] = {0.0};
        for (int j = tid; j < n; j += BT) r1[0] = fmax(r1[0], fabs(Dg[j] * sdx[j]));
        bblock_reduce<1, true>(r1, red);
        const double ndx = r1[0];
        if (ndx > edi) {
          double qdx[1] = {0.0};
          for (int j = tid; j < n; j += BT) qdx[0] += qs[j] * sdx[j];
          bblock_reduce<1, false>(qdx, red);
          if (qdx[0] < -cscale * edi * ndx) {
            double npx[1] = {0.0};
            for (int j = tid; j < n; j += BT) {
              double px = 0.0;
              for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * sdx[d.Fi[t]];
              npx[0] = fmax(npx[0], fabs(px / Dg[j]));
            }
            bblock_reduce<1, true>(npx, red);
            if (npx[0] < cscale * edi * ndx) {
              double bad[1] = {0.0};
              for (int i = tid; i < m; i += BT) {
                double adx = 0.0;
                for (int s = d.Rp[i]; s < d.Rp[i + 1]; s++) adx += As[d.Rpos[s]] * sdx[d.Rj[s]];
                adx /= Eg[i];
                if ((us[i] < SCO_INFTY * SCO_MIN_SCALING && adx > edi * ndx) ||
                    (ls[i] > -SCO_INFTY * SCO_MIN_SCALING && adx < -edi * ndx)) bad[0] = 1.0;
              }
              bblock_reduce<1, true>(bad, red);
              if (bad[0] == 0.0) { status = approximate ? SCO_QP_DUAL_INFEASIBLE_INACCURATE : SCO_QP_DUAL_INFEASIBLE; break; }
            }
          }
        }
      }
    }
    __syncthreads();
    if (status) break;
  }
  if (!status) status = SCO_QP_MAX_ITER_REACHED;
  if (iter > a.max_iter) iter = a.max_iter;
  __syncthreads();
  {
    const double cinv = 1.0 / cscale;
    for (int j = tid; j < n; j += BT) x[j] = Dg[j] * x[j];
    for (int i = tid; i < m; i += BT) y[i] = cinv * Eg[i] * y[i] * (double)w[i];
    if (tid == 0) {
      d.status[b] = status; d.iters[b] = iter;
      d.resid[2 * (size_t)b] = pri; d.resid[2 * (size_t)b + 1] = dua;
    }
  }
}

// --------------------------------------------------------------------------
// host glue
// --------------------------------------------------------------------------
template <typename T>
static int upb(std::vector<void *> &allocs, const std::vector<T> &v, const T **out) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  SCO_HIP(hipMalloc(&p, bytes));
  allocs.push_back(p);
  if (!v.empty()) SCO_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)p;
  return SCO_OK;
}

int big_upload(const BigHost &bh, int batch, std::vector<void *> &allocs, BigDev &bd) {
  int rc;
  if ((rc = upb(allocs, bh.row_elim, &bd.row_elim))) return rc;
  if ((rc = upb(allocs, bh.row_epos, &bd.row_epos))) return rc;
  if ((rc = upb(allocs, bh.er_ptr, &bd.er_ptr))) return rc;
  if ((rc = upb(allocs, bh.er_row, &bd.er_row))) return rc;
  if ((rc = upb(allocs, bh.free_rows, &bd.free_rows))) return rc;
  if ((rc = upb(allocs, bh.pc_ptr, &bd.pc_ptr))) return rc;
  if ((rc = upb(allocs, bh.pc_pos, &bd.pc_pos))) return rc;
  if ((rc = upb(allocs, bh.pc_core, &bd.pc_core))) return rc;
  void *p = nullptr;
  SCO_HIP(hipMalloc(&p, (size_t)batch * bh.ws_doubles * sizeof(double)));
  SCO_HIP(hipMemset(p, 0, (size_t)batch * bh.ws_doubles * sizeof(double)));
  allocs.push_back(p);
  bd.ws = (double *)p;
  return SCO_OK;
}

int big_launch(const AdmmArgs &a, int scaling, const int *Pp, const int *Pi, const BigHost &bh, const BigDev &bd,
               hipStream_t st, hipEvent_t ev_mid, hipEvent_t mid2) {
  BigArgs ba;
  ba.d = a.d; ba.Pp = Pp; ba.Pi = Pi;
  ba.row_elim = bd.row_elim; ba.row_epos = bd.row_epos; ba.er_ptr = bd.er_ptr; ba.er_row = bd.er_row;
  ba.free_rows = bd.free_rows; ba.pc_ptr = bd.pc_ptr; ba.pc_pos = bd.pc_pos; ba.pc_core = bd.pc_core;
  ba.n_free = (int)bh.free_rows.size();
  ba.ws = bd.ws; ba.ws_stride = bh.ws_doubles;
  ba.rho = a.rho; ba.sigma = a.sigma; ba.alpha = a.alpha; ba.eps_abs = a.eps_abs; ba.eps_rel = a.eps_rel;
  ba.eps_prim_inf = a.eps_prim_inf; ba.eps_dual_inf = a.eps_dual_inf;
  ba.max_iter = a.max_iter; ba.check = a.check; ba.scaling = scaling;
  hipLaunchKernelGGL(qp_setup_big_kernel, dim3(a.d.batch), dim3(BT), 0, st, ba);
  SCO_HIP(hipGetLastError());
  if (ev_mid) SCO_HIP(hipEventRecord(ev_mid, st));
  if (mid2) SCO_HIP(hipEventRecord(mid2, st));
  hipLaunchKernelGGL(qp_admm_big_kernel, dim3(a.d.batch), dim3(BT), 0, st, ba);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}
