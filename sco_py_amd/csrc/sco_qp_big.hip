// sco_qp_big.hip -- QP layer tier for problems whose working set does not fit a
// CU's LDS (BASELINE config 5: 12-DOF x 50 steps, n = 5600, m = 10 624, core of order
// 600).  Same algorithm as the LDS tiers (OSQP's ADMM, the third-party call behind
// /root/reference/sco_py/sco_osqp/osqp_utils.py:195-216, with the two-level reduced
// solve of qp_plan.h and the row-local rewrite of sco_admm_rl.hip); one workgroup per
// problem, values and row state in HBM/L2.  Two forms:
//   * structured (qp_bt_factor_kernel + qp_admm_bt_kernel, second half of this file):
//     for a banded core -- twisted block LDL' of the block-tridiagonal Schur
//     complement (factors resident in LDS), dense row chunks addressed without index
//     arrays, A't reduced inside the wavefront.  This is what 12 x 50 runs on.
//   * dense (qp_setup_big_kernel's Cholesky + qp_admm_big_kernel): explicit inverse W
//     of the core streamed from L2 every iteration (n_c^2 * 8 = 2.9 MB at 12 x 50);
//     the fall-back for cores that are not banded and the cross-check of the above.
#include <type_traits>

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
template <int NR, bool IS_MAX, int NT = BT>
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
    for (int w = 1; w < NT / 64; w++) r = IS_MAX ? fmax(r, red[w * NR + k]) : r + red[w * NR + k];
    v[k] = r;
  }
}
__device__ __forceinline__ double blimit(double v) {
  v = v < SCO_MIN_SCALING ? 1.0 : v;
  return v > SCO_MAX_SCALING ? SCO_MAX_SCALING : v;
}

// chunk descriptor of the structured form (bt_plan_build): 16 ints per chunk
#define CH_STRIDE 16      // kind, nact, ncols, r0, c0, pos0, col stride, e0, j0, r1, ep0, ep0 stride, ep1, ep1 stride, partial base, pad

struct BigArgs {
  QpDev d;
  const int *Pp, *Pi;
  const int *row_elim, *row_epos, *er_ptr, *er_row, *free_rows, *pc_ptr, *pc_pos, *pc_core;
  int n_free;
  double *ws; size_t ws_stride;
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, check, scaling, warm;
  // block-tridiagonal core (bt_bs > 0): S goes to block storage instead of the dense array
  int bt_bs, bt_nb, bt_mid;
  double *bt_blk; size_t bt_stride;
  const int *ch_desc, *it, *cent;
  int nchunks, npart, use_part;
  int bt_triple;     // 1: three dense chunks in flight per wavefront where they qualify (SCO_QP_BT_TRIPLE=0: pairs only)
  // park / resume of the structured kernel (time slicing, adaptive rho; see RlArgs in sco_admm_rl.hip)
  int slice, adaptive, ad_interval, per_problem_rho;
  double ad_tol;
  double *park_part;
  // per problem and dense chunk: which of the per-row constants (l, u, rho, weight of the primary and of the secondary rows)
  // are known values for every row of the chunk (flags [batch][nchunks]); ccon[batch]: the common weight the flags refer to
  int *cflag; double *ccon;
  // block normal matrices by MFMA (BtHost::use_mfma): null = the vector path
  const int *mf_id, *mf_h0, *mf_h1;
  double *stamp;     // diagnostic build only (SCO_STAMP)
};

// --------------------------------------------------------------------------
// setup: scaling, rho, K_EE^-1, coupling, S, Cholesky, inverse  (global memory)
// --------------------------------------------------------------------------
__global__ __launch_bounds__(BT) void qp_setup_big_kernel(BigArgs a) {
  const QpDev &d = a.d;
  const int b = d.list ? d.list[blockIdx.x + d.b0] : (int)blockIdx.x + d.b0, tid = threadIdx.x;
  if (b < 0 || (d.active && !d.active[b])) return;
  const int n = d.n, m = d.m, nnzP = d.nnzP, nnzA = d.nnzA, n_e = d.n_e, n_c = d.n_c, ncpl = d.ncpl;
  const double rho0 = a.per_problem_rho ? d.rho_b[b] : a.rho;
  __shared__ double red[BWV * 2];
  double *Ps = d.Ps + (size_t)b * nnzP, *As = d.As + (size_t)b * nnzA, *qs = d.qs + (size_t)b * n;
  double *D = d.D + (size_t)b * n, *E = d.E + (size_t)b * m;
  double *ws = a.ws + (size_t)b * a.ws_stride;
  double *Dt = ws;                 // n   (scratch during scaling)
  double *Et = ws + n;             // m   (later: rw)
  double *S = a.bt_bs ? nullptr : ws + (a.ws_stride - (size_t)n_c * n_c);   // n_c x n_c (dense route only)
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
      else if (ui - li < SCO_RHO_TOL) r = SCO_RHO_EQ_OVER_RHO_INEQ * rho0;
      else r = rho0;
      rho[i] = r; Et[i] = r * (double)w[i];
    }
    if (tid == 0) d.cscale[b] = c;
  }
  __syncthreads();
  if (a.bt_bs && a.cflag) {
    // Constants shared by the rows of a dense chunk.  A hinge block's rows all have l = -inf, the base rho and one
    // weight, their slack rows l = 0, u = +inf, the base rho and weight 1: six of the eight per-row constants of such a
    // chunk are known without a load, and the ADMM kernel streams every one it does load per row from L2 / MALL / HBM
    // in every iteration (48 of 232 bytes per row pair).  Flag bits of the primary rows (secondary rows: << 8):
    //   1 l = -inf   2 l = 0   4 u = +inf   8 rho = the problem's base rho   16 weight = 1   32 weight = ccon[b]
    // (ccon[b] = the weight of the first dense chunk's first row: the duplication count of the penalty rows).  A bound
    // beyond the infinity threshold counts as infinite whatever its scaling (the row update only clamps with it).
    const double *ls = d.ls + (size_t)b * m, *us = d.us + (size_t)b * m, *rho = d.rho + (size_t)b * m;
    const int *w = d.w + (size_t)b * m;
    const double BIGV = SCO_INFTY * SCO_MIN_SCALING;
    int wk = 1;
    for (int ch = 0; ch < a.nchunks; ch++)
      if (a.ch_desc[(size_t)ch * CH_STRIDE] == 0) { wk = w[a.ch_desc[(size_t)ch * CH_STRIDE + 3]]; break; }
    if (tid == 0) a.ccon[b] = (double)wk;
    for (int ch = tid; ch < a.nchunks; ch += BT) {
      const int *dsc = a.ch_desc + (size_t)ch * CH_STRIDE;
      int fl = 0;
      if (dsc[0] == 0) {
        const int cnt = dsc[1];
        for (int side = 0; side < 2; side++) {
          const int r0 = side == 0 ? dsc[3] : dsc[9];
          if (r0 < 0) continue;
          bool l_inf = true, l_zero = true, u_inf = true, r_base = true, w_one = true, w_k = true;
          for (int k = 0; k < cnt; k++) {
            const int i = r0 + k;
            l_inf = l_inf && ls[i] < -BIGV; l_zero = l_zero && ls[i] == 0.0; u_inf = u_inf && us[i] > BIGV;
            r_base = r_base && rho[i] == rho0; w_one = w_one && w[i] == 1; w_k = w_k && w[i] == wk;
          }
          const int f = (l_inf ? 1 : 0) | (l_zero ? 2 : 0) | (u_inf ? 4 : 0) | (r_base ? 8 : 0) | (w_one ? 16 : 0) | (w_k ? 32 : 0);
          fl |= f << (8 * side);
        }
      }
      a.cflag[(size_t)b * a.nchunks + ch] = fl;
    }
  }
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
  const int bs = a.bt_bs;
  double *blk = bs ? a.bt_blk + (size_t)b * a.bt_stride : nullptr;
  if (bs) {
    // block storage [t][0] = S_tt (full), [t][1] = S_{t,t-1}; padding rows get a unit diagonal
    const int bb = bs * bs;
    for (size_t t = tid; t < a.bt_stride; t += BT) {
      const int blkid = (int)(t / (2 * bb)), r = (int)(t % (2 * bb));
      const int i = r / bs, j = r % bs;
      blk[t] = (r < bb && i == j && blkid * bs + i >= n_c) ? 1.0 : 0.0;
    }
  } else {
    for (size_t t = tid; t < (size_t)n_c * n_c; t += BT) S[t] = 0.0;
  }
  __syncthreads();
  if (bs && a.mf_id) {
    // ---- N1 (north_star: "MFMA for the dense batched Jacobian x step contraction"): the block normal matrices J' R J
    // on v_mfma_f64_16x16x4.  An entry of a diagonal block is  sigma + P + [terms of rows in front of the block's hinge
    // rows] + [hinge rows: (R J)' J] + [terms behind them] - [Schur terms of the eliminated slacks], summed in exactly
    // this order by the vector path below.  Here the first bracket initialises the accumulator tile, the matrix unit adds
    // the hinge rows in their order (four rows per instruction; scripts/microbench/jtrj_mfma.hip: bit-identical to the
    // vector loop, 4.3 x faster at 100 x 12 blocks), and the loop below continues with the rest.
    //   (a) accumulator start values into the block storage
    for (int id = tid; id < d.nS; id += BT) {
      const int sa = d.s_a[id], sb = d.s_b[id];
      if (sa / bs != sb / bs) continue;
      double v = (sa == sb) ? a.sigma : 0.0;
      if (d.s_ppos[id] >= 0) v += Ps[d.s_ppos[id]];
      for (int t = d.sa_ptr[id]; t < a.mf_h0[id]; t++) v += Et[d.sa_row[t]] * As[d.sa_pa[t]] * As[d.sa_pb[t]];
      blk[(size_t)(sa / bs) * 2 * bs * bs + (sa % bs) * bs + (sb % bs)] = v;
    }
    __syncthreads();
    //   (b) one wavefront per block: D += A B with A[i][k] = rw_r J[r][i], B[k][j] = J[r][j], r = 4 step + k; lane l supplies
    //       A[l % 16][l / 16], B[l / 16][l % 16] and holds D[4 v + l / 16][l % 16]
    {
      typedef double d4 __attribute__((ext_vector_type(4)));
      const int wv = tid >> 6, lane = tid & 63, nwv = BT / 64;
      const int li = lane & 15, lk = lane >> 4;
      for (int t = wv; t < a.bt_nb; t += nwv) {
        const int *ids = a.mf_id + (size_t)t * 256;
        const int idd = li < bs ? ids[li * 16 + li] : -1;           // the diagonal entry of column li: its list walks J[:, li]
        const int h0 = idd >= 0 ? a.mf_h0[idd] : 0, nr = idd >= 0 ? a.mf_h1[idd] - h0 : 0;
        const int nrows = __shfl(nr, 0);                              // same for every column of the block (the plan checked)
        if (nrows == 0) continue;
        double *base = blk + (size_t)t * 2 * bs * bs;
        d4 acc;
#pragma unroll
        for (int v = 0; v < 4; v++) { const int i = 4 * v + lk; acc[v] = (i < bs && li < bs && li <= i) ? base[i * bs + li] : 0.0; }
        for (int st = 0; 4 * st < nrows; st++) {
          const int r = 4 * st + lk;
          double av = 0.0, bv = 0.0;
          if (r < nrows && idd >= 0) {
            const int q = h0 + r;
            const double jv = As[d.sa_pa[q]];
            av = Et[d.sa_row[q]] * jv; bv = jv;
          }
          acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int v = 0; v < 4; v++) { const int i = 4 * v + lk; if (i < bs && li <= i && ids[i * 16 + li] >= 0) base[i * bs + li] = acc[v]; }
      }
    }
    __syncthreads();
  }
  for (int id = tid; id < d.nS; id += BT) {
    const int sa = d.s_a[id], sb = d.s_b[id];
    double v;
    int t0 = d.sa_ptr[id];
    if (bs && a.mf_id && sa / bs == sb / bs) {
      v = blk[(size_t)(sa / bs) * 2 * bs * bs + (sa % bs) * bs + (sb % bs)];     // start value + hinge rows (above)
      t0 = a.mf_h1[id];
    } else {
      v = (sa == sb) ? a.sigma : 0.0;
      if (d.s_ppos[id] >= 0) v += Ps[d.s_ppos[id]];
    }
    for (int t = t0; t < d.sa_ptr[id + 1]; t++) v += Et[d.sa_row[t]] * As[d.sa_pa[t]] * As[d.sa_pb[t]];
    for (int t = d.ss_ptr[id]; t < d.ss_ptr[id + 1]; t++) v -= cpl[d.ss_k1[t]] * cpl[d.ss_k2[t]] * kinv[d.ss_e[t]];
    if (bs) {
      const int ta = sa / bs, tb = sb / bs, i = sa % bs, j = sb % bs;
      double *base = blk + (size_t)ta * 2 * bs * bs;
      if (ta == tb) { base[i * bs + j] = v; base[j * bs + i] = v; }
      else base[bs * bs + i * bs + j] = v;          // ta == tb + 1 (the plan checked the bandwidth)
    } else {
      S[(size_t)sa * n_c + sb] = v;       // lower triangle (sa >= sb)
    }
  }
  if (bs) return;                        // factorisation: qp_bt_factor_kernel
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

// Termination test on the unscaled residuals + infeasibility certificates (formulas of
// admm_check in sco_qp.hip); every thread of the workgroup calls it.  Returns the status
// (0 = keep iterating).
// Sparse dot products of the termination test with EIGHT entries in flight (r03).  One thread walks one row or column
// (a core column of 12-DOF x 50 holds 105 entries) and every entry is an index load followed by the gather it addresses:
// taken one at a time the walk is a chain of dependent memory round trips -- the test was 13 % of an iteration's time at
// 12-DOF x 50.  The eight index loads, then the eight gathers, go out together; the multiply-adds keep the order of the
// one-at-a-time loop, so every sum has the same bits as before.
#define BIG_CHK_U 8
// sum_s As[Rpos[s]] * vec[Rj[s]] over s in [s0, s1)
__device__ __forceinline__ double big_row_dot8(const QpDev &d, const double *As, const double *vec, int s0, int s1) {
  double acc = 0.0;
  for (int s = s0; s < s1; s += BIG_CHK_U) {
    int pos[BIG_CHK_U], cj[BIG_CHK_U];
    double av[BIG_CHK_U], xv[BIG_CHK_U];
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) { const int su = s + u < s1 ? s + u : s1 - 1; pos[u] = d.Rpos[su]; cj[u] = d.Rj[su]; }
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) { av[u] = As[pos[u]]; xv[u] = vec[cj[u]]; }
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) if (s + u < s1) acc += av[u] * xv[u];
  }
  return acc;
}
// sum_t Ps[Fpos[t]] * vec[Fi[t]] over t in [t0, t1)
__device__ __forceinline__ double big_p_dot8(const QpDev &d, const double *Ps, const double *vec, int t0, int t1) {
  double acc = 0.0;
  for (int t = t0; t < t1; t += BIG_CHK_U) {
    int pos[BIG_CHK_U], ci[BIG_CHK_U];
    double pv[BIG_CHK_U], xv[BIG_CHK_U];
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) { const int tu = t + u < t1 ? t + u : t1 - 1; pos[u] = d.Fpos[tu]; ci[u] = d.Fi[tu]; }
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) { pv[u] = Ps[pos[u]]; xv[u] = vec[ci[u]]; }
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) if (t + u < t1) acc += pv[u] * xv[u];
  }
  return acc;
}
// sum_t As[t] * yv[Ai[t]] * w[Ai[t]] (WFIRST: As[t] * (w[Ai[t]] * yv[Ai[t]])) over t in [t0, t1)
template <bool WFIRST>
__device__ __forceinline__ double big_col_dot8(const QpDev &d, const double *As, const double *yv, const int *w, int t0, int t1) {
  double acc = 0.0;
  for (int t = t0; t < t1; t += BIG_CHK_U) {
    int ri[BIG_CHK_U], wi[BIG_CHK_U];
    double av[BIG_CHK_U], yy[BIG_CHK_U];
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) { const int tu = t + u < t1 ? t + u : t1 - 1; ri[u] = d.Ai[tu]; av[u] = As[tu]; }
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++) { yy[u] = yv[ri[u]]; wi[u] = w[ri[u]]; }
#pragma unroll
    for (int u = 0; u < BIG_CHK_U; u++)
      if (t + u < t1) { if (WFIRST) acc += av[u] * ((double)wi[u] * yy[u]); else acc += av[u] * yy[u] * (double)wi[u]; }
  }
  return acc;
}

struct BigChk {
  const double *As, *Ps, *qs, *ls, *us, *Dg, *Eg, *x, *y, *z, *sdx;
  double *sdy;
  const int *w;
  double cscale;
};
template <int NT>
__device__ int big_check(const BigArgs &a, const BigChk &k, int iter, double *red, double &pri, double &dua) {
  const QpDev &d = a.d;
  const int tid = threadIdx.x, n = d.n, m = d.m;
  const double *As = k.As, *Ps = k.Ps, *qs = k.qs, *ls = k.ls, *us = k.us, *Dg = k.Dg, *Eg = k.Eg;
  const double *x = k.x, *y = k.y, *z = k.z, *sdx = k.sdx;
  double *sdy = k.sdy;
  const int *w = k.w;
  const double cscale = k.cscale;
  int status = 0;
  for (int approximate = 0; approximate < 2 && !status; approximate++) {
    if (approximate && iter < a.max_iter) break;
    const double cinv = 1.0 / cscale;
    double ea = a.eps_abs, er = a.eps_rel, epi = a.eps_prim_inf, edi = a.eps_dual_inf;
    if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
    double v[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int i = tid; i < m; i += NT) {
      const double ax = big_row_dot8(d, As, x, d.Rp[i], d.Rp[i + 1]);
      const double ei = 1.0 / Eg[i];
      v[0] = fmax(v[0], fabs(ei * (ax - z[i]))); v[1] = fmax(v[1], fabs(ei * z[i])); v[2] = fmax(v[2], fabs(ei * ax));
    }
    for (int j = tid; j < n; j += NT) {
      const double px = big_p_dot8(d, Ps, x, d.Fp[j], d.Fp[j + 1]);
      const double aty = big_col_dot8<false>(d, As, y, w, d.Ap[j], d.Ap[j + 1]);
      const double dj = 1.0 / Dg[j];
      v[3] = fmax(v[3], fabs(dj * (qs[j] + px + aty))); v[4] = fmax(v[4], fabs(dj * qs[j]));
      v[5] = fmax(v[5], fabs(dj * aty)); v[6] = fmax(v[6], fabs(dj * px));
    }
    bblock_reduce<7, true, NT>(v, red);
    pri = v[0]; dua = cinv * v[3];
    if (!(pri <= SCO_INFTY) || !(dua <= SCO_INFTY)) { status = SCO_QP_NON_CVX; break; }
    const double eps_p = ea + er * fmax(v[1], v[2]);
    const double eps_d = ea + er * cinv * fmax(v[4], fmax(v[5], v[6]));
    const bool prim_ok = (m == 0) || (pri < eps_p), dual_ok = dua < eps_d;
    if (prim_ok && dual_ok) { status = approximate ? SCO_QP_SOLVED_INACCURATE : SCO_QP_SOLVED; break; }
    if (!prim_ok) {
      double r1[1] = {0.0};
      for (int i = tid; i < m; i += NT) {
        double dy = sdy[i];
        if (us[i] > SCO_INFTY * SCO_MIN_SCALING) {
          if (ls[i] < -SCO_INFTY * SCO_MIN_SCALING) dy = 0.0; else dy = fmin(dy, 0.0);
        } else if (ls[i] < -SCO_INFTY * SCO_MIN_SCALING) dy = fmax(dy, 0.0);
        sdy[i] = dy;
        r1[0] = fmax(r1[0], fabs(Eg[i] * dy));
      }
      bblock_reduce<1, true, NT>(r1, red);
      const double ndy = r1[0];
      if (ndy > epi) {
        double lhs[1] = {0.0};
        for (int i = tid; i < m; i += NT) lhs[0] += (double)w[i] * (us[i] * fmax(sdy[i], 0.0) + ls[i] * fmin(sdy[i], 0.0));
        bblock_reduce<1, false, NT>(lhs, red);
        if (lhs[0] < -epi * ndy) {
          double nat[1] = {0.0};
          for (int j = tid; j < n; j += NT) {
            double aty = 0.0;
            for (int t = d.Ap[j]; t < d.Ap[j + 1]; t++) { const int i = d.Ai[t]; aty += As[t] * sdy[i] * (double)w[i]; }
            nat[0] = fmax(nat[0], fabs(aty / Dg[j]));
          }
          bblock_reduce<1, true, NT>(nat, red);
          if (nat[0] < epi * ndy) { status = approximate ? SCO_QP_PRIMAL_INFEASIBLE_INACCURATE : SCO_QP_PRIMAL_INFEASIBLE; break; }
        }
      }
    }
    if (!dual_ok) {
      double r1[1] = {0.0};
      for (int j = tid; j < n; j += NT) r1[0] = fmax(r1[0], fabs(Dg[j] * sdx[j]));
      bblock_reduce<1, true, NT>(r1, red);
      const double ndx = r1[0];
      if (ndx > edi) {
        double qdx[1] = {0.0};
        for (int j = tid; j < n; j += NT) qdx[0] += qs[j] * sdx[j];
        bblock_reduce<1, false, NT>(qdx, red);
        if (qdx[0] < -cscale * edi * ndx) {
          double npx[1] = {0.0};
          for (int j = tid; j < n; j += NT) {
            double px = 0.0;
            for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * sdx[d.Fi[t]];
            npx[0] = fmax(npx[0], fabs(px / Dg[j]));
          }
          bblock_reduce<1, true, NT>(npx, red);
          if (npx[0] < cscale * edi * ndx) {
            double bad[1] = {0.0};
            for (int i = tid; i < m; i += NT) {
              double adx = 0.0;
              for (int s = d.Rp[i]; s < d.Rp[i + 1]; s++) adx += As[d.Rpos[s]] * sdx[d.Rj[s]];
              adx /= Eg[i];
              if ((us[i] < SCO_INFTY * SCO_MIN_SCALING && adx > edi * ndx) ||
                  (ls[i] > -SCO_INFTY * SCO_MIN_SCALING && adx < -edi * ndx)) bad[0] = 1.0;
            }
            bblock_reduce<1, true, NT>(bad, red);
            if (bad[0] == 0.0) { status = approximate ? SCO_QP_DUAL_INFEASIBLE_INACCURATE : SCO_QP_DUAL_INFEASIBLE; break; }
          }
        }
      }
    }
  }
  return status;
}

// OSQP's rho estimate from the SCALED iterates (same rule as admm_rho_estimate in sco_qp.hip)
template <int NT>
__device__ double big_rho_estimate(const BigArgs &a, const BigChk &k, double *red, double rho) {
  const QpDev &d = a.d;
  const int tid = threadIdx.x, n = d.n, m = d.m;
  double v[7] = {0, 0, 0, 0, 0, 0, 0};
  for (int i = tid; i < m; i += NT) {
    const double ax = big_row_dot8(d, k.As, k.x, d.Rp[i], d.Rp[i + 1]);
    v[0] = fmax(v[0], fabs(ax - k.z[i])); v[1] = fmax(v[1], fabs(k.z[i])); v[2] = fmax(v[2], fabs(ax));
  }
  for (int j = tid; j < n; j += NT) {
    const double px = big_p_dot8(d, k.Ps, k.x, d.Fp[j], d.Fp[j + 1]);
    const double aty = big_col_dot8<true>(d, k.As, k.y, k.w, d.Ap[j], d.Ap[j + 1]);
    v[3] = fmax(v[3], fabs(px + k.qs[j] + aty)); v[4] = fmax(v[4], fabs(k.qs[j]));
    v[5] = fmax(v[5], fabs(aty)); v[6] = fmax(v[6], fabs(px));
  }
  bblock_reduce<7, true, NT>(v, red);
  const double pri = v[0] / (fmax(v[1], v[2]) + 1e-10);
  const double dua = v[3] / (fmax(v[4], fmax(v[5], v[6])) + 1e-10);
  return fmin(fmax(rho * sqrt(pri / (dua + 1e-10)), SCO_RHO_MIN), 1e6);
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

    {
      BigChk ck{As, Ps, qs, ls, us, Dg, Eg, x, y, z, sdx, sdy, w, cscale};
      status = big_check<BT>(a, ck, iter, red, pri, dua);
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


// ==========================================================================
// Structured tier ("bt"): block-tridiagonal core + dense row blocks
// ==========================================================================
// When the Schur complement S of the core is banded (half-bandwidth hb <= 16: a
// trajectory problem whose cost couples step t with t+1 and whose constraints touch
// one step each, hb = DOF) it is block tridiagonal with blocks of order BS >= hb, and
// the dense inverse W (n_c^2 doubles, 2.9 MB at 12x50, streamed from L2 every iteration
// by the dense route above) is replaced by the block LDL' factors
//     S = (I + E) D (I + E)',   E_t = S_{t,t-1} D_{t-1}^-1,   D_t = S_tt - E_t S_{t,t-1}'
// (2 nb BS^2 doubles: 115 KB at 12x50, resident in LDS).  One solve = forward sweep
// y_t = r_t - E_t y_{t-1} (one wavefront, nb dependent steps), z_t = D_t^-1 y_t (all
// threads), backward sweep x_t = z_t - E_{t+1}' x_{t+1}.
//
// Rows are processed in wavefront-sized chunks.  A "dense" chunk is a run of
// consecutive rows with the same core columns (one 100 x 12 block of the constraint
// Jacobian per timestep): its values are addressed as pos0[k] + lane, so the sweep over
// A reads coalesced lines and no index arrays.  A' t' is formed by scattering the
// products A_ic t'_i into CSC order (coalesced for the same reason) and summing every
// core column's contiguous segment with 16 lanes.
#define BT_MAXBS 16
#define BTT 512           // threads of the structured ADMM kernel (256 VGPRs per thread)
#define BTWV (BTT / 64)

bool bt_plan_build(const QpPlan &pl, const BigHost &bh, BtHost &th) {
  int hb = 0;
  for (int id = 0; id < pl.nS; id++) hb = std::max(hb, pl.s_a[id] - pl.s_b[id]);
  if (hb > BT_MAXBS || pl.n_c <= 0) return false;
  th.bs = hb <= 4 ? 4 : hb <= 8 ? 8 : hb <= 12 ? 12 : 16;
  th.nb = (pl.n_c + th.bs - 1) / th.bs;
  const size_t ncp = (size_t)th.nb * th.bs;
  th.blk_doubles = 2 * ncp * th.bs;
  const int m = pl.m;
  // core entries of every row: (core index, CSC position)
  auto row_core = [&](int i, std::vector<int> &cols, std::vector<int> &pos) {
    cols.clear(); pos.clear();
    for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
      const int c = pl.core_of[pl.Rj[s]];
      if (c >= 0) { cols.push_back(c); pos.push_back(pl.Rpos[s]); }
    }
  };
  // one item = an eliminated variable with its (at most two) rows, or a row without one
  struct Item { int e, j, r0, r1, ep0, ep1; bool dense_ok, simple; std::vector<int> cols, pos; };
  std::vector<Item> items;
  std::vector<int> c0, p0, c1, p1;
  for (int e = 0; e < pl.n_e; e++) {
    Item it; it.e = e; it.j = pl.elim_var[e]; it.r0 = it.r1 = it.ep0 = it.ep1 = -1; it.dense_ok = false; it.simple = true;
    const int nr = bh.er_ptr[e + 1] - bh.er_ptr[e];
    if (nr == 0) { items.push_back(it); continue; }
    int ra = bh.er_row[bh.er_ptr[e]], rb = nr > 1 ? bh.er_row[bh.er_ptr[e] + 1] : -1;
    row_core(ra, c0, p0);
    c1.clear(); p1.clear();
    if (rb >= 0) { row_core(rb, c1, p1); if (c1.size() > c0.size()) { std::swap(ra, rb); std::swap(c0, c1); std::swap(p0, p1); } }
    it.r0 = ra; it.r1 = rb; it.ep0 = bh.row_epos[ra]; it.ep1 = rb >= 0 ? bh.row_epos[rb] : -1;
    // dense: the primary row carries every core entry, in consecutive core columns at a constant CSC stride
    it.dense_ok = c1.empty() && !c0.empty() && (int)c0.size() <= 16;
    for (size_t k = 1; it.dense_ok && k < c0.size(); k++)
      it.dense_ok = c0[k] == c0[0] + (int)k && p0[k] - p0[k - 1] == p0[1] - p0[0];
    it.cols = c0; it.pos = p0;
    it.simple = c1.empty() && c0.size() <= 1;
    items.push_back(it);
  }
  th.ch_desc.clear(); th.it.clear(); th.npart = 0;
  auto push_chunk = [&](int kind, const std::vector<Item> &its, size_t first, size_t cnt) {
    std::vector<int> dsc(CH_STRIDE, 0);
    const Item &f = its[first];
    dsc[0] = kind; dsc[1] = (int)cnt;
    if (kind == 0) {
      dsc[2] = (int)f.cols.size(); dsc[3] = f.r0; dsc[4] = f.cols[0]; dsc[5] = f.pos[0];
      dsc[6] = f.pos.size() > 1 ? f.pos[1] - f.pos[0] : 0;
      dsc[7] = f.e; dsc[8] = f.j; dsc[9] = f.r1; dsc[10] = f.ep0; dsc[12] = f.ep1;
      if (cnt > 1) { dsc[11] = its[first + 1].ep0 - f.ep0; dsc[13] = its[first + 1].ep1 - f.ep1; }
      dsc[14] = th.npart; th.npart += (int)f.cols.size();
    }
    th.ch_desc.insert(th.ch_desc.end(), dsc.begin(), dsc.end());
    for (size_t l = 0; l < 64; l++) {
      const bool on = l < cnt;
      const Item *t = on ? &its[first + l] : nullptr;
      const bool one = on && t->cols.size() == 1;
      const int rec[8] = {on ? t->e : -1, on ? t->j : -1, on ? t->r0 : -1, on ? t->r1 : -1, on ? t->ep0 : -1,
                          on ? t->ep1 : -1, one ? t->cols[0] : -1, one ? t->pos[0] : -1};
      th.it.insert(th.it.end(), rec, rec + 8);
    }
  };
  // dense runs: consecutive eliminated variables whose primary rows are consecutive rows with the
  // same core columns, and whose every index is an affine function of the position in the run
  std::vector<Item> generic;
  size_t i0 = 0;
  while (i0 < items.size()) {
    size_t i1 = i0 + 1;
    if (items[i0].dense_ok) {
      while (i1 < items.size() && items[i1].dense_ok) {
        const Item &p = items[i1 - 1], &c = items[i1], &f = items[i0], &g = items[i0 + 1];
        bool ok = c.r0 == p.r0 + 1 && c.j == p.j + 1 && c.cols == f.cols && (c.r1 < 0) == (f.r1 < 0) &&
                  (c.r1 < 0 || c.r1 == p.r1 + 1) && c.ep0 - p.ep0 == g.ep0 - f.ep0 && c.ep1 - p.ep1 == g.ep1 - f.ep1;
        for (size_t k = 0; ok && k < f.pos.size(); k++) ok = c.pos[k] == p.pos[k] + 1;
        if (!ok) break;
        i1++;
      }
    }
    if (items[i0].dense_ok && i1 - i0 >= 16) {
      for (size_t f = i0; f < i1; f += 64) push_chunk(0, items, f, std::min<size_t>(64, i1 - f));
    } else {
      for (size_t f = i0; f < i1; f++) generic.push_back(items[f]);
    }
    i0 = i1;
  }
  for (int i : bh.free_rows) {
    Item it; it.e = it.j = it.r1 = it.ep0 = it.ep1 = -1; it.r0 = i; it.dense_ok = false; it.simple = true;
    row_core(i, it.cols, it.pos);
    if (it.cols.size() > 1) it.simple = false;
    generic.push_back(it);
  }
  // kind 2: items whose rows hold at most one core entry (bound rows, pins) -- direct indices;
  // kind 1: everything else walks the CSR arrays
  std::vector<Item> simple, complex_;
  for (const Item &it : generic) (it.simple ? simple : complex_).push_back(it);
  for (size_t f = 0; f < simple.size(); f += 64) push_chunk(2, simple, f, std::min<size_t>(64, simple.size() - f));
  for (size_t f = 0; f < complex_.size(); f += 64) push_chunk(1, complex_, f, std::min<size_t>(64, complex_.size() - f));
  th.nchunks = (int)(th.ch_desc.size() / CH_STRIDE);
  // contributions to every core column of A' t: partial sums of the dense chunks + products of the other rows
  th.cent.assign(4 * ncp, -1);
  th.use_part = th.npart > 0;
  {
    std::vector<int> fill(ncp, 0);
    auto add = [&](int c, int v) { if (fill[c] < 4) th.cent[4 * c + fill[c]] = v; fill[c]++; };
    for (int ch = 0; ch < th.nchunks; ch++) {
      const int *dsc = &th.ch_desc[(size_t)ch * CH_STRIDE];
      if (dsc[0] == 0) {
        for (int k = 0; k < dsc[2]; k++) add(dsc[4] + k, dsc[14] + k);
      } else {
        for (int l = 0; l < dsc[1]; l++) {
          const int *rec = &th.it[((size_t)ch * 64 + l) * 8];
          for (int q = 2; q <= 3; q++) {
            const int i = rec[q];
            if (i < 0) continue;
            for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
              const int c = pl.core_of[pl.Rj[s]];
              if (c >= 0) add(c, -(pl.Rpos[s] + 2));
            }
          }
        }
      }
    }
    for (size_t c = 0; c < ncp; c++) if (fill[c] > 4) th.use_part = false;
  }
  if (!th.use_part) th.npart = 0;
  // factors; r/y, z/x~ (+ 64 zeros), x, q; partial sums; column metadata; chunk descriptors
  th.lds_bytes = 8 * (th.blk_doubles + 4 * ncp + 64 + th.npart) + 4 * (4 * ncp + th.ch_desc.size() + th.nchunks);
  if (th.lds_bytes + 1024 > 160 * 1024) return false;
  th.ws_doubles = 3 * (size_t)m + pl.n_e + pl.n + pl.nnzA + 64;
  // ---- N1: MFMA formation of the diagonal blocks' J' R J (blocks of order 12 .. 16, dense hinge rows)
  th.use_mfma = false;
  { const char *e = getenv("SCO_QP_NO_MFMA"); if (e && e[0] == '1') { th.mf_why = 1; return true; } }
  th.mf_why = 2;
  if (th.bs >= 12 && th.bs <= 16) {
    const int bs = th.bs, nb = th.nb;
    // hinge rows of a block: one eliminated variable, core entries in that block only
    std::vector<int> row_blk(m, -1);
    for (int i = 0; i < m; i++) {
      int ne = 0, blk = -1; bool ok = true;
      for (int s2 = pl.Rp[i]; s2 < pl.Rp[i + 1]; s2++) {
        const int j = pl.Rj[s2];
        if (pl.elim_of[j] >= 0) ne++;
        else { const int t = pl.core_of[j] / bs; if (blk >= 0 && t != blk) ok = false; blk = t; }
      }
      if (ne == 1 && blk >= 0 && ok) row_blk[i] = blk;
    }
    th.mf_id.assign((size_t)nb * 256, -1); th.mf_h0.assign(pl.nS, 0); th.mf_h1.assign(pl.nS, 0);
    bool ok = true; int used = 0;
    std::vector<std::vector<int>> ref_rows(nb);
    for (int id = 0; id < pl.nS && ok; id++) {
      const int sa = pl.s_a[id], sb = pl.s_b[id], t = sa / bs;
      if (sb / bs != t) continue;
      int h0 = -1, h1 = -1;
      std::vector<int> rows;
      for (int q = pl.sa_ptr[id]; q < pl.sa_ptr[id + 1]; q++)
        if (row_blk[pl.sa_row[q]] == t) { if (h0 < 0) h0 = q; else if (q != h1) { ok = false; th.mf_why = 3; } h1 = q + 1; rows.push_back(pl.sa_row[q]); }
      if (h0 < 0) { h0 = h1 = pl.sa_ptr[id]; }
      // every entry of the block sums over the SAME hinge rows in the same order (dense Jacobian block): else no MFMA
      if (sa % bs == 0 && sb % bs == 0) ref_rows[t] = rows;
      th.mf_h0[id] = h0; th.mf_h1[id] = h1;
      th.mf_id[(size_t)t * 256 + (sa % bs) * 16 + (sb % bs)] = id;
    }
    for (int id = 0; id < pl.nS && ok; id++) {
      const int sa = pl.s_a[id], sb = pl.s_b[id], t = sa / bs;
      if (sb / bs != t) continue;
      if (th.mf_h1[id] - th.mf_h0[id] != (int)ref_rows[t].size()) { ok = false; th.mf_why = 4; break; }
      for (int q = th.mf_h0[id]; q < th.mf_h1[id]; q++) if (pl.sa_row[q] != ref_rows[t][q - th.mf_h0[id]]) { ok = false; th.mf_why = 4; break; }
      used += th.mf_h1[id] > th.mf_h0[id];
    }
    // (every lower-triangle position of every full block must be a structural entry: the diagonal always is)
    for (int t = 0; t < nb && ok; t++)
      for (int i = 0; i < bs && ok; i++)
        for (int j = 0; j <= i && ok; j++)
          if ((size_t)t * bs + i < (size_t)pl.n_c && th.mf_id[(size_t)t * 256 + i * 16 + j] < 0 && !ref_rows[t].empty()) { ok = false; th.mf_why = 5; }
    if (ok) th.mf_why = used > 0 ? 0 : 6;
    th.use_mfma = ok && used > 0;
    if (!th.use_mfma) { th.mf_id.clear(); th.mf_h0.clear(); th.mf_h1.clear(); }
  }
  return true;
}

// ---- twisted block LDL' of S in place ---------------------------------------------------
// Blocks 0 .. mid-1 are eliminated top-down, blocks nb-1 .. mid+1 bottom-up, block `mid` last, so a
// solve is two independent chains of about nb/2 dependent steps (one wavefront each) instead of one
// chain of nb:
//   top     D_t = S_tt - E_t S_{t,t-1}',      E_t = S_{t,t-1} D_{t-1}^-1          (t = 1 .. mid)
//   bottom  D_t = S_tt - G_t S_{t+1,t},       G_t = S_{t+1,t}' D_{t+1}^-1         (t = nb-2 .. mid)
//   middle  D_mid = S_mm - E_mid S_{mid,mid-1}' - G_mid S_{mid+1,mid},   H = D_mid^-1 G_mid
// Storage: [t][0] <- D_t^-1;  [t][1] <- E_t for 1 <= t <= mid, G_{t-1} for t > mid (the slot of
// S_{t,t-1}, which both use up);  [0][1] <- H.
template <int BS>
__device__ __forceinline__ void bt_invert_spd(double *M, double *rowk, double *colk, int tid) {
  // Gauss-Jordan inversion of a symmetric positive definite block (no pivoting needed)
  const int i = tid / BS, j = tid % BS;
  const bool on = tid < BS * BS;
  for (int k = 0; k < BS; k++) {
    if (tid < BS) {
      const double piv = M[k * BS + k];
      rowk[tid] = (tid == k ? 1.0 : M[k * BS + tid]) / piv;
      colk[tid] = M[tid * BS + k];
    }
    __syncthreads();
    if (on) {
      const double v = (j == k) ? 0.0 : M[tid];
      M[tid] = (i == k) ? rowk[j] : v - colk[i] * rowk[j];
    }
    __syncthreads();
  }
}

template <int BS>
__global__ __launch_bounds__(256) void qp_bt_factor_kernel(BigArgs a) {
  const QpDev &d = a.d;
  const int b = d.list ? d.list[blockIdx.x + d.b0] : (int)blockIdx.x + d.b0, tid = threadIdx.x;
  if (b < 0 || (d.active && !d.active[b])) return;
  constexpr int BB = BS * BS;
  __shared__ double M[BB], Sub[BB], Ev[BB], Cb[BB], rowk[BS], colk[BS];
  double *blk = a.bt_blk + (size_t)b * a.bt_stride;
  const int i = tid / BS, j = tid % BS, nb = a.bt_nb, mid = a.bt_mid;
  const bool on = tid < BB;
  // ---- bottom chain: t = nb-1 .. mid+1; leaves Cb = G_mid S_{mid+1,mid} for the middle block
  if (on) { Cb[tid] = 0.0; if (nb - 1 > mid) M[tid] = blk[(size_t)(nb - 1) * 2 * BB + tid]; }
  __syncthreads();
  for (int t = nb - 1; t > mid; t--) {
    bt_invert_spd<BS>(M, rowk, colk, tid);
    double *cur = blk + (size_t)t * 2 * BB;
    if (on) { cur[tid] = M[tid]; Sub[tid] = cur[BB + tid]; }            // Sub = S_{t,t-1}
    __syncthreads();
    if (on) {
      double g = 0.0;                                                    // G_{t-1} = S_{t,t-1}' D_t^-1
#pragma unroll
      for (int k = 0; k < BS; k++) g += Sub[k * BS + i] * M[k * BS + j];
      Ev[tid] = g; cur[BB + tid] = g;
    }
    __syncthreads();
    if (on) {
      double v = 0.0;                                                    // G_{t-1} S_{t,t-1}
#pragma unroll
      for (int k = 0; k < BS; k++) v += Ev[i * BS + k] * Sub[k * BS + j];
      if (t - 1 > mid) M[tid] = blk[(size_t)(t - 1) * 2 * BB + tid] - v;
      else Cb[tid] = v;
    }
    __syncthreads();
  }
  // ---- top chain: t = 0 .. mid
  if (on) M[tid] = blk[tid] - (mid == 0 ? Cb[tid] : 0.0);
  __syncthreads();
  for (int t = 0; t <= mid; t++) {
    bt_invert_spd<BS>(M, rowk, colk, tid);
    double *cur = blk + (size_t)t * 2 * BB;
    if (on) cur[tid] = M[tid];
    if (t < mid) {
      double *nxt = cur + 2 * BB;
      if (on) Sub[tid] = nxt[BB + tid];                                  // S_{t+1,t}
      __syncthreads();
      if (on) {
        double e = 0.0;                                                  // E_{t+1} = S_{t+1,t} D_t^-1
#pragma unroll
        for (int k = 0; k < BS; k++) e += Sub[i * BS + k] * M[k * BS + j];
        Ev[tid] = e; nxt[BB + tid] = e;
      }
      __syncthreads();
      if (on) {
        double v = nxt[tid];
#pragma unroll
        for (int k = 0; k < BS; k++) v -= Ev[i * BS + k] * Sub[j * BS + k];
        if (t + 1 == mid) v -= Cb[tid];
        M[tid] = v;
      }
      __syncthreads();
    }
  }
  // ---- H = D_mid^-1 G_mid (G_mid sits in slot [mid+1][1]) -> slot [0][1]
  if (nb - 1 > mid) {
    if (on) Sub[tid] = blk[(size_t)(mid + 1) * 2 * BB + BB + tid];
    __syncthreads();
    if (on) {
      double h = 0.0;
#pragma unroll
      for (int k = 0; k < BS; k++) h += M[i * BS + k] * Sub[k * BS + j];
      blk[BB + tid] = h;
    }
  }
}

// ---- ADMM ---------------------------------------------------------------------------
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double bt_row16_sum(double v) {     // total of a 16-lane DPP row, valid in its lane 15
  int lo, hi, lo2, hi2;
#define BT_DPP_STEP(ctrl)                                                                    \
  lo = __double2loint(v); hi = __double2hiint(v);                                            \
  lo2 = __builtin_amdgcn_update_dpp(0, lo, ctrl, 0xf, 0xf, true);                            \
  hi2 = __builtin_amdgcn_update_dpp(0, hi, ctrl, 0xf, 0xf, true);                            \
  v += __hiloint2double(hi2, lo2);
  BT_DPP_STEP(0x111) BT_DPP_STEP(0x112) BT_DPP_STEP(0x114) BT_DPP_STEP(0x118)
#undef BT_DPP_STEP
  return v;
}

// per-problem pointers of the structured kernel
struct BtPtrs {
  const double *As, *rho, *ls, *us, *kinv, *qs;
  const int *w;
  double *x, *y, *z, *sdy, *sdx, *ge, *prod;
  const double *s_xc;
  double *s_part;       // partial column sums of the dense chunks (LDS) or null: products go to `prod`
  double alpha, sigma;
  double *v;                // pre-clamp value of every dense-chunk row (bt_row_step_v)
  const int *s_cflag;       // LDS: this problem's chunk flags (qp_setup_big_kernel), all zero if there are none
  double rho0, wk, rinv0;   // the constants the flags refer to: base rho, common weight; 1 / base rho
};
struct BtRow { double zc, ae, rh, w, z, y, l, u, ri; };     // ri = 1 / rh (dense chunks: formed once per problem where rho is the base rho)
// one row of the ADMM update; returns t_i = w (rho z+ - y+)
__device__ __forceinline__ double bt_row_step(const BtRow &r, double alpha, double xte, double &zn, double &yn, double &dy) {
  const double zt = r.zc + r.ae * xte;
  const double zr = alpha * zt + (1.0 - alpha) * r.z;
  zn = zr + (1.0 / r.rh) * r.y;
  zn = fmin(fmax(zn, r.l), r.u);
  dy = r.rh * (zr - zn);
  yn = r.y + dy;
  return r.w * (r.rh * zn - yn);
}
// t_i = w (rho z - y) and g_e = ((sigma x_e - q_e) + a_0 t_0 + a_1 t_1) / K_ee.  (Re-deriving g_e at the next iteration from v and
// x instead of storing it saves 16 B per row pair and costs more VALU than it gains: 70.1 -> 69.6 us per iteration with 256
// problems resident, 58.2 -> 60.1 with 128; not kept.)
__device__ __forceinline__ double bt_tq(const BtRow &r, double z, double y) { return r.w * (r.rh * z - y); }
__device__ __forceinline__ double bt_ge(double sigma, double x, double qj, double ae0, double tq0, double ae1, double tq1, double ki) {
  return ((sigma * x - qj) + ae0 * tq0 + ae1 * tq1) * ki;
}
// The same update for the rows of dense chunks, which keep ONE number per row between iterations: v = z~ + y / rho, the
// value before the clamp.  The iterates are z = clamp(v, l, u) and y = rho (v - z) -- what OSQP's update produces in
// exact arithmetic (y+ = y + rho (z~ - z+) = rho (v - z+)) -- so z and y need not be stored and loaded separately
// (16 B less per row and iteration in each direction); they are written out on the iterations the termination test,
// a park or the end of the solve read them.  r.z / r.y come from bt_row_zy on all but the first iteration of a launch.
__device__ __forceinline__ double bt_row_step_v(const BtRow &r, double alpha, double xte, double &zn, double &yn, double &dy, double &v) {
  const double zt = r.zc + r.ae * xte;
  const double zr = alpha * zt + (1.0 - alpha) * r.z;
  v = zr + r.ri * r.y;
  zn = fmin(fmax(v, r.l), r.u);
  dy = r.rh * (zr - zn);
  yn = r.rh * (v - zn);
  return bt_tq(r, zn, yn);
}
__device__ __forceinline__ void bt_row_zy(BtRow &r, double v) { r.z = fmin(fmax(v, r.l), r.u); r.y = r.rh * (v - r.z); }

// sum over the wavefront, valid in lane 63; fixed association order (16-lane rows by
// row_shr 1,2,4,8, then row_bcast15 into rows 1 and 3, then row_bcast31 into rows 2 and 3)
__device__ __forceinline__ double bt_wave_sum63(double v) {
  int lo, hi, lo2, hi2;
#define BT_DPP_STEP(ctrl, rmask)                                                             \
  lo = __double2loint(v); hi = __double2hiint(v);                                            \
  lo2 = __builtin_amdgcn_update_dpp(0, lo, ctrl, rmask, 0xf, true);                          \
  hi2 = __builtin_amdgcn_update_dpp(0, hi, ctrl, rmask, 0xf, true);                          \
  v += __hiloint2double(hi2, lo2);
  BT_DPP_STEP(0x111, 0xf) BT_DPP_STEP(0x112, 0xf) BT_DPP_STEP(0x114, 0xf) BT_DPP_STEP(0x118, 0xf)
  BT_DPP_STEP(0x142, 0xa) BT_DPP_STEP(0x143, 0xc)
#undef BT_DPP_STEP
  return v;
}

// Column totals of NC per-lane values over the wavefront with the gfx950 lane-swap instructions:
// v_permlane32_swap folds the two 32-lane halves of two columns into one register, v_permlane16_swap
// folds the 16-lane rows of two such registers, and a row_shr scan finishes inside the rows: 63
// instructions for 12 columns instead of 12 full wavefront reductions.  Fixed association order.
typedef unsigned int bt_uint2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double bt_fold32(double a, double b) {
  const bt_uint2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const bt_uint2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double bt_fold16(double a, double b) {
  const bt_uint2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const bt_uint2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
// v[k] = value of column k in this lane (0 for idle lanes / unused columns); dst[k] <- sum over the wavefront
template <int NC>
__device__ __forceinline__ void bt_reduce_cols(const double (&v)[NC], double *dst, int ncols, int lane) {
  static_assert(NC % 4 == 0, "columns are processed four at a time");
  double u[NC / 2], t[NC / 4];
#pragma unroll
  for (int m = 0; m < NC / 2; m++) u[m] = bt_fold32(v[2 * m], v[2 * m + 1]);     // halves: column 2m | 2m+1
#pragma unroll
  for (int n = 0; n < NC / 4; n++) t[n] = bt_fold16(u[2 * n], u[2 * n + 1]);     // rows: columns 4n, 4n+2, 4n+1, 4n+3
#pragma unroll
  for (int n = 0; n < NC / 4; n++) t[n] = bt_row16_sum(t[n]);                    // lane 15 of every row: the total
  if ((lane & 15) == 15) {
    const int row = lane >> 4;
    const int off = row == 0 ? 0 : row == 1 ? 2 : row == 2 ? 1 : 3;
#pragma unroll
    for (int n = 0; n < NC / 4; n++)
      if (4 * n + off < ncols) dst[4 * n + off] = t[n];
  }
}

// Dense chunk: up to 64 consecutive rows r0 + lane with core entries in columns c0 .. c0 + ncols - 1
// at CSC positions pos0 + k cs + lane; their eliminated variables e0 + lane (QP variable j0 + lane)
// each with an optional second row r1 + lane that has no core entry.  Every address is a
// descriptor scalar plus the (clamped) lane and every load is unconditional, so a chunk costs one
// memory round trip; two chunks are kept in flight per wavefront.  All 64 lanes run the arithmetic
// (idle lanes repeat the last row and store nothing) so that A' t can be reduced across the wavefront.
template <int NC>
struct BtDenseRegs {
  double av[NC];
  BtRow p, s;
  double g, ki, xo, qj;
  int r0, r1, e, j, ncols, pos0, cs, c0, pbase, ln;
  bool two, on;
};
// FULL: the chunk uses all NC columns of its instantiation (ncols is then a compile-time constant and the column masks --
// a compare, a 64-bit select mask and, with the scalar registers this kernel spills, three v_readlane reloads per column
// -- drop out of the loads, the dot product, the products and the stores of the column totals)
// Element `idx` (>= 0; the arrays are far below 4 GB) of a per-problem array with a wave-uniform base: the byte offset is
// formed in 32 bits, so the access is `global_load v, v_off, s[base:base+1]` -- one 32-bit shift per lane instead of a sign
// extension and a 64-bit shift-add (r03: ~30 accesses per chunk).
__device__ __forceinline__ double bt_ldg(const double *base, int idx) { return *(const double *)((const char *)base + (unsigned)idx * 8u); }
__device__ __forceinline__ int bt_ldgi(const int *base, int idx) { return *(const int *)((const char *)base + (unsigned)idx * 4u); }
__device__ __forceinline__ void bt_stg(double *base, int idx, double v) { *(double *)((char *)base + (unsigned)idx * 8u) = v; }
template <int NC, bool FULL = false>
__device__ __forceinline__ void bt_dense_load(const int (&dsc)[CH_STRIDE], int ch, int lane, bool first, const BtPtrs &q, BtDenseRegs<NC> &R) {
  R.on = lane < dsc[1];
  R.ln = R.on ? lane : dsc[1] - 1;
  R.ncols = FULL ? NC : dsc[2]; R.pos0 = dsc[5]; R.cs = dsc[6]; R.c0 = dsc[4]; R.pbase = dsc[14];
  R.r0 = dsc[3] + R.ln; R.e = dsc[7] + R.ln; R.j = dsc[8] + R.ln;
  R.two = dsc[9] >= 0;
  R.r1 = R.two ? dsc[9] + R.ln : R.r0;
#pragma unroll
  for (int k = 0; k < NC; k++) R.av[k] = bt_ldg(q.As, R.pos0 + (k < R.ncols ? k : 0) * R.cs + R.ln);
  const int ep0 = dsc[10] + dsc[11] * R.ln, ep1 = R.two ? dsc[12] + dsc[13] * R.ln : ep0;
  // per-row constants the chunk's rows are known to share are not loaded (flags: qp_setup_big_kernel; wave-uniform)
  const int fl = __builtin_amdgcn_readfirstlane(q.s_cflag[ch]), fs = fl >> 8;
  // first iteration of a launch: z and y as the start / the previous launch left them; afterwards one number per row
  double vp = 0.0, vs = 0.0;
  R.p.zc = 0.0; R.p.ae = bt_ldg(q.As, ep0); R.s.zc = 0.0; R.s.ae = bt_ldg(q.As, ep1);
  if (first) { R.p.z = bt_ldg(q.z, R.r0); R.p.y = bt_ldg(q.y, R.r0); R.s.z = bt_ldg(q.z, R.r1); R.s.y = bt_ldg(q.y, R.r1); }
  else { vp = bt_ldg(q.v, R.r0); vs = bt_ldg(q.v, R.r1); }
  if (fl & 1) R.p.l = -SCO_INFTY; else if (fl & 2) R.p.l = 0.0; else R.p.l = bt_ldg(q.ls, R.r0);
  if (fl & 4) R.p.u = SCO_INFTY; else R.p.u = bt_ldg(q.us, R.r0);
  if (fl & 8) { R.p.rh = q.rho0; R.p.ri = q.rinv0; } else { R.p.rh = bt_ldg(q.rho, R.r0); R.p.ri = 1.0 / R.p.rh; }
  if (fl & 16) R.p.w = 1.0; else if (fl & 32) R.p.w = q.wk; else R.p.w = (double)bt_ldgi(q.w, R.r0);
  if (!R.two) { R.s.l = R.p.l; R.s.u = R.p.u; R.s.rh = R.p.rh; R.s.ri = R.p.ri; R.s.w = R.p.w; }      // r1 = r0: never used
  else {
    if (fs & 1) R.s.l = -SCO_INFTY; else if (fs & 2) R.s.l = 0.0; else R.s.l = bt_ldg(q.ls, R.r1);
    if (fs & 4) R.s.u = SCO_INFTY; else R.s.u = bt_ldg(q.us, R.r1);
    if (fs & 8) { R.s.rh = q.rho0; R.s.ri = q.rinv0; } else { R.s.rh = bt_ldg(q.rho, R.r1); R.s.ri = 1.0 / R.s.rh; }
    if (fs & 16) R.s.w = 1.0; else if (fs & 32) R.s.w = q.wk; else R.s.w = (double)bt_ldgi(q.w, R.r1);
  }
  if (!first) { bt_row_zy(R.p, vp); bt_row_zy(R.s, vs); }
  R.g = bt_ldg(q.ge, R.e); R.ki = bt_ldg(q.kinv, R.e); R.xo = bt_ldg(q.x, R.j); R.qj = bt_ldg(q.qs, R.j);
}
template <int NC>
__device__ __forceinline__ void bt_dense_compute(BtDenseRegs<NC> &R, int lane, bool chk, const BtPtrs &q) {
  const double *xc = q.s_xc + R.c0;
  double xv[NC];
#pragma unroll
  for (int k = 0; k < NC; k++) xv[k] = xc[k];             // the slack behind x~_C is zero
  double z0 = 0.0, z1 = 0.0;
#pragma unroll
  for (int k = 0; k < NC; k += 2) {
    z0 += (k < R.ncols ? R.av[k] : 0.0) * xv[k];
    z1 += (k + 1 < R.ncols ? R.av[k + 1] : 0.0) * xv[k + 1];
  }
  R.p.zc = z0 + z1;
  if (!R.two) { R.s.ae = 0.0; R.s.w = 0.0; }
  const double rwp = R.p.w * R.p.rh;
  const double xte = R.g - R.ki * (rwp * R.p.ae * R.p.zc);         // the second row has no core entry
  double zn, yn, dy, vv;
  const double tq0 = bt_row_step_v(R.p, q.alpha, xte, zn, yn, dy, vv);
  if (R.on) { bt_stg(q.v, R.r0, vv); if (chk) { bt_stg(q.z, R.r0, zn); bt_stg(q.y, R.r0, yn); bt_stg(q.sdy, R.r0, dy); } }
  double tq1 = 0.0;
  if (R.two) {
    tq1 = bt_row_step_v(R.s, q.alpha, xte, zn, yn, dy, vv);
    if (R.on) { bt_stg(q.v, R.r1, vv); if (chk) { bt_stg(q.z, R.r1, zn); bt_stg(q.y, R.r1, yn); bt_stg(q.sdy, R.r1, dy); } }
  }
  const double xn = q.alpha * xte + (1.0 - q.alpha) * R.xo;
  const double gn = bt_ge(q.sigma, xn, R.qj, R.p.ae, tq0, R.s.ae, tq1, R.ki);
  if (R.on) { if (chk) bt_stg(q.sdx, R.j, xn - R.xo); bt_stg(q.x, R.j, xn); bt_stg(q.ge, R.e, gn); }
  const double t0 = R.on ? tq0 - rwp * R.p.ae * gn : 0.0;
  if (q.s_part) {
    double pv[NC];
#pragma unroll
    for (int k = 0; k < NC; k++) pv[k] = (k < R.ncols ? R.av[k] : 0.0) * t0;
    bt_reduce_cols<NC>(pv, q.s_part + R.pbase, R.ncols, lane);
  } else if (R.on) {
#pragma unroll
    for (int k = 0; k < NC; k++)
      if (k < R.ncols) bt_stg(q.prod, R.pos0 + k * R.cs + lane, R.av[k] * t0);
  }
}
// ---- three chunks in flight (r03) ------------------------------------------------------------------------------
// Between its loads and its arithmetic a chunk is only what was loaded: the values of A, the two slack coefficients, v of
// both rows, the upper bound of the primary row and the four numbers of the eliminated variable -- 21 doubles per lane
// at 12 columns (+ 4 row / variable indices).  That is what BtLean holds; the full BtDenseRegs (78 registers) is built
// from it, the descriptor and the chunk's flags only when the arithmetic starts, so THREE chunks' loads fit the
// register file where two full sets did.  Eligible (bt_lean_ok): chunks of full width whose flags say that every other
// per-row constant is a known value -- lower bounds, rho and weights of both rows, the upper bound of the secondary rows
// (hinge rows with their slack rows, qp_setup_big_kernel) -- on every iteration but the first of a launch (which reads z
// and y, not v).
template <int NC>
struct BtLean { double av[NC], ae0, ae1, v0, v1, u0, g, ki, xo, qj; int r0, r1, e, j; };
__device__ __forceinline__ bool bt_lean_ok(int fl, bool two) {
  const int fs = fl >> 8;
  return (fl & 3) && (fl & 8) && (fl & 48) && (!two || ((fs & 3) && (fs & 4) && (fs & 8) && (fs & 48)));
}
template <int NC>
__device__ __forceinline__ void bt_lean_load(const int (&dsc)[CH_STRIDE], int fl, int lane, const BtPtrs &q, BtLean<NC> &L) {
  const int ln = lane < dsc[1] ? lane : dsc[1] - 1;
  const bool two = dsc[9] >= 0;
  L.r0 = dsc[3] + ln; L.r1 = two ? dsc[9] + ln : L.r0; L.e = dsc[7] + ln; L.j = dsc[8] + ln;
#pragma unroll
  for (int k = 0; k < NC; k++) L.av[k] = bt_ldg(q.As, dsc[5] + k * dsc[6] + ln);
  const int ep0 = dsc[10] + dsc[11] * ln, ep1 = two ? dsc[12] + dsc[13] * ln : ep0;
  L.ae0 = bt_ldg(q.As, ep0); L.ae1 = bt_ldg(q.As, ep1);
  L.v0 = bt_ldg(q.v, L.r0); L.v1 = bt_ldg(q.v, L.r1);
  L.u0 = SCO_INFTY;
  if (!(fl & 4)) L.u0 = bt_ldg(q.us, L.r0);
  L.g = bt_ldg(q.ge, L.e); L.ki = bt_ldg(q.kinv, L.e); L.xo = bt_ldg(q.x, L.j); L.qj = bt_ldg(q.qs, L.j);
}
template <int NC>
__device__ __forceinline__ void bt_lean_compute(const int *s_dsc, int ch, int lane, bool chk, const BtPtrs &q, const BtLean<NC> &L) {
  const int nact = __builtin_amdgcn_readfirstlane(s_dsc[ch * CH_STRIDE + 1]);
  const int fl = __builtin_amdgcn_readfirstlane(q.s_cflag[ch]), fs = fl >> 8;
  BtDenseRegs<NC> R;
  R.on = lane < nact;
  R.ln = R.on ? lane : nact - 1;
  R.ncols = NC; R.pos0 = 0; R.cs = 0;
  R.c0 = __builtin_amdgcn_readfirstlane(s_dsc[ch * CH_STRIDE + 4]);
  R.pbase = __builtin_amdgcn_readfirstlane(s_dsc[ch * CH_STRIDE + 14]);
  R.two = __builtin_amdgcn_readfirstlane(s_dsc[ch * CH_STRIDE + 9]) >= 0;
  R.r0 = L.r0; R.r1 = L.r1; R.e = L.e; R.j = L.j;
#pragma unroll
  for (int k = 0; k < NC; k++) R.av[k] = L.av[k];
  R.p.zc = 0.0; R.p.ae = L.ae0; R.s.zc = 0.0; R.s.ae = L.ae1;
  R.p.l = (fl & 1) ? -SCO_INFTY : 0.0; R.p.u = L.u0; R.p.rh = q.rho0; R.p.ri = q.rinv0; R.p.w = (fl & 16) ? 1.0 : q.wk;
  if (!R.two) { R.s.l = R.p.l; R.s.u = R.p.u; R.s.rh = R.p.rh; R.s.ri = R.p.ri; R.s.w = R.p.w; }
  else { R.s.l = (fs & 1) ? -SCO_INFTY : 0.0; R.s.u = SCO_INFTY; R.s.rh = q.rho0; R.s.ri = q.rinv0; R.s.w = (fs & 16) ? 1.0 : q.wk; }
  bt_row_zy(R.p, L.v0); bt_row_zy(R.s, L.v1);
  R.g = L.g; R.ki = L.ki; R.xo = L.xo; R.qj = L.qj;
  bt_dense_compute<NC>(R, lane, chk, q);
}
// chunks A and, if hasB, B: loads of both before the arithmetic of either
template <int NC, bool FULL = false>
__device__ __forceinline__ void bt_dense_pair(const int (&dA)[CH_STRIDE], const int (&dB)[CH_STRIDE], int chA, int chB, bool hasB, int lane,
                                              bool chk, bool first, const BtPtrs &q) {
  BtDenseRegs<NC> RA, RB;
  bt_dense_load<NC, FULL>(dA, chA, lane, first, q, RA);
  if (hasB) bt_dense_load<NC, FULL>(dB, chB, lane, first, q, RB);
  bt_dense_compute<NC>(RA, lane, chk, q);
  if (hasB) bt_dense_compute<NC>(RB, lane, chk, q);
}

// One chain of the twisted block solve, in place on `vec`: for s = 0 .. nsteps-1, t = t0 + s dt:
//     vec_t -= B_{t + boff} vec_{t - dt}        (COL: B' instead of B)
// B_u = second matrix of block slot u.  Lane i < BS owns entry i of every block; the next block's
// row (column) is fetched while the current step waits for vec_{t - dt} (LDS, same wavefront).
template <int BS, bool COL>
__device__ __forceinline__ void bt_chain(double *vec, const double *s_blk, int i, int t0, int dt, int nsteps, int boff) {
  constexpr int BB = BS * BS;
  if (nsteps <= 0) return;
  auto fetch = [&](int t, double (&e)[BS]) {
    const double *B = s_blk + (size_t)(t + boff) * 2 * BB + BB;
#pragma unroll
    for (int k = 0; k < BS; k++) e[k] = COL ? B[k * BS + i] : B[i * BS + k];
  };
  double e0[BS], e1[BS];
  fetch(t0, e0);
  wave_lds_fence();
  auto step = [&](int sidx, const double (&ec)[BS], double (&en)[BS]) {
    const int t = t0 + sidx * dt;
    double pv[BS];
    const double own = vec[t * BS + i];
#pragma unroll
    for (int k = 0; k < BS; k++) pv[k] = vec[(t - dt) * BS + k];
    if (sidx + 1 < nsteps) fetch(t + dt, en);
    double a0 = own, a1 = 0.0;
#pragma unroll
    for (int k = 0; k < BS; k += 2) { a0 -= ec[k] * pv[k]; a1 -= ec[k + 1] * pv[k + 1]; }
    vec[t * BS + i] = a0 + a1;
    wave_lds_fence();
  };
  int sidx = 0;
  for (; sidx + 1 < nsteps; sidx += 2) { step(sidx, e0, e1); step(sidx + 1, e1, e0); }
  if (sidx < nsteps) step(sidx, e0, e1);
}

// One generic chunk (kind 1: rows walk the CSR arrays; kind 2: at most one core entry per item).
__device__ __forceinline__ void bt_generic_chunk(const BigArgs &a, const QpDev &d, const int (&dsc)[CH_STRIDE], int ch, int lane, bool chk,
                                                 const BtPtrs &bp, const double *As, const double *rho, const int *w, double *z, double *y,
                                                 const double *ls, const double *us, double *ge, const double *kinv, double *x,
                                                 const double *qs, double *sdy, double *sdx, double *prod, const double *s_xc,
                                                 double alpha, double sigma) {
  if (lane >= dsc[1]) return;
  const int kind = dsc[0];
      const int *rec = a.it + ((size_t)ch * 64 + lane) * 8;
      if (kind == 2) {
        // rows with at most one core entry (bound rows, pins): direct indices; every load is
        // unconditional on a clamped index so the chunk costs two memory round trips
        typedef int int4v __attribute__((ext_vector_type(4)));
        const int4v ra4 = *(const int4v *)rec, rb4 = *(const int4v *)(rec + 4);
        const int e = ra4.x, j = ra4.y, r0 = ra4.z, r1 = ra4.w, ep0 = rb4.x, ep1 = rb4.y, cc = rb4.z, cp = rb4.w;
        const int r0c = r0 >= 0 ? r0 : 0, r1c = r1 >= 0 ? r1 : r0c, ec = e >= 0 ? e : 0, jc = j >= 0 ? j : 0;
        const double a0 = As[cp >= 0 ? cp : 0];
        BtRow p{0.0, As[ep0 >= 0 ? ep0 : 0], rho[r0c], (double)w[r0c], z[r0c], y[r0c], ls[r0c], us[r0c]};
        BtRow q{0.0, As[ep1 >= 0 ? ep1 : 0], rho[r1c], (double)w[r1c], z[r1c], y[r1c], ls[r1c], us[r1c]};
        const double g = ge[ec], ki = e >= 0 ? kinv[ec] : 0.0, xo = x[jc], qj = qs[jc];
        if (cp >= 0) p.zc = a0 * s_xc[cc];
        if (r0 < 0 || e < 0) p.ae = 0.0;
        if (r1 < 0 || e < 0) q.ae = 0.0;
        if (r0 < 0) p.w = 0.0;
        if (r1 < 0) q.w = 0.0;
        const double rwp = p.w * p.rh, rwq = q.w * q.rh;
        const double xte = e >= 0 ? g - ki * (rwp * p.ae * p.zc + rwq * q.ae * q.zc) : 0.0;
        double zn, yn, dy, tq0 = 0.0, tq1 = 0.0;
        if (r0 >= 0) { tq0 = bt_row_step(p, alpha, xte, zn, yn, dy); z[r0] = zn; y[r0] = yn; if (chk) sdy[r0] = dy; }
        if (r1 >= 0) { tq1 = bt_row_step(q, alpha, xte, zn, yn, dy); z[r1] = zn; y[r1] = yn; if (chk) sdy[r1] = dy; }
        double gn = 0.0;
        if (e >= 0) {
          const double xn = alpha * xte + (1.0 - alpha) * xo;
          if (chk) sdx[j] = xn - xo;
          x[j] = xn;
          gn = ((sigma * xn - qj) + p.ae * tq0 + q.ae * tq1) * ki;
          ge[e] = gn;
        }
        if (cp >= 0) prod[cp] = a0 * (tq0 - rwp * p.ae * gn);
        return;
      }
      const int e = rec[0], r0 = rec[2], r1 = rec[3];
      BtRow p{0.0, 0.0, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0}, q = p;
      if (r0 >= 0) p = BtRow{big_row_core_dot(d, As, s_xc, r0), 0.0, rho[r0], (double)w[r0], z[r0], y[r0], ls[r0], us[r0]};
      if (r1 >= 0) q = BtRow{big_row_core_dot(d, As, s_xc, r1), 0.0, rho[r1], (double)w[r1], z[r1], y[r1], ls[r1], us[r1]};
      double g = 0.0, ki = 0.0, xte = 0.0;
      const double rwp = p.w * p.rh, rwq = q.w * q.rh;
      if (e >= 0) {
        if (r0 >= 0) p.ae = As[rec[4]];
        if (r1 >= 0) q.ae = As[rec[5]];
        g = ge[e]; ki = kinv[e];
        xte = g - ki * (rwp * p.ae * p.zc + rwq * q.ae * q.zc);
      }
      double zn, yn, dy, tq0 = 0.0, tq1 = 0.0;
      if (r0 >= 0) { tq0 = bt_row_step(p, alpha, xte, zn, yn, dy); z[r0] = zn; y[r0] = yn; if (chk) sdy[r0] = dy; }
      if (r1 >= 0) { tq1 = bt_row_step(q, alpha, xte, zn, yn, dy); z[r1] = zn; y[r1] = yn; if (chk) sdy[r1] = dy; }
      double gn = 0.0;
      if (e >= 0) {
        const int j = rec[1];
        const double xo = x[j], xn = alpha * xte + (1.0 - alpha) * xo;
        if (chk) sdx[j] = xn - xo;
        x[j] = xn;
        gn = ((sigma * xn - qs[j]) + p.ae * tq0 + q.ae * tq1) * ki;
        ge[e] = gn;
      }
      const double t0 = tq0 - rwp * p.ae * gn, t1 = tq1 - rwq * q.ae * gn;
      if (r0 >= 0)
        for (int s = d.Rp[r0]; s < d.Rp[r0 + 1]; s++)
          if (d.core_of[d.Rj[s]] >= 0) prod[d.Rpos[s]] = As[d.Rpos[s]] * t0;
      if (r1 >= 0)
        for (int s = d.Rp[r1]; s < d.Rp[r1 + 1]; s++)
          if (d.core_of[d.Rj[s]] >= 0) prod[d.Rpos[s]] = As[d.Rpos[s]] * t1;
}

template <int BS>
__global__ __launch_bounds__(BTT) void qp_admm_bt_kernel(BigArgs a) {
  const QpDev &d = a.d;
  const int b = d.list ? d.list[blockIdx.x + d.b0] : (int)blockIdx.x + d.b0, tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (b < 0 || (d.active && !d.active[b])) return;
  const int n = d.n, m = d.m, n_e = d.n_e, n_c = d.n_c, nb = a.bt_nb, mid = a.bt_mid;
  constexpr int BB = BS * BS;
  const int ncp = nb * BS;
  const bool use_part = a.use_part != 0;
  __shared__ double red[BTWV * 8];
  extern __shared__ double s_dyn[];
  double *s_blk = s_dyn;                       // [nb][2][BS][BS]: D_t^-1, E_t
  double *s_r = s_blk + (size_t)nb * 2 * BB;   // core right-hand side r; the forward sweep turns it into y in place
  double *s_xc = s_r + ncp;                    // z = D^-1 y; the backward sweep turns it into x~_C in place; then 64 zeros
  double *s_x = s_xc + ncp + 64;               // x_C (scaled iterate of the core variables)
  double *s_q = s_x + ncp;                     // q_C
  double *s_part = s_q + ncp;                  // partial column sums of the dense chunks
  int *s_cent = (int *)(s_part + a.npart);     // use_part: [ncp][4] column contributions; else: segment start, length
  int *s_dsc = s_cent + 4 * ncp;
  int *s_cflag = s_dsc + a.nchunks * CH_STRIDE;      // chunk flags (shared per-row constants)
  const double *As = d.As + (size_t)b * d.nnzA, *Ps = d.Ps + (size_t)b * d.nnzP;
  const double *qs = d.qs + (size_t)b * n, *ls = d.ls + (size_t)b * m, *us = d.us + (size_t)b * m;
  const double *rho = d.rho + (size_t)b * m, *kinv = d.kee_inv + (size_t)b * n_e;
  const double *Dg = d.D + (size_t)b * n, *Eg = d.E + (size_t)b * m;
  const int *w = d.w + (size_t)b * m;
  double *x = d.x + (size_t)b * n, *y = d.y + (size_t)b * m;     // scaled iterates; unscaled at the end
  double *ws = a.ws + (size_t)b * a.ws_stride;
  double *z = ws, *tp = z + m, *sdy = tp + m, *ge = sdy + m, *sdx = ge + n_e, *prod = sdx + n;
  const double cscale = d.cscale[b];
  // Scalar kernel arguments of the iteration loop through opaque copies, and the termination test / park path through a
  // fresh view of the argument segment (r03): rho, sigma, alpha, the four tolerances, max_iter and check arrive in ONE
  // 16-word scalar load, hipcc keeps the 16 words as one register tuple for as long as any of them lives, and -- scalar
  // registers being what this kernel spills -- the row loop reloaded all 16 words six times per chunk pair (96 of its
  // 150 v_readlane).  With the copies the tuple dies in the prologue; the test, every 25th iteration, loads what it
  // needs from the (constant) argument segment again.
  double alpha = a.alpha, sigma = a.sigma, rho_arg = a.rho;
  int max_iter = a.max_iter, check = a.check;
  asm volatile("" : "+s"(alpha), "+s"(sigma), "+s"(rho_arg), "+s"(max_iter), "+s"(check));
  // (BigArgs is the kernel's ONLY explicit parameter, passed by value: the argument segment starts with it.  A second
  // parameter, or one in front of it, would make this view read something else -- the static_assert behind the kernel
  // pins the signature at compile time.)
  const BigArgs *akp = (const BigArgs *)__builtin_amdgcn_kernarg_segment_ptr();
  // v shares the storage of t': the start point's t' is only read by the prologue below
  const BtPtrs bp{As, rho, ls, us, kinv, qs, w, x, y, z, sdy, sdx, ge, prod, s_xc, use_part ? s_part : nullptr, alpha, sigma,
                  tp, s_cflag, a.per_problem_rho ? d.rho_b[b] : rho_arg, a.ccon ? a.ccon[b] : 1.0,
                  1.0 / (a.per_problem_rho ? d.rho_b[b] : rho_arg)};

  {
    const double *blk = a.bt_blk + (size_t)b * a.bt_stride;
    for (size_t t = tid; t < (size_t)nb * 2 * BB; t += BTT) s_blk[t] = blk[t];
    for (int c = tid; c < 4 * ncp + 64 + a.npart; c += BTT) s_r[c] = 0.0;
    if (use_part) {
      for (int t = tid; t < 4 * ncp; t += BTT) s_cent[t] = a.cent[t];
    } else {
      for (int c = tid; c < ncp; c += BTT) {
        int p0 = 0, len = 0;
        if (c < n_c) { const int j = d.core_var[c]; p0 = d.Ap[j]; len = d.Ap[j + 1] - p0; }
        s_cent[c] = p0; s_cent[ncp + c] = len;
      }
    }
    for (int t = tid; t < a.nchunks * CH_STRIDE; t += BTT) s_dsc[t] = a.ch_desc[t];
    for (int t = tid; t < a.nchunks; t += BTT) s_cflag[t] = a.cflag ? a.cflag[(size_t)b * a.nchunks + t] : 0;
  }
  // a parked solve (time slicing, adaptive rho) resumes from its scaled x, y (left in place) and z, t', g_e and the
  // partial column sums (save area); when rho has changed meanwhile only x, y, z carry over and t, g_e, t' are
  // rebuilt with the new rho, exactly as a warm start does from its x, y
  const int it0 = a.slice > 0 ? d.prog[b] : 0;
  const bool rederive = it0 > 0 && a.adaptive && d.rflag[b] != 0;
  const bool restore = it0 > 0 && !rederive;
  const double *pz = d.sz + (size_t)b * m;
  if (restore) {
    // nothing ran on this problem's workspace since it parked (setup only runs when rho changes): z, g_e and the
    // products are where the loop left them; only the LDS state comes back
    for (int c = tid; c < n_c; c += BTT) { s_q[c] = qs[d.core_var[c]]; s_x[c] = x[d.core_var[c]]; }
  } else if (rederive || a.warm) {
    // OSQP-style warm start from the handle's previous (unscaled) solution in x / y:
    //   x_s = x / D,  y_s = c y / (E w),  z = A_s x_s,  t = w (rho z - y)
    if (!rederive) {
      for (int j = tid; j < n; j += BTT) x[j] = x[j] / Dg[j];
      for (int i = tid; i < m; i += BTT) y[i] = y[i] * cscale / (Eg[i] * (double)w[i]);
    }
    for (int j = tid; j < n; j += BTT) sdx[j] = 0.0;
    for (int i = tid; i < m; i += BTT) sdy[i] = 0.0;
    __syncthreads();
    for (int c = tid; c < n_c; c += BTT) { s_q[c] = qs[d.core_var[c]]; s_x[c] = x[d.core_var[c]]; }
    for (int i = tid; i < m; i += BTT) {
      double ax = 0.0;
      if (rederive) ax = pz[i];
      else for (int s2 = d.Rp[i]; s2 < d.Rp[i + 1]; s2++) ax += As[d.Rpos[s2]] * x[d.Rj[s2]];
      z[i] = ax;
      tp[i] = (double)w[i] * (rho[i] * ax - y[i]);
    }
    __syncthreads();
    for (int e = tid; e < n_e; e += BTT) {
      const int j = d.elim_var[e];
      double acc = sigma * x[j] - qs[j];
      for (int r = a.er_ptr[e]; r < a.er_ptr[e + 1]; r++) acc += As[a.row_epos[a.er_row[r]]] * tp[a.er_row[r]];
      ge[e] = acc * kinv[e];
    }
    __syncthreads();
    for (int e = tid; e < n_e; e += BTT)              // t' = t - rw a_e g_e on the rows of every eliminated variable
      for (int r = a.er_ptr[e]; r < a.er_ptr[e + 1]; r++) {
        const int i = a.er_row[r];
        tp[i] -= ((double)w[i] * rho[i]) * As[a.row_epos[i]] * ge[e];
      }
  } else {
    for (int j = tid; j < n; j += BTT) { x[j] = 0.0; sdx[j] = 0.0; }
    for (int i = tid; i < m; i += BTT) { z[i] = 0.0; y[i] = 0.0; tp[i] = 0.0; sdy[i] = 0.0; }
    __syncthreads();
    for (int c = tid; c < n_c; c += BTT) s_q[c] = qs[d.core_var[c]];
    for (int e = tid; e < n_e; e += BTT) {
      const double g = -qs[d.elim_var[e]] * kinv[e];
      ge[e] = g;
      for (int r = a.er_ptr[e]; r < a.er_ptr[e + 1]; r++) {
        const int i = a.er_row[r];
        tp[i] = -((double)w[i] * rho[i]) * As[a.row_epos[i]] * g;
      }
    }
  }
  __syncthreads();
  if (a.adaptive && tid == 0) { d.smask[b] = 0; d.rflag[b] = 0; }
  if (!restore)
    for (int j = tid; j < n; j += BTT)
      for (int p = d.Ap[j]; p < d.Ap[j + 1]; p++) prod[p] = As[p] * tp[d.Ai[p]];
  if (use_part && restore) {
    const double *pp = a.park_part + (size_t)b * a.npart;
    for (int t = tid; t < a.npart; t += BTT) s_part[t] = pp[t];
  } else if (use_part) {
    // partial sums of the start point: one dense chunk per wavefront pass
    for (int ch = wave; ch < a.nchunks; ch += BTWV) {
      const int *dsc = s_dsc + ch * CH_STRIDE;
      if (dsc[0] != 0) continue;
      for (int k = 0; k < dsc[2]; k++) {
        const double v = lane < dsc[1] ? As[dsc[5] + k * dsc[6] + lane] * tp[dsc[3] + lane] : 0.0;
        const double tot = bt_wave_sum63(v);
        if (lane == 63) s_part[dsc[14] + k] = tot;
      }
    }
  }
  __syncthreads();

  int status = 0, iter = 0;
  double pri = 0.0, dua = 0.0;
#ifdef SCO_STAMP
  // diagnostic build only: cycles per phase per wavefront (never compiled into the product)
  long long st_acc[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#define BSTAMP(k) { const long long now_ = __builtin_readcyclecounter(); st_acc[k] += now_ - st_t; st_t = now_; }
#else
#define BSTAMP(k)
#endif
  for (iter = it0 + 1; iter <= max_iter; iter++) {
    const bool chk = (check > 0 && iter % check == 0) || iter == max_iter;
    const bool first = iter == it0 + 1;       // dense rows: z, y from their arrays (start, resume); afterwards from v
    BSTAMP(11)
    // (1) core right-hand side r_c = sigma x_c - q_c + (A' t)_c
    if (use_part) {
      // at most four contributions per column: partial sums of dense chunks (LDS) and products of other rows
      for (int c = tid; c < n_c; c += BTT) {
        double v = 0.0;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int en = s_cent[4 * c + u];
          if (en >= 0) v += s_part[en];
          else if (en <= -2) v += prod[-(en + 2)];
        }
        s_r[c] = (sigma * s_x[c] - s_q[c]) + v;
      }
    } else {
      // 16 lanes per core column sum its CSC segment of products, four columns in flight per lane group
      for (int c4 = (tid >> 4) * 4; c4 < n_c; c4 += (BTT / 16) * 4) {
        double pv[4][8];
        int pp[4], ln[4];
#pragma unroll
        for (int u = 0; u < 4; u++) { pp[u] = s_cent[c4 + u] + (tid & 15); ln[u] = s_cent[ncp + c4 + u] - (tid & 15); }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
          for (int k = 0; k < 8; k++) pv[u][k] = prod[pp[u] + (16 * k < ln[u] ? 16 * k : 0)];     // unconditional loads
#pragma unroll
        for (int u = 0; u < 4; u++) {
#pragma unroll
          for (int k = 0; k < 8; k++) pv[u][k] = (16 * k < ln[u]) ? pv[u][k] : 0.0;
          double v = ((pv[u][0] + pv[u][1]) + (pv[u][2] + pv[u][3])) + ((pv[u][4] + pv[u][5]) + (pv[u][6] + pv[u][7]));
          for (int k = 128; k < ln[u]; k += 16) v += prod[pp[u] + k];
          v = bt_row16_sum(v);
          if ((tid & 15) == 15 && c4 + u < n_c) s_r[c4 + u] = (sigma * s_x[c4 + u] - s_q[c4 + u]) + v;
        }
      }
    }
    BSTAMP(0)
    __syncthreads();
    BSTAMP(1)
    // (3) twisted block solve, in place.  Forward: wavefront 0 runs the top chain y_t = r_t - E_t y_{t-1}
    // (t = 1 .. mid), wavefront 1 the bottom chain y_t = r_t - G_t y_{t+1} (t = nb-2 .. mid+1)
    if (wave == 0 && lane < BS) bt_chain<BS, false>(s_r, s_blk, lane, 1, 1, mid, 0);
    if (wave == 1 && lane < BS) bt_chain<BS, false>(s_r, s_blk, lane, nb - 2, -1, nb - 2 - mid, 1);
    BSTAMP(2)
    __syncthreads();
    BSTAMP(3)
    for (int c = tid; c < ncp; c += BTT) {
      const int t = c / BS, i = c % BS;
      const double *Dv = s_blk + (size_t)t * 2 * BB + i * BS;
      double acc = 0.0;
#pragma unroll
      for (int k = 0; k < BS; k++) acc += Dv[k] * s_r[t * BS + k];
      if (t == mid && mid + 1 < nb) {            // the middle block also takes the bottom chain: - H y_{mid+1}
        const double *Hv = s_blk + BB + i * BS;
#pragma unroll
        for (int k = 0; k < BS; k++) acc -= Hv[k] * s_r[(mid + 1) * BS + k];
      }
      s_xc[c] = acc;
    }
    __syncthreads();
    BSTAMP(4)
    // backward from the middle block outwards: x_t = z_t - E_{t+1}' x_{t+1} (t = mid-1 .. 0) and
    // x_t = z_t - G_{t-1}' x_{t-1} (t = mid+1 .. nb-1)
    if (wave == 0 && lane < BS) bt_chain<BS, true>(s_xc, s_blk, lane, mid - 1, -1, mid, 1);
    if (wave == 1 && lane < BS) bt_chain<BS, true>(s_xc, s_blk, lane, mid + 1, 1, nb - 1 - mid, 0);
    BSTAMP(5)
    __syncthreads();
    BSTAMP(6)
    // (Y) rows in wavefront-sized chunks, two dense chunks in flight per wavefront
    for (int ch = wave; ch < a.nchunks; ch += BTWV) {
      int dsc[CH_STRIDE], dsb[CH_STRIDE];          // wave-uniform: descriptor words in SGPRs
#pragma unroll
      for (int k = 0; k < CH_STRIDE; k++) dsc[k] = __builtin_amdgcn_readfirstlane(s_dsc[ch * CH_STRIDE + k]);
      if (dsc[0] == 0 && a.bt_triple && !first && use_part && dsc[2] == BS && ch + 2 * BTWV < a.nchunks) {
        // three full-width chunks with known per-row constants: all their loads before the arithmetic of the first
        const int chb = ch + BTWV, chc = ch + 2 * BTWV;
        int dsc2[CH_STRIDE];
#pragma unroll
        for (int k = 0; k < CH_STRIDE; k++) dsb[k] = __builtin_amdgcn_readfirstlane(s_dsc[chb * CH_STRIDE + k]);
#pragma unroll
        for (int k = 0; k < CH_STRIDE; k++) dsc2[k] = __builtin_amdgcn_readfirstlane(s_dsc[chc * CH_STRIDE + k]);
        const int fa = __builtin_amdgcn_readfirstlane(s_cflag[ch]), fb = __builtin_amdgcn_readfirstlane(s_cflag[chb]),
                  fc = __builtin_amdgcn_readfirstlane(s_cflag[chc]);
        if (dsb[0] == 0 && dsc2[0] == 0 && dsb[2] == BS && dsc2[2] == BS && bt_lean_ok(fa, dsc[9] >= 0) &&
            bt_lean_ok(fb, dsb[9] >= 0) && bt_lean_ok(fc, dsc2[9] >= 0)) {
          BtLean<BS> LA, LB, LC;
          bt_lean_load<BS>(dsc, fa, lane, bp, LA);
          bt_lean_load<BS>(dsb, fb, lane, bp, LB);
          bt_lean_load<BS>(dsc2, fc, lane, bp, LC);
          bt_lean_compute<BS>(s_dsc, ch, lane, chk, bp, LA);
          bt_lean_compute<BS>(s_dsc, chb, lane, chk, bp, LB);
          bt_lean_compute<BS>(s_dsc, chc, lane, chk, bp, LC);
          ch = chc;
          BSTAMP(7)
          continue;
        }
      }
      if (dsc[0] == 0) {
        const int ncols = dsc[2], chb = ch + BTWV, cha = ch;
        bool hasB = false;
        if (chb < a.nchunks) {
#pragma unroll
          for (int k = 0; k < CH_STRIDE; k++) dsb[k] = __builtin_amdgcn_readfirstlane(s_dsc[chb * CH_STRIDE + k]);
          if (dsb[0] == 0 && (dsb[2] + 3) / 4 == (ncols + 3) / 4) { hasB = true; ch = chb; }
        }
        if (ncols == BS && (!hasB || dsb[2] == BS)) bt_dense_pair<BS, true>(dsc, dsb, cha, chb, hasB, lane, chk, first, bp);
        else if (ncols <= 4) bt_dense_pair<4>(dsc, dsb, cha, chb, hasB, lane, chk, first, bp);
        else if (ncols <= 8) bt_dense_pair<8>(dsc, dsb, cha, chb, hasB, lane, chk, first, bp);
        else if (ncols <= 12) bt_dense_pair<12>(dsc, dsb, cha, chb, hasB, lane, chk, first, bp);
        else bt_dense_pair<16>(dsc, dsb, cha, chb, hasB, lane, chk, first, bp);
        BSTAMP(7)
        continue;
      }
      BSTAMP(7)
      bt_generic_chunk(a, d, dsc, ch, lane, chk, bp, As, rho, w, z, y, ls, us, ge, kinv, x, qs, sdy, sdx, prod, s_xc, alpha, sigma);
      BSTAMP(12)
    }
    BSTAMP(7)
    for (int c = tid; c < n_c; c += BTT) {
      const double xo = s_x[c], xn = alpha * s_xc[c] + (1.0 - alpha) * xo;
      s_x[c] = xn;
      if (chk) { const int j = d.core_var[c]; sdx[j] = xn - xo; x[j] = xn; }
    }
    BSTAMP(8)
    __syncthreads();
    BSTAMP(9)
    if (!chk) continue;
    double rho_new = 0.0;
    {
      const BigArgs *ak = akp;
      asm volatile("" : "+s"(ak));                 // the arguments as the segment holds them, loaded here
      BigChk ck{As, Ps, qs, ls, us, Dg, Eg, x, y, z, sdx, sdy, w, cscale};
      status = big_check<BTT>(*ak, ck, iter, red, pri, dua);
      if (!status && ak->adaptive && iter % ak->ad_interval == 0 && iter < max_iter) {
        const double rho_b = d.rho_b[b], est = big_rho_estimate<BTT>(*ak, ck, red, rho_b);
        if (est > rho_b * ak->ad_tol || est < rho_b / ak->ad_tol) rho_new = est;
      }
    }
    __syncthreads();
    BSTAMP(10)
    if (status) break;
    if (iter < max_iter && (rho_new > 0.0 || (a.slice > 0 && iter == it0 + a.slice))) {
      // rho must change (setup + factorisation run again, then the solve resumes) or the slice is used up: park.
      // x, y stay in place (scaled) and so does the workspace (z, g_e, products) unless rho changes: then setup
      // uses the front of the workspace as scratch, so z goes to the save area; the LDS partial sums always do
      double *qz = d.sz + (size_t)b * m;
      for (int i = tid; i < m; i += BTT) qz[i] = z[i];
      if (use_part) {
        double *pp = a.park_part + (size_t)b * a.npart;
        for (int t = tid; t < a.npart; t += BTT) pp[t] = s_part[t];
      }
      if (tid == 0) {
        d.prog[b] = iter; d.status[b] = 0; d.iters[b] = iter;
        if (rho_new > 0.0) { d.rho_b[b] = rho_new; d.rflag[b] = 1; d.smask[b] = 1; d.nupd[b] += 1; }
      }
      return;
    }
  }
  if (a.slice > 0 && tid == 0) d.prog[b] = 0;
  if (!status) status = SCO_QP_MAX_ITER_REACHED;
  if (iter > max_iter) iter = max_iter;
#ifdef SCO_STAMP
  if (lane == 0 && b == 0 && a.stamp) {
    for (int k = 0; k < 14; k++) a.stamp[wave * 16 + k] = (double)st_acc[k];
    a.stamp[wave * 16 + 15] = (double)iter;
  }
#endif
  __syncthreads();
  {
    const double cinv = 1.0 / cscale;
    for (int j = tid; j < n; j += BTT) x[j] = Dg[j] * x[j];
    for (int i = tid; i < m; i += BTT) y[i] = cinv * Eg[i] * y[i] * (double)w[i];
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

#ifdef SCO_STAMP
static double *sco_debug_stamp_bt_ptr = nullptr;
extern "C" int sco_debug_stamps_bt(double *out) {      // diagnostic build only
  if (!sco_debug_stamp_bt_ptr) return -1;
  return hipMemcpy(out, sco_debug_stamp_bt_ptr, 256 * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
int bt_upload(const BtHost &th, int batch, std::vector<void *> &allocs, BtDev &td) {
  int rc;
  if ((rc = upb(allocs, th.ch_desc, &td.ch_desc))) return rc;
  if ((rc = upb(allocs, th.it, &td.it))) return rc;
  if ((rc = upb(allocs, th.cent, &td.cent))) return rc;
  void *p = nullptr;
  SCO_HIP(hipMalloc(&p, (size_t)batch * th.blk_doubles * sizeof(double)));
  allocs.push_back(p);
  td.blk = (double *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * th.ws_doubles * sizeof(double)));
  SCO_HIP(hipMemset(p, 0, (size_t)batch * th.ws_doubles * sizeof(double)));
  allocs.push_back(p);
  td.ws = (double *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * std::max(th.npart, 1) * sizeof(double)));
  allocs.push_back(p);
  td.park_part = (double *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * std::max(th.nchunks, 1) * sizeof(int)));
  SCO_HIP(hipMemset(p, 0, (size_t)batch * std::max(th.nchunks, 1) * sizeof(int)));
  allocs.push_back(p);
  td.cflag = (int *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * sizeof(double)));
  allocs.push_back(p);
  td.ccon = (double *)p;
  td.mf_id = td.mf_h0 = td.mf_h1 = nullptr;
  if (th.use_mfma) {
    if ((rc = upb(allocs, th.mf_id, &td.mf_id))) return rc;
    if ((rc = upb(allocs, th.mf_h0, &td.mf_h0))) return rc;
    if ((rc = upb(allocs, th.mf_h1, &td.mf_h1))) return rc;
  }
  return SCO_OK;
}

template <int BS>
static int bt_launch_bs(const BigArgs &bs, const BigArgs &ba, int batch, size_t lds, hipStream_t st, hipEvent_t ev_mid, hipEvent_t mid2) {
  // hipFuncSetAttribute applies to the current device only
  static bool attr_done[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  dev_ &= 63;
  if (!attr_done[dev_]) {
    SCO_HIP(hipFuncSetAttribute((const void *)qp_admm_bt_kernel<BS>, hipFuncAttributeMaxDynamicSharedMemorySize, 159 * 1024));
    attr_done[dev_] = true;
  }
  hipLaunchKernelGGL(qp_bt_factor_kernel<BS>, dim3(batch), dim3(256), 0, st, bs);
  SCO_HIP(hipGetLastError());
  if (ev_mid) SCO_HIP(hipEventRecord(ev_mid, st));
  if (mid2) SCO_HIP(hipEventRecord(mid2, st));
  hipLaunchKernelGGL(qp_admm_bt_kernel<BS>, dim3(batch), dim3(BTT), lds, st, ba);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}

// qp_admm_bt_kernel re-reads its arguments through the kernel-argument segment pointer (see there): that is only right
// while BigArgs is its one and only parameter
static_assert(std::is_same<decltype(&qp_admm_bt_kernel<12>), void (*)(BigArgs)>::value,
              "qp_admm_bt_kernel must take exactly one parameter, BigArgs by value: it views the kernel-argument segment as a BigArgs");

int big_launch(const AdmmArgs &a, const int *setup_mask, int scaling, const int *Pp, const int *Pi, const BigHost &bh,
               const BigDev &bd, const BtHost *th, const BtDev *td, hipStream_t st, hipEvent_t ev_mid, hipEvent_t mid2) {
  BigArgs ba;
  ba.stamp = nullptr;
#ifdef SCO_STAMP
  {
    static double *g_stamp = nullptr;
    if (!g_stamp) { SCO_HIP(hipMalloc((void **)&g_stamp, 256 * sizeof(double))); SCO_HIP(hipMemset(g_stamp, 0, 256 * sizeof(double))); }
    ba.stamp = g_stamp; sco_debug_stamp_bt_ptr = g_stamp;
  }
#endif
  ba.bt_bs = 0; ba.bt_nb = 0; ba.bt_mid = 0; ba.bt_blk = nullptr; ba.bt_stride = 0; ba.nchunks = 0; ba.bt_triple = 0;
  ba.ch_desc = ba.it = ba.cent = nullptr; ba.npart = 0; ba.use_part = 0; ba.cflag = nullptr; ba.ccon = nullptr;
  ba.mf_id = ba.mf_h0 = ba.mf_h1 = nullptr;
  ba.d = a.d; ba.Pp = Pp; ba.Pi = Pi;
  ba.row_elim = bd.row_elim; ba.row_epos = bd.row_epos; ba.er_ptr = bd.er_ptr; ba.er_row = bd.er_row;
  ba.free_rows = bd.free_rows; ba.pc_ptr = bd.pc_ptr; ba.pc_pos = bd.pc_pos; ba.pc_core = bd.pc_core;
  ba.n_free = (int)bh.free_rows.size();
  ba.ws = bd.ws; ba.ws_stride = bh.ws_doubles;
  ba.rho = a.rho; ba.sigma = a.sigma; ba.alpha = a.alpha; ba.eps_abs = a.eps_abs; ba.eps_rel = a.eps_rel;
  ba.eps_prim_inf = a.eps_prim_inf; ba.eps_dual_inf = a.eps_dual_inf;
  ba.max_iter = a.max_iter; ba.check = a.check; ba.scaling = scaling; ba.warm = a.warm;
  if (th) {
    ba.bt_bs = th->bs; ba.bt_nb = th->nb; ba.bt_mid = th->nb >= 4 ? th->nb / 2 : th->nb - 1; ba.bt_blk = td->blk; ba.bt_stride = th->blk_doubles;
    ba.ch_desc = td->ch_desc; ba.it = td->it; ba.cent = td->cent; ba.nchunks = th->nchunks;
    ba.npart = th->npart; ba.use_part = th->use_part ? 1 : 0;
    { const char *te = getenv("SCO_QP_BT_TRIPLE"); ba.bt_triple = !(te && atoi(te) == 0); }
    ba.ws = td->ws; ba.ws_stride = th->ws_doubles;
    ba.cflag = td->cflag; ba.ccon = td->ccon;
    ba.mf_id = td->mf_id; ba.mf_h0 = td->mf_h0; ba.mf_h1 = td->mf_h1;
  }
  ba.slice = th ? a.slice : 0; ba.adaptive = th ? a.adaptive : 0; ba.ad_interval = a.ad_interval; ba.ad_tol = a.ad_tol;
  ba.per_problem_rho = ba.adaptive; ba.park_part = th ? td->park_part : nullptr;
  BigArgs bs = ba; bs.d.active = setup_mask;          // setup + factorisation: the problems that need them
  const int nwg = (th && a.d.nb > 0) ? a.d.nb : a.d.batch;       // launch window / index list: structured form only
  hipLaunchKernelGGL(qp_setup_big_kernel, dim3(nwg), dim3(BT), 0, st, bs);
  SCO_HIP(hipGetLastError());
  if (th) {
    switch (th->bs) {
      case 4: return bt_launch_bs<4>(bs, ba, nwg, th->lds_bytes, st, ev_mid, mid2);
      case 8: return bt_launch_bs<8>(bs, ba, nwg, th->lds_bytes, st, ev_mid, mid2);
      case 12: return bt_launch_bs<12>(bs, ba, nwg, th->lds_bytes, st, ev_mid, mid2);
      default: return bt_launch_bs<16>(bs, ba, nwg, th->lds_bytes, st, ev_mid, mid2);
    }
  }
  if (ev_mid) SCO_HIP(hipEventRecord(ev_mid, st));
  if (mid2) SCO_HIP(hipEventRecord(mid2, st));
  hipLaunchKernelGGL(qp_admm_big_kernel, dim3(a.d.batch), dim3(BT), 0, st, ba);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}
