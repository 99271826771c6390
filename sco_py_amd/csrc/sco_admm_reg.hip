// sco_admm_reg.hip -- register-resident ADMM kernel for gfx950 (MI355X).
//
// Same mathematics and iterate sequence as qp_admm_kernel (sco_qp.hip) and
// qp_admm_fast_kernel (sco_admm_fast.hip): OSQP's ADMM, the third-party call behind
// /root/reference/sco_py/sco_osqp/osqp_utils.py:216.  This is the variant tuned for
// the instruction-issue bound the LDS-operator kernel ran into (profiles/r01_v3_*):
//
//   * 512 threads per problem (8 wavefronts, 2 per SIMD), one problem per CU,
//     the whole ADMM solve in one launch;
//   * the W tile (TR x TC doubles, thread (gi, gj) of a 32 x 16 grid) and every
//     GATHER OFFSET of the thread's sparse work (CW column entries, 2 x RW row
//     entries, PX coupling entries -- K_CE by core for core owners, by eliminated
//     variable for the owner of that variable's column; the host assigns core
//     ownership so that no thread holds both roles) are loaded ONCE into registers,
//     already scaled to byte offsets and packed two per VGPR;
//   * sparse VALUES stay in LDS as sliced-ELL images (entry k of lane l of slice s
//     at base[s] + 64 k + l: conflict-free, read with immediate offsets);
//   * slots past an item's length point at an element that is always 0.0, so the
//     dot products are straight-line `value load, gather, fma` with no masking
//     selects, no index loads and no address arithmetic;
//   * the vectors that cross threads (x~, t, eliminated rhs, core rhs and solution,
//     the 16-way partial sums of the W mat-vec) sit in static LDS (44 KB) at
//     compile-time addresses;
//   * kernel arguments are a slim struct (no SGPR spills in the loop);
//   * r03: the solve parks and resumes like the other on-chip kernels (time slicing, adaptive rho).
#include "sco_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#define RT 512
#define RWV (RT / 64)
#define RGJ 16
#define RGI (RT / RGJ)
#define CAP_N 512
#define CAP_M 1024
#define CAP_NC 160

// --------------------------------------------------------------------------
// host: per-thread programs
// --------------------------------------------------------------------------
void build_sell(int nitems, const std::vector<int> &ptr, const std::vector<int> &idx,
                const std::vector<int> &src, SellHost &out);   // sco_admm_fast.hip

bool reg_plan_build(const QpPlan &pl, int CW, int RW, int PX, RegHost &rh) {
  // one spare element per gathered vector serves as the always-zero target of padded slots
  if (pl.n >= CAP_N || pl.m >= CAP_M || pl.n_c >= CAP_NC || pl.n_e >= CAP_N || pl.nnzA >= 65536) return false;
  rh.CW = CW; rh.RW = RW; rh.PX = PX;
  {
    std::vector<int> ident(std::max(pl.nnzA, pl.ncpl) + 1);
    for (size_t i = 0; i < ident.size(); i++) ident[i] = (int)i;
    build_sell(pl.n, pl.Ap, pl.Ai, ident, rh.Ac);
    build_sell(pl.m, pl.Rp, pl.Rj, pl.Rpos, rh.Ar);
    std::vector<int> eidx(pl.ncpl), srck(pl.ncpl);
    for (int t = 0; t < pl.ncpl; t++) { eidx[t] = pl.pair_elim[pl.a_pair[t]]; srck[t] = pl.a_pair[t]; }
    build_sell(pl.n_c, pl.a_ptr, eidx, srck, rh.Ca);
    build_sell(pl.n_e, pl.e_ptr, pl.pair_core, ident, rh.Ce);
    // room for the unrolled reads that run past the last slice
    rh.lds_bytes = 8 * ((size_t)rh.Ac.total + rh.Ar.total + rh.Ca.total + rh.Ce.total + 64 * 16);
    if (rh.lds_bytes + 44032 > 160 * 1024) return false;
  }
  rh.TR = std::max(1, (pl.n_c + RGI - 1) / RGI);
  rh.TC = 2 * rh.TR;
  if (rh.TR == 5 && pl.n_c <= RGJ * 9) rh.TC = 9;
  const int slots = CW + 2 * RW + PX;
  rh.off.assign((size_t)slots * RT, 0);
  rh.role.assign((size_t)8 * RT, -1);     // [0] core owned, [1] core variable, [2] elim index,
                                          // [3] col base, [4] row0 base, [5] row1 base, [6] pair base (LDS element index)
  // padded slots gather the spare zero element of their vector
  for (int t = 0; t < RT; t++) {
    for (int k = 0; k < CW; k++) rh.off[(size_t)k * RT + t] = (unsigned short)(8 * pl.m);
    for (int k = 0; k < 2 * RW; k++) rh.off[(size_t)(CW + k) * RT + t] = (unsigned short)(8 * pl.n);
    for (int k = 0; k < PX; k++) rh.off[(size_t)(CW + 2 * RW + k) * RT + t] = (unsigned short)(8 * pl.n_c);
    rh.role[(size_t)3 * RT + t] = 0; rh.role[(size_t)4 * RT + t] = 0; rh.role[(size_t)5 * RT + t] = 0;
    rh.role[(size_t)6 * RT + t] = 0;
  }
  // columns: thread j
  for (int j = 0; j < pl.n; j++) {
    const int w = pl.Ap[j + 1] - pl.Ap[j];
    if (w > CW) return false;
    for (int k = 0; k < w; k++)
      rh.off[(size_t)k * RT + j] = (unsigned short)(8 * pl.Ai[pl.Ap[j] + k]);      // byte offset into t / w*y
    rh.role[(size_t)3 * RT + j] = rh.Ac.base[j / 64] + (j % 64);
  }
  // rows: thread i % RT, slot i / RT
  for (int i = 0; i < pl.m; i++) {
    const int w = pl.Rp[i + 1] - pl.Rp[i];
    if (w > RW) return false;
    const int t = i % RT, q = i / RT;
    for (int k = 0; k < w; k++)
      rh.off[(size_t)(CW + q * RW + k) * RT + t] = (unsigned short)(8 * pl.Rj[pl.Rp[i] + k]);   // into x~ / x
    rh.role[(size_t)(4 + q) * RT + t] = rh.Ar.base[i / 64] + (i % 64);
  }
  // eliminated variables: owner = the thread of their column
  std::vector<char> elim_owner(RT, 0);
  for (int e = 0; e < pl.n_e; e++) {
    const int t = pl.elim_var[e];
    const int w = pl.e_ptr[e + 1] - pl.e_ptr[e];
    if (w > PX) return false;
    elim_owner[t] = 1;
    rh.role[(size_t)2 * RT + t] = e;
    for (int k = 0; k < w; k++)
      rh.off[(size_t)(CW + 2 * RW + k) * RT + t] = (unsigned short)(8 * pl.pair_core[pl.e_ptr[e] + k]);   // into x_C
    rh.role[(size_t)6 * RT + t] = rh.Ac.total + rh.Ar.total + rh.Ca.total + rh.Ce.base[e / 64] + (e % 64);
  }
  // core variables: prefer threads without a column, never an eliminated-variable owner
  {
    std::vector<int> cand;
    for (int t = RT - 1; t >= pl.n; t--) cand.push_back(t);
    for (int t = 0; t < pl.n; t++) if (!elim_owner[t]) cand.push_back(t);
    if ((int)cand.size() < pl.n_c) return false;
    for (int c = 0; c < pl.n_c; c++) {
      const int t = cand[c];
      const int w = pl.a_ptr[c + 1] - pl.a_ptr[c];
      if (w > PX) return false;
      rh.role[t] = c;
      rh.role[(size_t)RT + t] = pl.core_var[c];
      for (int k = 0; k < PX; k++)       // padded slots of a core owner gather the zero element of g_E
        rh.off[(size_t)(CW + 2 * RW + k) * RT + t] = (unsigned short)(8 * pl.n_e);
      for (int k = 0; k < w; k++) {
        const int pair = pl.a_pair[pl.a_ptr[c] + k];
        rh.off[(size_t)(CW + 2 * RW + k) * RT + t] = (unsigned short)(8 * pl.pair_elim[pair]);   // into g_E
      }
      rh.role[(size_t)6 * RT + t] = rh.Ac.total + rh.Ar.total + rh.Ca.base[c / 64] + (c % 64);
    }
  }
  return true;
}

// --------------------------------------------------------------------------
// device
// --------------------------------------------------------------------------
struct RegArgs {
  // sizes
  int n, m, n_e, n_c, nnzA, ncpl, nnzP, max_iter, check;
  double sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  // per-thread programs + sliced-ELL sources of the four value images
  const unsigned short *off; const int *role;
  const int *srcAc, *srcAr, *srcCa, *srcCe;
  int totAc, totAr, totCa, totCe;
  // per-problem data
  const double *As, *cpl, *W, *qs, *kee_inv, *ls, *us, *rho, *cscale, *Ps, *D, *E;
  const int *w, *core_of, *Fp, *Fi, *Fpos, *active;
  double *x, *y, *resid;
  int *status, *iters;
  // park / resume (time slicing, adaptive rho; r03 -- the same protocol as qp_admm_fast_kernel): the loop carries x, z, y
  // and t = w (rho z - y); a parked solve leaves the first three in sx / sz / sy and its iteration count in prog
  int slice, adaptive, ad_interval;
  double ad_tol;
  int *prog, *rflag, *smask, *nupd;
  double *sx, *sz, *sy, *rho_b;
};

__device__ __forceinline__ double rwmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double rwsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
template <int NR, bool IS_MAX>
__device__ __forceinline__ void rblock_reduce(double (&v)[NR], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = IS_MAX ? rwmax(v[k]) : rwsum(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NR; k++) red[wv * NR + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NR; k++) {
    double r = red[k];
#pragma unroll
    for (int w = 1; w < RWV; w++) r = IS_MAX ? fmax(r, red[w * NR + k]) : r + red[w * NR + k];
    v[k] = r;
  }
}

__device__ __forceinline__ double gat(const double *base, unsigned int byte_off) {
  return *(const double *)((const char *)base + byte_off);
}

// N-entry dot product: values V[64 k] (sliced-ELL image in LDS, immediate offsets),
// gathered operand at byte offset o[k] (two 16-bit offsets per register).
// N-entry dot product: values V[64 k] (sliced-ELL image in LDS, immediate offsets),
// gathered operand at byte offset o[k] (two 16-bit offsets per register).  The empty
// asm keeps the offsets PACKED: without it the compiler hoists the 38 unpacked
// offsets out of the ADMM loop, runs out of VGPRs and reloads them from scratch
// every iteration (profiles/r01_v4_*).
template <int N>
__device__ __forceinline__ double reg_dot(const double *V, unsigned int (&o)[(N + 1) / 2], const double *vec) {
  double val[N], g[N];
#pragma unroll
  for (int h = 0; h < (N + 1) / 2; h++) asm volatile("" : "+v"(o[h]));
#pragma unroll
  for (int k = 0; k < N; k++) {
    val[k] = V[64 * k];
    g[k] = gat(vec, (k & 1) ? (o[k / 2] >> 16) : (o[k / 2] & 0xffffu));
  }
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < N; k++) acc += val[k] * g[k];
  return acc;
}

template <int TR, int TC, int CW, int RW, int PX>
__global__ __launch_bounds__(RT) void qp_admm_reg_kernel(RegArgs a) {
  constexpr int PS = RGI * TR + 1;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (a.active && !a.active[b]) return;
  const int n = a.n, m = a.m, n_e = a.n_e, n_c = a.n_c;
  const int it0 = a.slice > 0 ? a.prog[b] : 0;         // > 0: a parked solve resumes from its scaled x, z, y

  __shared__ double s_xt[CAP_N];          // x~ (and rhs before the solve)
  __shared__ double s_tv[CAP_M];          // t = w (rho z - y)
  __shared__ double s_ge[CAP_N];          // K_EE^-1 rhs_E
  __shared__ double s_xc[CAP_NC];         // x~_C
  __shared__ double s_rv[CAP_NC];         // core right-hand side (zero padded)
  __shared__ double s_scr[2 * CAP_N + 2 * CAP_M];   // partial sums / check scratch
  __shared__ double s_red[RWV * 8];
  double *sx = s_scr, *swy = s_scr + CAP_N, *sdy = swy + CAP_M, *sdx = sdy + CAP_M;

  // ---- prologue: value images into LDS, gather offsets into registers -------------
  extern __shared__ double s_val[];
  {
    const double *gAs = a.As + (size_t)b * a.nnzA;
    const double *gcp = a.cpl + (size_t)b * a.ncpl;
    double *V = s_val;
    for (int p = tid; p < a.totAc; p += RT) { const int s = a.srcAc[p]; V[p] = s >= 0 ? gAs[s] : 0.0; }
    V += a.totAc;
    for (int p = tid; p < a.totAr; p += RT) { const int s = a.srcAr[p]; V[p] = s >= 0 ? gAs[s] : 0.0; }
    V += a.totAr;
    for (int p = tid; p < a.totCa; p += RT) { const int s = a.srcCa[p]; V[p] = s >= 0 ? gcp[s] : 0.0; }
    V += a.totCa;
    for (int p = tid; p < a.totCe; p += RT) { const int s = a.srcCe[p]; V[p] = s >= 0 ? gcp[s] : 0.0; }
    V += a.totCe;
    for (int p = tid; p < 64 * 16; p += RT) V[p] = 0.0;      // slack behind the last slice
  }
  auto pack = [&](int slot) -> unsigned int {
    return (unsigned int)a.off[(size_t)slot * RT + tid] | ((unsigned int)a.off[(size_t)(slot + 1) * RT + tid] << 16);
  };
  unsigned int co[CW / 2], ro[2][RW / 2], po[PX / 2];
#pragma unroll
  for (int k = 0; k < CW / 2; k++) co[k] = pack(2 * k);
#pragma unroll
  for (int q = 0; q < 2; q++)
#pragma unroll
    for (int k = 0; k < RW / 2; k++) ro[q][k] = pack(CW + q * RW + 2 * k);
#pragma unroll
  for (int k = 0; k < PX / 2; k++) po[k] = pack(CW + 2 * RW + 2 * k);
  const double *vc = s_val + a.role[(size_t)3 * RT + tid];
  const double *vr0 = s_val + a.totAc + a.role[(size_t)4 * RT + tid];
  const double *vr1 = s_val + a.totAc + a.role[(size_t)5 * RT + tid];
  const double *vp = s_val + a.role[(size_t)6 * RT + tid];
  const int gi = tid / RGJ, gj = tid % RGJ;
  double wreg[TR][TC];
  {
    const double *W = a.W + (size_t)b * n_c * n_c;
#pragma unroll
    for (int rr = 0; rr < TR; rr++)
#pragma unroll
      for (int cc = 0; cc < TC; cc++) {
        const int row = gi * TR + rr, col = gj * TC + cc;
        wreg[rr][cc] = (row < n_c && col < n_c) ? W[(size_t)row * n_c + col] : 0.0;
      }
  }
  // column state
  const bool colon = tid < n;
  double qj = 0.0, xj = 0.0, kinv = 0.0;
  int cj = -1;
  const int ej = a.role[(size_t)2 * RT + tid];
  if (colon) {
    qj = a.qs[(size_t)b * n + tid]; cj = a.core_of[tid];
    if (it0 > 0) xj = a.sx[(size_t)b * n + tid];
    if (ej >= 0) kinv = a.kee_inv[(size_t)b * n_e + ej];
  }
  // core-owner state
  const int cown = a.role[tid];
  const int cvar = cown >= 0 ? a.role[(size_t)RT + tid] : 0;
  // row state
  double r_ls[2], r_us[2], r_rho[2], r_rinv[2], r_z[2], r_y[2], r_w[2];
  bool r_on[2];
#pragma unroll
  for (int q = 0; q < 2; q++) {
    const int i = tid + q * RT;
    r_on[q] = i < m;
    r_ls[q] = r_us[q] = r_z[q] = r_y[q] = 0.0; r_rho[q] = r_rinv[q] = r_w[q] = 1.0;
    if (r_on[q]) {
      r_ls[q] = a.ls[(size_t)b * m + i]; r_us[q] = a.us[(size_t)b * m + i];
      r_rho[q] = a.rho[(size_t)b * m + i]; r_rinv[q] = 1.0 / r_rho[q];
      r_w[q] = (double)a.w[(size_t)b * m + i];
      if (it0 > 0) { r_z[q] = a.sz[(size_t)b * m + i]; r_y[q] = a.sy[(size_t)b * m + i]; }
    }
  }
  for (int i = tid; i < CAP_M; i += RT) s_tv[i] = 0.0;
  for (int i = tid; i < CAP_NC; i += RT) { s_rv[i] = 0.0; s_xc[i] = 0.0; }
  for (int i = tid; i < CAP_N; i += RT) { s_ge[i] = 0.0; s_xt[i] = 0.0; }
  const double cscale = a.cscale[b];
  const double alpha = a.alpha, sigma = a.sigma;
  if (a.adaptive && tid == 0) { a.smask[b] = 0; a.rflag[b] = 0; }
  __syncthreads();
  if (it0 > 0) {                      // t with the rho in force now (it may have changed while the solve was parked)
#pragma unroll
    for (int q = 0; q < 2; q++)
      if (r_on[q]) s_tv[tid + q * RT] = r_w[q] * (r_rho[q] * r_z[q] - r_y[q]);
    __syncthreads();
  }

  int status = 0, iter = 0;
  double pri = 0.0, dua = 0.0;
  for (iter = it0 + 1; iter <= a.max_iter; iter++) {
    const bool chk = (a.check > 0 && iter % a.check == 0) || iter == a.max_iter;
    // (1) rhs_j = sigma x_j - q_j + sum_i A_ij t_i
    double gev = 0.0;
    if (colon) {
      double v = reg_dot<CW>(vc, co, s_tv);
      v += sigma * xj - qj;
      s_xt[tid] = v;
      if (ej >= 0) { gev = v * kinv; s_ge[ej] = gev; }
    }
    __syncthreads();
    // (2) core rhs  r = rhs_C - K_CE K_EE^-1 rhs_E
    if (cown >= 0) s_rv[cown] = s_xt[cvar] - reg_dot<PX>(vp, po, s_ge);
    __syncthreads();
    // (3a) register-tile mat-vec
    {
      double rr_[TC];
#pragma unroll
      for (int cc = 0; cc < TC; cc++) rr_[cc] = s_rv[gj * TC + cc];
#pragma unroll
      for (int rr = 0; rr < TR; rr++) {
        double acc = 0.0;
#pragma unroll
        for (int cc = 0; cc < TC; cc++) acc += wreg[rr][cc] * rr_[cc];
        s_scr[gj * PS + gi * TR + rr] = acc;
      }
    }
    __syncthreads();
    // (3b) x~_C
    if (tid < n_c) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < RGJ; g++) v += s_scr[g * PS + tid];
      s_xc[tid] = v;
    }
    __syncthreads();
    // (4) x~ : eliminated variables by back-substitution, core variables copied
    if (colon) s_xt[tid] = (ej >= 0) ? (gev - kinv * reg_dot<PX>(vp, po, s_xc)) : s_xc[cj];
    __syncthreads();
    // (5) z~ = A x~, z / y / x updates, next t
#pragma unroll
    for (int q = 0; q < 2; q++) {
      if (r_on[q]) {
        const double zt = reg_dot<RW>(q ? vr1 : vr0, ro[q], s_xt);
        const double zr = alpha * zt + (1.0 - alpha) * r_z[q];
        double zn = zr + r_rinv[q] * r_y[q];
        zn = fmin(fmax(zn, r_ls[q]), r_us[q]);
        const double dy = r_rho[q] * (zr - zn);
        r_y[q] += dy; r_z[q] = zn;
        s_tv[tid + q * RT] = r_w[q] * (r_rho[q] * zn - r_y[q]);
        if (chk) { sdy[tid + q * RT] = dy; swy[tid + q * RT] = r_w[q] * r_y[q]; }
      }
    }
    if (colon) {
      const double xn = alpha * s_xt[tid] + (1.0 - alpha) * xj;
      if (chk) { sdx[tid] = xn - xj; sx[tid] = xn; }
      xj = xn;
    }
    if (chk && tid == 0) { sx[n] = 0.0; sdx[n] = 0.0; swy[m] = 0.0; }   // zero targets of padded slots
    __syncthreads();
    if (!chk) continue;

    // ---- termination test (formulas of admm_check in sco_qp.hip) -------------------------
    const bool adapt_pt = a.adaptive && iter % a.ad_interval == 0 && iter < a.max_iter;
    double vs[7] = {0, 0, 0, 0, 0, 0, 0};         // adaptive rho: the same norms of the SCALED iterates
    for (int approximate = 0; approximate < 2 && !status; approximate++) {
      if (approximate && iter < a.max_iter) break;
      const double *Ps = a.Ps + (size_t)b * a.nnzP;
      const double *Dg = a.D + (size_t)b * n, *Eg = a.E + (size_t)b * m;
      const double cinv = 1.0 / cscale;
      double ea = a.eps_abs, er = a.eps_rel, epi = a.eps_prim_inf, edi = a.eps_dual_inf;
      if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
      double v[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < 2; q++)
        if (r_on[q]) {
          const double ax = reg_dot<RW>(q ? vr1 : vr0, ro[q], sx);
          const double ei = 1.0 / Eg[tid + q * RT];
          v[0] = fmax(v[0], fabs(ei * (ax - r_z[q])));
          v[1] = fmax(v[1], fabs(ei * r_z[q]));
          v[2] = fmax(v[2], fabs(ei * ax));
          if (adapt_pt) { vs[0] = fmax(vs[0], fabs(ax - r_z[q])); vs[1] = fmax(vs[1], fabs(r_z[q])); vs[2] = fmax(vs[2], fabs(ax)); }
        }
      if (colon) {
        double px = 0.0;
        for (int t = a.Fp[tid]; t < a.Fp[tid + 1]; t++) px += Ps[a.Fpos[t]] * sx[a.Fi[t]];
        const double aty = reg_dot<CW>(vc, co, swy);
        const double dj = 1.0 / Dg[tid];
        v[3] = fabs(dj * (qj + px + aty)); v[4] = fabs(dj * qj); v[5] = fabs(dj * aty); v[6] = fabs(dj * px);
        if (adapt_pt) { vs[3] = fabs(qj + px + aty); vs[4] = fabs(qj); vs[5] = fabs(aty); vs[6] = fabs(px); }
      }
      rblock_reduce<7, true>(v, s_red);
      pri = v[0]; dua = cinv * v[3];
      if (!(pri <= SCO_INFTY) || !(dua <= SCO_INFTY)) { status = SCO_QP_NON_CVX; break; }
      const double eps_p = ea + er * fmax(v[1], v[2]);
      const double eps_d = ea + er * cinv * fmax(v[4], fmax(v[5], v[6]));
      const bool prim_ok = (m == 0) || (pri < eps_p), dual_ok = dua < eps_d;
      if (prim_ok && dual_ok) { status = approximate ? SCO_QP_SOLVED_INACCURATE : SCO_QP_SOLVED; break; }
      if (!prim_ok) {
        double r1[1] = {0.0};
#pragma unroll
        for (int q = 0; q < 2; q++)
          if (r_on[q]) {
            double dy = sdy[tid + q * RT];
            if (r_us[q] > SCO_INFTY * SCO_MIN_SCALING) {
              if (r_ls[q] < -SCO_INFTY * SCO_MIN_SCALING) dy = 0.0; else dy = fmin(dy, 0.0);
            } else if (r_ls[q] < -SCO_INFTY * SCO_MIN_SCALING) dy = fmax(dy, 0.0);
            sdy[tid + q * RT] = dy;
            r1[0] = fmax(r1[0], fabs(Eg[tid + q * RT] * dy));
          }
        rblock_reduce<1, true>(r1, s_red);
        const double ndy = r1[0];
        if (ndy > epi) {
          double lhs[1] = {0.0};
#pragma unroll
          for (int q = 0; q < 2; q++)
            if (r_on[q]) { const double dy = sdy[tid + q * RT]; lhs[0] += r_w[q] * (r_us[q] * fmax(dy, 0.0) + r_ls[q] * fmin(dy, 0.0)); }
          rblock_reduce<1, false>(lhs, s_red);
          if (lhs[0] < -epi * ndy) {
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; q++) if (r_on[q]) swy[tid + q * RT] = r_w[q] * sdy[tid + q * RT];
            __syncthreads();
            double nat[1] = {0.0};
            if (colon) nat[0] = fabs(reg_dot<CW>(vc, co, swy) / Dg[tid]);
            rblock_reduce<1, true>(nat, s_red);
#pragma unroll
            for (int q = 0; q < 2; q++) if (r_on[q]) swy[tid + q * RT] = r_w[q] * r_y[q];
            __syncthreads();
            if (nat[0] < epi * ndy) { status = approximate ? SCO_QP_PRIMAL_INFEASIBLE_INACCURATE : SCO_QP_PRIMAL_INFEASIBLE; break; }
          }
        }
      }
      if (!dual_ok) {
        double r1[1] = {0.0};
        if (colon) r1[0] = fabs(Dg[tid] * sdx[tid]);
        rblock_reduce<1, true>(r1, s_red);
        const double ndx = r1[0];
        if (ndx > edi) {
          double qdx[1] = {0.0};
          if (colon) qdx[0] = qj * sdx[tid];
          rblock_reduce<1, false>(qdx, s_red);
          if (qdx[0] < -cscale * edi * ndx) {
            double npx[1] = {0.0};
            if (colon) {
              double px = 0.0;
              for (int t = a.Fp[tid]; t < a.Fp[tid + 1]; t++) px += Ps[a.Fpos[t]] * sdx[a.Fi[t]];
              npx[0] = fabs(px / Dg[tid]);
            }
            rblock_reduce<1, true>(npx, s_red);
            if (npx[0] < cscale * edi * ndx) {
              double bad[1] = {0.0};
#pragma unroll
              for (int q = 0; q < 2; q++)
                if (r_on[q]) {
                  const double adx = reg_dot<RW>(q ? vr1 : vr0, ro[q], sdx) / Eg[tid + q * RT];
                  if ((r_us[q] < SCO_INFTY * SCO_MIN_SCALING && adx > edi * ndx) ||
                      (r_ls[q] > -SCO_INFTY * SCO_MIN_SCALING && adx < -edi * ndx)) bad[0] = 1.0;
                }
              rblock_reduce<1, true>(bad, s_red);
              if (bad[0] == 0.0) { status = approximate ? SCO_QP_DUAL_INFEASIBLE_INACCURATE : SCO_QP_DUAL_INFEASIBLE; break; }
            }
          }
        }
      }
    }
    __syncthreads();     // scratch slab is reused by the next iteration's partial sums
    if (status) break;
    double rho_new = 0.0;
    if (adapt_pt) {
      // OSQP's rho estimate (same rule as admm_rho_estimate in sco_qp.hip)
      rblock_reduce<7, true>(vs, s_red);
      const double rho = a.rho_b[b];
      const double pn = vs[0] / (fmax(vs[1], vs[2]) + 1e-10);
      const double dn = vs[3] / (fmax(vs[4], fmax(vs[5], vs[6])) + 1e-10);
      const double est = fmin(fmax(rho * sqrt(pn / (dn + 1e-10)), SCO_RHO_MIN), 1e6);
      if (est > rho * a.ad_tol || est < rho / a.ad_tol) rho_new = est;
    }
    if (iter < a.max_iter && (rho_new > 0.0 || (a.slice > 0 && iter == it0 + a.slice))) {
      // rho must change or the slice is used up: park the solve
      if (colon) a.sx[(size_t)b * n + tid] = xj;
#pragma unroll
      for (int q = 0; q < 2; q++)
        if (r_on[q]) { a.sz[(size_t)b * m + tid + q * RT] = r_z[q]; a.sy[(size_t)b * m + tid + q * RT] = r_y[q]; }
      if (tid == 0) {
        a.prog[b] = iter; a.status[b] = 0; a.iters[b] = iter;
        if (rho_new > 0.0) { a.rho_b[b] = rho_new; a.rflag[b] = 1; a.smask[b] = 1; a.nupd[b] += 1; }
      }
      return;
    }
  }
  if (a.slice > 0 && tid == 0) a.prog[b] = 0;
  if (!status) status = SCO_QP_MAX_ITER_REACHED;
  if (iter > a.max_iter) iter = a.max_iter;
  {
    const double *Dg = a.D + (size_t)b * n, *Eg = a.E + (size_t)b * m;
    const double cinv = 1.0 / cscale;
    if (colon) a.x[(size_t)b * n + tid] = Dg[tid] * xj;
#pragma unroll
    for (int q = 0; q < 2; q++)
      if (r_on[q]) a.y[(size_t)b * m + tid + q * RT] = cinv * Eg[tid + q * RT] * r_y[q] * r_w[q];
    if (tid == 0) {
      a.status[b] = status; a.iters[b] = iter;
      a.resid[2 * (size_t)b] = pri; a.resid[2 * (size_t)b + 1] = dua;
    }
  }
}

// --------------------------------------------------------------------------
// host glue
// --------------------------------------------------------------------------
template <typename T>
static int upv(std::vector<void *> &allocs, const std::vector<T> &v, const T **out) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  SCO_HIP(hipMalloc(&p, bytes));
  allocs.push_back(p);
  if (!v.empty()) SCO_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)p;
  return SCO_OK;
}

int reg_upload(const RegHost &rh, std::vector<void *> &allocs, RegDev &rd) {
  int rc;
  if ((rc = upv(allocs, rh.off, &rd.off))) return rc;
  if ((rc = upv(allocs, rh.role, &rd.role))) return rc;
  if ((rc = upv(allocs, rh.Ac.src, &rd.srcAc))) return rc;
  if ((rc = upv(allocs, rh.Ar.src, &rd.srcAr))) return rc;
  if ((rc = upv(allocs, rh.Ca.src, &rd.srcCa))) return rc;
  if ((rc = upv(allocs, rh.Ce.src, &rd.srcCe))) return rc;
  return SCO_OK;
}

template <int TR, int TC, int CW, int RW, int PX>
static int reg_launch_one(const RegArgs &ra, int batch, size_t lds, hipStream_t st) {
  // hipFuncSetAttribute applies to the current device only
  static bool attr_done[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  dev_ &= 63;
  if (!attr_done[dev_]) {
    SCO_HIP(hipFuncSetAttribute((const void *)qp_admm_reg_kernel<TR, TC, CW, RW, PX>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 44032));
    attr_done[dev_] = true;
  }
  hipLaunchKernelGGL((qp_admm_reg_kernel<TR, TC, CW, RW, PX>), dim3(batch), dim3(RT), lds, st, ra);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}

// cap sets instantiated: A = (12, 8, 10) [7-DOF x 20 penalty QP], B = (12, 8, 12) generic small
int reg_caps_for(const QpPlan &pl, int *CW, int *RW, int *PX) {
  int cw = 0, rw = 0, px = 0;
  for (int j = 0; j < pl.n; j++) cw = std::max(cw, pl.Ap[j + 1] - pl.Ap[j]);
  for (int i = 0; i < pl.m; i++) rw = std::max(rw, pl.Rp[i + 1] - pl.Rp[i]);
  for (int c = 0; c < pl.n_c; c++) px = std::max(px, pl.a_ptr[c + 1] - pl.a_ptr[c]);
  for (int e = 0; e < pl.n_e; e++) px = std::max(px, pl.e_ptr[e + 1] - pl.e_ptr[e]);
  if (cw > 12 || rw > 8 || px > 12) return 0;
  *CW = 12; *RW = 8; *PX = px <= 10 ? 10 : 12;
  return 1;
}

int reg_launch(const AdmmArgs &a, const RegHost &rh, const RegDev &rd, hipStream_t st) {
  const QpDev &d = a.d;
  RegArgs ra;
  ra.n = d.n; ra.m = d.m; ra.n_e = d.n_e; ra.n_c = d.n_c; ra.nnzA = d.nnzA; ra.ncpl = d.ncpl; ra.nnzP = d.nnzP;
  ra.max_iter = a.max_iter; ra.check = a.check;
  ra.sigma = a.sigma; ra.alpha = a.alpha; ra.eps_abs = a.eps_abs; ra.eps_rel = a.eps_rel;
  ra.eps_prim_inf = a.eps_prim_inf; ra.eps_dual_inf = a.eps_dual_inf;
  ra.off = rd.off; ra.role = rd.role;
  ra.srcAc = rd.srcAc; ra.srcAr = rd.srcAr; ra.srcCa = rd.srcCa; ra.srcCe = rd.srcCe;
  ra.totAc = rh.Ac.total; ra.totAr = rh.Ar.total; ra.totCa = rh.Ca.total; ra.totCe = rh.Ce.total;
  ra.As = d.As; ra.cpl = d.cpl; ra.W = d.W; ra.qs = d.qs; ra.kee_inv = d.kee_inv; ra.ls = d.ls; ra.us = d.us;
  ra.rho = d.rho; ra.cscale = d.cscale; ra.Ps = d.Ps; ra.D = d.D; ra.E = d.E;
  ra.w = d.w; ra.core_of = d.core_of; ra.Fp = d.Fp; ra.Fi = d.Fi; ra.Fpos = d.Fpos; ra.active = d.active;
  ra.x = d.x; ra.y = d.y; ra.resid = d.resid; ra.status = d.status; ra.iters = d.iters;
  ra.slice = a.slice; ra.adaptive = a.adaptive; ra.ad_interval = a.ad_interval; ra.ad_tol = a.ad_tol;
  ra.prog = d.prog; ra.rflag = d.rflag; ra.smask = d.smask; ra.nupd = d.nupd;
  ra.sx = d.sx; ra.sz = d.sz; ra.sy = d.sy; ra.rho_b = d.rho_b;
  const int key = rh.TR * 10000 + rh.TC * 100 + rh.PX;
  switch (key) {
    case 10210: return reg_launch_one<1, 2, 12, 8, 10>(ra, d.batch, rh.lds_bytes, st);
    case 10212: return reg_launch_one<1, 2, 12, 8, 12>(ra, d.batch, rh.lds_bytes, st);
    case 20410: return reg_launch_one<2, 4, 12, 8, 10>(ra, d.batch, rh.lds_bytes, st);
    case 20412: return reg_launch_one<2, 4, 12, 8, 12>(ra, d.batch, rh.lds_bytes, st);
    case 30610: return reg_launch_one<3, 6, 12, 8, 10>(ra, d.batch, rh.lds_bytes, st);
    case 30612: return reg_launch_one<3, 6, 12, 8, 12>(ra, d.batch, rh.lds_bytes, st);
    case 40810: return reg_launch_one<4, 8, 12, 8, 10>(ra, d.batch, rh.lds_bytes, st);
    case 40812: return reg_launch_one<4, 8, 12, 8, 12>(ra, d.batch, rh.lds_bytes, st);
    case 50910: return reg_launch_one<5, 9, 12, 8, 10>(ra, d.batch, rh.lds_bytes, st);
    case 50912: return reg_launch_one<5, 9, 12, 8, 12>(ra, d.batch, rh.lds_bytes, st);
    case 51010: return reg_launch_one<5, 10, 12, 8, 10>(ra, d.batch, rh.lds_bytes, st);
    case 51012: return reg_launch_one<5, 10, 12, 8, 12>(ra, d.batch, rh.lds_bytes, st);
  }
  sco_set_error("reg_launch: unsupported tile");
  return SCO_ERR_CAPACITY;
}
