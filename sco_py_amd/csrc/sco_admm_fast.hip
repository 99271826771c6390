// sco_admm_fast.hip -- the ADMM loop of libsco_hip tuned for CDNA4 (gfx950).
//
// Same mathematics and the same iterate sequence as qp_admm_kernel in sco_qp.hip
// (OSQP's ADMM, third-party call behind
// /root/reference/sco_py/sco_osqp/osqp_utils.py:216); different data movement:
//
//   * one workgroup of 512 threads (8 wavefronts, 2 per SIMD) per problem, one
//     problem per CU, the whole solve inside one launch;
//   * the dense core inverse W (n_c x n_c) lives in REGISTERS: thread (gi, gj) of a
//     32 x 16 thread grid owns the TR x TC tile W[gi*TR.., gj*TC..]; a mat-vec is
//     TR*TC FMAs per thread + one LDS hop for the 16 partial sums of every row;
//   * the four sparse operators of an iteration (A by column, A by row, the
//     coupling block K_CE by core variable and by eliminated variable) are kept in
//     LDS in sliced-ELL form: slices of 64 items = one wavefront, entry k of item
//     (slice s, lane l) at base[s] + 64 k + l.  Consecutive lanes read consecutive
//     addresses (no bank conflicts), every lane of a wavefront runs the same trip
//     count (no divergence), indices are 16-bit;
//   * per-row state (l, u, rho, z, y, w) and per-column state (q, x, 1/K_ee) are
//     private to the owning thread and stay in registers for the whole solve;
//     only the vectors that cross threads (x~, t, the eliminated right-hand side,
//     the core right-hand side and solution) go through LDS;
//   * HBM is read once (problem + W) and written once (answer).
#include "sco_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

#define FT 512            // threads per workgroup
#define FW (FT / 64)      // wavefronts per workgroup
#define GJ 16             // column groups of the W tiling
#define GI (FT / GJ)      // row groups (32)

// --------------------------------------------------------------------------
// host: sliced-ELL construction
// --------------------------------------------------------------------------
void build_sell(int nitems, const std::vector<int> &ptr, const std::vector<int> &idx,
                       const std::vector<int> &src, SellHost &out) {
  out.nitems = nitems;
  const int ns = (nitems + 63) / 64;
  out.base.assign(ns + 1, 0); out.width.assign(std::max(ns, 1), 0);
  for (int s = 0; s < ns; s++) {
    int w = 0;
    for (int it = s * 64; it < std::min(nitems, s * 64 + 64); it++) w = std::max(w, ptr[it + 1] - ptr[it]);
    out.width[s] = w; out.base[s + 1] = out.base[s] + 64 * w;
  }
  out.total = out.base[ns];
  out.idx.assign(std::max(out.total, 1), 0); out.src.assign(std::max(out.total, 1), -1);
  for (int it = 0; it < nitems; it++) {
    const int s = it / 64, l = it % 64;
    for (int k = 0; k < ptr[it + 1] - ptr[it]; k++) {
      const int p = out.base[s] + 64 * k + l;
      out.idx[p] = (unsigned short)idx[ptr[it] + k];
      out.src[p] = src[ptr[it] + k];
    }
  }
}

bool fast_plan_build(const QpPlan &pl, FastHost &fh) {
  if (pl.n > FT || pl.m > 3 * FT || pl.nnzA >= 65536 || pl.n_c > 32 * 5) return false;
  std::vector<int> ident(std::max(pl.nnzA, pl.ncpl) + 1);
  for (size_t i = 0; i < ident.size(); i++) ident[i] = (int)i;
  build_sell(pl.n, pl.Ap, pl.Ai, ident, fh.Ac);                       // A by column: idx = row, src = CSC position
  build_sell(pl.m, pl.Rp, pl.Rj, pl.Rpos, fh.Ar);                      // A by row:    idx = column
  {
    std::vector<int> eidx(pl.ncpl), srck(pl.ncpl);
    for (int t = 0; t < pl.ncpl; t++) { eidx[t] = pl.pair_elim[pl.a_pair[t]]; srck[t] = pl.a_pair[t]; }
    build_sell(pl.n_c, pl.a_ptr, eidx, srck, fh.Ca);                   // K_CE by core: idx = eliminated index
  }
  build_sell(pl.n_e, pl.e_ptr, pl.pair_core, ident, fh.Ce);            // K_CE by eliminated: idx = core index
  fh.TR = std::max(1, (pl.n_c + GI - 1) / GI);
  fh.TC = 2 * fh.TR;
  if (fh.TR == 5 && pl.n_c <= GJ * 9) fh.TC = 9;        // 140-order core: 5 x 9 tile
  auto maxw = [](const SellHost &h) { int w = 0; for (int x : h.width) w = std::max(w, x); return w; };
  fh.capped = maxw(fh.Ac) <= 12 && maxw(fh.Ar) <= 8 && maxw(fh.Ca) <= 12 && maxw(fh.Ce) <= 8;
  const int ps = GI * fh.TR + 1;
  const size_t scratch = std::max<size_t>((size_t)GJ * ps, 2 * (size_t)pl.n + 2 * (size_t)pl.m);
  fh.lds_doubles = (size_t)fh.Ac.total + fh.Ar.total + fh.Ca.total + fh.Ce.total +   // values
                   pl.n + pl.m + pl.n_e + (size_t)GJ * fh.TC + (size_t)GI * fh.TR +  // xt, t, ge, r, xc
                   scratch + FW * 8;
  fh.lds_bytes = fh.lds_doubles * 8 + 2 * ((size_t)fh.Ac.total + fh.Ar.total + fh.Ca.total + fh.Ce.total + 8 + 64 * 12);
  return fh.lds_bytes <= 160 * 1024;
}

// --------------------------------------------------------------------------
// device
// --------------------------------------------------------------------------
__device__ __forceinline__ double fwmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double fwsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// maximum over the wavefront, valid in lane 63, by DPP row shifts / row broadcasts (see lwmax63 in sco_admm_rl.hip)
__device__ __forceinline__ double fwmax63(double v) {
  int lo, hi, lo2, hi2;
#define FAST_DPP_MAX(ctrl, rmask)                                                            \
  lo = __double2loint(v); hi = __double2hiint(v);                                            \
  lo2 = __builtin_amdgcn_update_dpp(lo, lo, ctrl, rmask, 0xf, false);                        \
  hi2 = __builtin_amdgcn_update_dpp(hi, hi, ctrl, rmask, 0xf, false);                        \
  v = fmax(v, __hiloint2double(hi2, lo2));
  FAST_DPP_MAX(0x111, 0xf) FAST_DPP_MAX(0x112, 0xf) FAST_DPP_MAX(0x114, 0xf) FAST_DPP_MAX(0x118, 0xf)
  FAST_DPP_MAX(0x142, 0xa) FAST_DPP_MAX(0x143, 0xc)
#undef FAST_DPP_MAX
  return v;
}
template <int NR, bool IS_MAX>
__device__ __forceinline__ void fblock_reduce(double (&v)[NR], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = IS_MAX ? fwmax63(v[k]) : fwsum(v[k]);
  __syncthreads();
  if (lane == (IS_MAX ? 63 : 0)) {
#pragma unroll
    for (int k = 0; k < NR; k++) red[wv * NR + k] = v[k];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < NR; k++) {
    double r = red[k];
#pragma unroll
    for (int w = 1; w < FW; w++) r = IS_MAX ? fmax(r, red[w * NR + k]) : r + red[w * NR + k];
    v[k] = r;
  }
}

struct SellLds { const double *V; const unsigned short *I; int off, width; };

// sum_k V[k] * vec[I[k]] over the calling thread's item (off = base[slice] + lane).
// CAP > 0: the trip count is a compile-time bound, every index and value load is
// issued before the first gather, so a dot product costs two LDS round trips
// instead of two per entry (entries past `width` are read -- they stay inside the
// LDS allocation -- and masked).  CAP == 0: plain loop for patterns wider than the
// instantiated caps.  Summation order is k = 0, 1, ... in both forms.
template <int CAP>
__device__ __forceinline__ double sell_dot(const SellLds &s, const double *vec) {
  const double *V = s.V + s.off; const unsigned short *I = s.I + s.off;
  double acc = 0.0;
  if constexpr (CAP == 0) {
    for (int k = 0; k < s.width; k++) acc += V[64 * k] * vec[I[64 * k]];
  } else {
    unsigned int ix[CAP]; double vv[CAP];
#pragma unroll
    for (int k = 0; k < CAP; k++) {
      const unsigned int i = I[64 * k]; const double v = V[64 * k];
      const bool on = k < s.width;
      ix[k] = on ? i : 0u; vv[k] = on ? v : 0.0;
    }
    double g[CAP];
#pragma unroll
    for (int k = 0; k < CAP; k++) g[k] = vec[ix[k]];
#pragma unroll
    for (int k = 0; k < CAP; k++) acc += vv[k] * g[k];
  }
  return acc;
}

struct FastRow { double ls, us, rho, z, y, w; SellLds ar; int i; bool on; };

template <int TR, int TC, int CW, int RW, int PA, int PE, int NQ>
__global__ __launch_bounds__(FT) void qp_admm_fast_kernel(AdmmArgs a, FastDev f) {
  constexpr int PS = GI * TR + 1;            // padded row stride of the partial-sum slab
  const QpDev &d = a.d;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  if (d.active && !d.active[b]) return;
  const int n = d.n, m = d.m, n_e = d.n_e, n_c = d.n_c;
  // park / resume (time slicing, adaptive rho; see RlArgs in sco_admm_rl.hip): the loop carries x, z, y and
  // t = w (rho z - y); a resumed solve reloads the first three and rebuilds t with the rho in force now
  const int it0 = a.slice > 0 ? d.prog[b] : 0;

  extern __shared__ double lds[];
  double *VAc = lds;                    double *VAr = VAc + f.Ac.total;
  double *VCa = VAr + f.Ar.total;       double *VCe = VCa + f.Ca.total;
  double *xt = VCe + f.Ce.total;        double *tv = xt + n;
  double *ge = tv + m;                  double *rv = ge + n_e;           // rv: GJ * TC (padded core rhs)
  double *xc = rv + GJ * TC;            double *scr = xc + GI * TR;      // scr: partial sums / check scratch
  const size_t scratch = ((size_t)GJ * PS > 2 * (size_t)n + 2 * (size_t)m) ? (size_t)GJ * PS : 2 * (size_t)n + 2 * (size_t)m;
  double *red = scr + scratch;
  unsigned short *IAc = (unsigned short *)(red + FW * 8);
  unsigned short *IAr = IAc + f.Ac.total; unsigned short *ICa = IAr + f.Ar.total; unsigned short *ICe = ICa + f.Ca.total;
  // check-time aliases inside the scratch slab
  double *sx = scr, *swy = scr + n, *sdy = swy + m, *sdx = sdy + m;

  // ---- prologue: scatter values into the four sliced-ELL images ---------------
  {
    const double *gAs = d.As + (size_t)b * d.nnzA;
    const double *gcp = d.cpl + (size_t)b * d.ncpl;
    for (int p = tid; p < f.Ac.total; p += FT) { const int s = f.Ac.src[p]; VAc[p] = s >= 0 ? gAs[s] : 0.0; IAc[p] = f.Ac.idx[p]; }
    for (int p = tid; p < f.Ar.total; p += FT) { const int s = f.Ar.src[p]; VAr[p] = s >= 0 ? gAs[s] : 0.0; IAr[p] = f.Ar.idx[p]; }
    for (int p = tid; p < f.Ca.total; p += FT) { const int s = f.Ca.src[p]; VCa[p] = s >= 0 ? gcp[s] : 0.0; ICa[p] = f.Ca.idx[p]; }
    for (int p = tid; p < f.Ce.total; p += FT) { const int s = f.Ce.src[p]; VCe[p] = s >= 0 ? gcp[s] : 0.0; ICe[p] = f.Ce.idx[p]; }
    for (int i = tid; i < m; i += FT) tv[i] = 0.0;
    for (int c = tid; c < GJ * TC; c += FT) rv[c] = 0.0;
    for (int c = tid; c < GI * TR; c += FT) xc[c] = 0.0;
  }
  // ---- W tile -> registers ------------------------------------------------------
  const int gi = tid / GJ, gj = tid % GJ;
  double wreg[TR][TC];
  {
    const double *W = d.W + (size_t)b * n_c * n_c;
#pragma unroll
    for (int rr = 0; rr < TR; rr++)
#pragma unroll
      for (int cc = 0; cc < TC; cc++) {
        const int row = gi * TR + rr, col = gj * TC + cc;
        wreg[rr][cc] = (row < n_c && col < n_c) ? W[(size_t)row * n_c + col] : 0.0;
      }
  }
  // ---- private column state (column j = tid) --------------------------------------
  const bool colon = tid < n;
  const int j = colon ? tid : 0;
  double qj = 0.0, xj = 0.0, kinv = 0.0;
  int ej = -1, cj = -1;
  SellLds ac{VAc, IAc, 0, 0}, ce{VCe, ICe, 0, 0};
  if (colon) {
    qj = d.qs[(size_t)b * n + j];
    if (it0 > 0) xj = d.sx[(size_t)b * n + j];
    ej = d.elim_of[j]; cj = d.core_of[j];
    if (ej >= 0) {
      kinv = d.kee_inv[(size_t)b * n_e + ej];
      ce.off = f.Ce.base[ej >> 6] + (ej & 63); ce.width = f.Ce.width[ej >> 6];
    }
    ac.off = f.Ac.base[j >> 6] + lane; ac.width = f.Ac.width[j >> 6];
  }
  // ---- private core state (core c = tid) -------------------------------------------
  const bool coreon = tid < n_c;
  int cvar = 0;
  SellLds ca{VCa, ICa, 0, 0};
  if (coreon) { cvar = d.core_var[tid]; ca.off = f.Ca.base[tid >> 6] + lane; ca.width = f.Ca.width[tid >> 6]; }
  // ---- private row state (rows tid, tid + FT and, NQ = 3, tid + 2 FT) ----------------
  FastRow R[NQ];
#pragma unroll
  for (int q = 0; q < NQ; q++) {
    const int i = tid + q * FT;
    R[q].on = i < m; R[q].i = R[q].on ? i : 0;
    R[q].ls = R[q].us = R[q].z = R[q].y = 0.0; R[q].rho = 1.0; R[q].w = 1.0;
    R[q].ar = SellLds{VAr, IAr, 0, 0};
    if (R[q].on) {
      R[q].ls = d.ls[(size_t)b * m + i]; R[q].us = d.us[(size_t)b * m + i];
      R[q].rho = d.rho[(size_t)b * m + i]; R[q].w = (double)d.w[(size_t)b * m + i];
      R[q].ar.off = f.Ar.base[i >> 6] + lane; R[q].ar.width = f.Ar.width[i >> 6];
      if (it0 > 0) { R[q].z = d.sz[(size_t)b * m + i]; R[q].y = d.sy[(size_t)b * m + i]; }
    }
  }
  const double cscale = d.cscale[b];
  const double alpha = a.alpha, sigma = a.sigma;
  if (a.adaptive && tid == 0) { d.smask[b] = 0; d.rflag[b] = 0; }
  __syncthreads();
  if (it0 > 0) {
#pragma unroll
    for (int q = 0; q < NQ; q++)
      if (R[q].on) tv[R[q].i] = R[q].w * (R[q].rho * R[q].z - R[q].y);
    __syncthreads();
  }

  int status = 0, iter = 0;
  double pri = 0.0, dua = 0.0;
  for (iter = it0 + 1; iter <= a.max_iter; iter++) {
    const bool chk = (a.check > 0 && iter % a.check == 0) || iter == a.max_iter;
    // (1) rhs_j = sigma x_j - q_j + sum_i A_ij t_i ; eliminated part pre-scaled by 1/K_ee
    double gev = 0.0;
    if (colon) {
      double v = sell_dot<CW>(ac, tv);
      v += sigma * xj - qj;
      xt[j] = v;
      if (ej >= 0) { gev = v * kinv; ge[ej] = gev; }
    }
    __syncthreads();
    // (2) core rhs  r = rhs_C - K_CE K_EE^-1 rhs_E
    if (coreon) rv[tid] = xt[cvar] - sell_dot<PA>(ca, ge);
    __syncthreads();
    // (3a) register-tile mat-vec: TR partial sums per thread
    {
      double rr_[TC];
#pragma unroll
      for (int cc = 0; cc < TC; cc++) rr_[cc] = rv[gj * TC + cc];
#pragma unroll
      for (int rr = 0; rr < TR; rr++) {
        double acc = 0.0;
#pragma unroll
        for (int cc = 0; cc < TC; cc++) acc += wreg[rr][cc] * rr_[cc];
        scr[gj * PS + gi * TR + rr] = acc;
      }
    }
    __syncthreads();
    // (3b) x~_C = sum of the GJ partial sums (fixed order)
    if (coreon) {
      double v = 0.0;
#pragma unroll
      for (int g = 0; g < GJ; g++) v += scr[g * PS + tid];
      xc[tid] = v;
    }
    __syncthreads();
    // (4) back-substitute eliminated variables; x~ into xt
    if (colon) xt[j] = (ej >= 0) ? (gev - kinv * sell_dot<PE>(ce, xc)) : xc[cj];
    __syncthreads();
    // (5) z~ = A x~, then z / y / x updates and next iteration's t
#pragma unroll
    for (int q = 0; q < NQ; q++) {
      if (R[q].on) {
        const double zt = sell_dot<RW>(R[q].ar, xt);
        const double rho = R[q].rho, rinv = 1.0 / rho;
        const double zr = alpha * zt + (1.0 - alpha) * R[q].z;
        double zn = zr + rinv * R[q].y;
        zn = fmin(fmax(zn, R[q].ls), R[q].us);
        const double dy = rho * (zr - zn);
        R[q].y += dy; R[q].z = zn;
        tv[R[q].i] = R[q].w * (rho * zn - R[q].y);
        if (chk) { sdy[R[q].i] = dy; swy[R[q].i] = R[q].w * R[q].y; }
      }
    }
    if (colon) {
      const double xn = alpha * xt[j] + (1.0 - alpha) * xj;
      if (chk) { sdx[j] = xn - xj; sx[j] = xn; }
      xj = xn;
    }
    __syncthreads();
    if (!chk) continue;

    // ---- termination test (same formulas as admm_check in sco_qp.hip) ----------------
    const bool adapt_pt = a.adaptive && iter % a.ad_interval == 0 && iter < a.max_iter;
    double vs[7] = {0, 0, 0, 0, 0, 0, 0};         // adaptive rho: the same norms of the SCALED iterates
    for (int approximate = 0; approximate < 2 && !status; approximate++) {
      if (approximate && iter < a.max_iter) break;
      const double *Ps = d.Ps + (size_t)b * d.nnzP;
      const double *Dg = d.D + (size_t)b * n, *Eg = d.E + (size_t)b * m;
      const double cinv = 1.0 / cscale;
      double ea = a.eps_abs, er = a.eps_rel, epi = a.eps_prim_inf, edi = a.eps_dual_inf;
      if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
      double v[7] = {0, 0, 0, 0, 0, 0, 0};
#pragma unroll
      for (int q = 0; q < NQ; q++)
        if (R[q].on) {
          const double ax = sell_dot<RW>(R[q].ar, sx);
          const double ei = 1.0 / Eg[R[q].i];
          v[0] = fmax(v[0], fabs(ei * (ax - R[q].z)));
          v[1] = fmax(v[1], fabs(ei * R[q].z));
          v[2] = fmax(v[2], fabs(ei * ax));
          if (adapt_pt) { vs[0] = fmax(vs[0], fabs(ax - R[q].z)); vs[1] = fmax(vs[1], fabs(R[q].z)); vs[2] = fmax(vs[2], fabs(ax)); }
        }
      if (colon) {
        double px = 0.0;
        for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * sx[d.Fi[t]];
        const double aty = sell_dot<CW>(ac, swy);
        const double dj = 1.0 / Dg[j];
        v[3] = fabs(dj * (qj + px + aty)); v[4] = fabs(dj * qj); v[5] = fabs(dj * aty); v[6] = fabs(dj * px);
        if (adapt_pt) { vs[3] = fabs(qj + px + aty); vs[4] = fabs(qj); vs[5] = fabs(aty); vs[6] = fabs(px); }
      }
      fblock_reduce<7, true>(v, red);
      pri = v[0]; dua = cinv * v[3];
      if (!(pri <= SCO_INFTY) || !(dua <= SCO_INFTY)) { status = SCO_QP_NON_CVX; break; }
      const double eps_p = ea + er * fmax(v[1], v[2]);
      const double eps_d = ea + er * cinv * fmax(v[4], fmax(v[5], v[6]));
      const bool prim_ok = (m == 0) || (pri < eps_p), dual_ok = dua < eps_d;
      if (prim_ok && dual_ok) { status = approximate ? SCO_QP_SOLVED_INACCURATE : SCO_QP_SOLVED; break; }
      if (!prim_ok) {           // primal infeasibility certificate from delta_y
        double r1[1] = {0.0};
#pragma unroll
        for (int q = 0; q < NQ; q++)
          if (R[q].on) {
            double dy = sdy[R[q].i];
            if (R[q].us > SCO_INFTY * SCO_MIN_SCALING) {
              if (R[q].ls < -SCO_INFTY * SCO_MIN_SCALING) dy = 0.0; else dy = fmin(dy, 0.0);
            } else if (R[q].ls < -SCO_INFTY * SCO_MIN_SCALING) dy = fmax(dy, 0.0);
            sdy[R[q].i] = dy;
            r1[0] = fmax(r1[0], fabs(Eg[R[q].i] * dy));
          }
        fblock_reduce<1, true>(r1, red);
        const double ndy = r1[0];
        if (ndy > epi) {
          double lhs[1] = {0.0};
#pragma unroll
          for (int q = 0; q < NQ; q++)
            if (R[q].on) { const double dy = sdy[R[q].i]; lhs[0] += R[q].w * (R[q].us * fmax(dy, 0.0) + R[q].ls * fmin(dy, 0.0)); }
          fblock_reduce<1, false>(lhs, red);
          if (lhs[0] < -epi * ndy) {
            // A'(w dy): reuse swy as the weighted vector
            __syncthreads();
#pragma unroll
            for (int q = 0; q < NQ; q++) if (R[q].on) swy[R[q].i] = R[q].w * sdy[R[q].i];
            __syncthreads();
            double nat[1] = {0.0};
            if (colon) nat[0] = fabs(sell_dot<CW>(ac, swy) / Dg[j]);
            fblock_reduce<1, true>(nat, red);
            // restore w*y for a possible second (approximate) pass
#pragma unroll
            for (int q = 0; q < NQ; q++) if (R[q].on) swy[R[q].i] = R[q].w * R[q].y;
            __syncthreads();
            if (nat[0] < epi * ndy) { status = approximate ? SCO_QP_PRIMAL_INFEASIBLE_INACCURATE : SCO_QP_PRIMAL_INFEASIBLE; break; }
          }
        }
      }
      if (!dual_ok) {           // dual infeasibility certificate from delta_x
        double r1[1] = {0.0};
        if (colon) r1[0] = fabs(Dg[j] * sdx[j]);
        fblock_reduce<1, true>(r1, red);
        const double ndx = r1[0];
        if (ndx > edi) {
          double qdx[1] = {0.0};
          if (colon) qdx[0] = qj * sdx[j];
          fblock_reduce<1, false>(qdx, red);
          if (qdx[0] < -cscale * edi * ndx) {
            double npx[1] = {0.0};
            if (colon) {
              double px = 0.0;
              for (int t = d.Fp[j]; t < d.Fp[j + 1]; t++) px += Ps[d.Fpos[t]] * sdx[d.Fi[t]];
              npx[0] = fabs(px / Dg[j]);
            }
            fblock_reduce<1, true>(npx, red);
            if (npx[0] < cscale * edi * ndx) {
              double bad[1] = {0.0};
#pragma unroll
              for (int q = 0; q < NQ; q++)
                if (R[q].on) {
                  const double adx = sell_dot<RW>(R[q].ar, sdx) / Eg[R[q].i];
                  if ((R[q].us < SCO_INFTY * SCO_MIN_SCALING && adx > edi * ndx) ||
                      (R[q].ls > -SCO_INFTY * SCO_MIN_SCALING && adx < -edi * ndx)) bad[0] = 1.0;
                }
              fblock_reduce<1, true>(bad, red);
              if (bad[0] == 0.0) { status = approximate ? SCO_QP_DUAL_INFEASIBLE_INACCURATE : SCO_QP_DUAL_INFEASIBLE; break; }
            }
          }
        }
      }
    }
    __syncthreads();     // the scratch slab is reused by the next iteration's partial sums
    if (status) break;
    double rho_new = 0.0;
    if (adapt_pt) {
      // OSQP's rho estimate (same rule as admm_rho_estimate in sco_qp.hip)
      fblock_reduce<7, true>(vs, red);
      const double rho = d.rho_b[b];
      const double pn = vs[0] / (fmax(vs[1], vs[2]) + 1e-10);
      const double dn = vs[3] / (fmax(vs[4], fmax(vs[5], vs[6])) + 1e-10);
      const double est = fmin(fmax(rho * sqrt(pn / (dn + 1e-10)), SCO_RHO_MIN), 1e6);
      if (est > rho * a.ad_tol || est < rho / a.ad_tol) rho_new = est;
    }
    if (iter < a.max_iter && (rho_new > 0.0 || (a.slice > 0 && iter == it0 + a.slice))) {
      // rho must change or the slice is used up: park the solve
      if (colon) d.sx[(size_t)b * n + j] = xj;
#pragma unroll
      for (int q = 0; q < NQ; q++)
        if (R[q].on) { d.sz[(size_t)b * m + R[q].i] = R[q].z; d.sy[(size_t)b * m + R[q].i] = R[q].y; }
      if (tid == 0) {
        d.prog[b] = iter; d.status[b] = 0; d.iters[b] = iter;
        if (rho_new > 0.0) { d.rho_b[b] = rho_new; d.rflag[b] = 1; d.smask[b] = 1; d.nupd[b] += 1; }
      }
      return;
    }
  }
  if (a.slice > 0 && tid == 0) d.prog[b] = 0;
  if (!status) { status = SCO_QP_MAX_ITER_REACHED; iter = a.max_iter; }
  if (iter > a.max_iter) iter = a.max_iter;
  // ---- unscale and store ------------------------------------------------------------
  {
    const double *Dg = d.D + (size_t)b * n, *Eg = d.E + (size_t)b * m;
    const double cinv = 1.0 / cscale;
    if (colon) d.x[(size_t)b * n + j] = Dg[j] * xj;
#pragma unroll
    for (int q = 0; q < NQ; q++)
      if (R[q].on) d.y[(size_t)b * m + R[q].i] = cinv * Eg[R[q].i] * R[q].y * R[q].w;
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
static int up(std::vector<void *> &allocs, const std::vector<T> &v, const T **out) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  SCO_HIP(hipMalloc(&p, bytes));
  allocs.push_back(p);
  if (!v.empty()) SCO_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)p;
  return SCO_OK;
}

static int up_sell(std::vector<void *> &allocs, const SellHost &h, SellDev &dv) {
  dv.total = h.total;
  int rc;
  if ((rc = up(allocs, h.base, &dv.base))) return rc;
  if ((rc = up(allocs, h.width, &dv.width))) return rc;
  if ((rc = up(allocs, h.idx, &dv.idx))) return rc;
  if ((rc = up(allocs, h.src, &dv.src))) return rc;
  return SCO_OK;
}

int fast_upload(const FastHost &fh, std::vector<void *> &allocs, FastDev &fd) {
  int rc;
  if ((rc = up_sell(allocs, fh.Ac, fd.Ac))) return rc;
  if ((rc = up_sell(allocs, fh.Ar, fd.Ar))) return rc;
  if ((rc = up_sell(allocs, fh.Ca, fd.Ca))) return rc;
  if ((rc = up_sell(allocs, fh.Ce, fd.Ce))) return rc;
  return SCO_OK;
}

template <int TR, int TC, int CW, int RW, int PA, int PE, int NQ>
static int launch_one(const AdmmArgs &a, const FastDev &fd, size_t lds, hipStream_t st) {
  // hipFuncSetAttribute applies to the current device only
  static bool attr_done[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  dev_ &= 63;
  if (!attr_done[dev_]) {
    SCO_HIP(hipFuncSetAttribute((const void *)qp_admm_fast_kernel<TR, TC, CW, RW, PA, PE, NQ>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_done[dev_] = true;
  }
  hipLaunchKernelGGL((qp_admm_fast_kernel<TR, TC, CW, RW, PA, PE, NQ>), dim3(a.d.batch), dim3(FT), lds, st, a, fd);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}

template <int TR, int TC>
static int launch_tile(const AdmmArgs &a, const FastHost &fh, const FastDev &fd, hipStream_t st) {
  // more than 2 x 512 rows: three rows per thread (looping dots only: patterns this large have wide columns anyway)
  if (a.d.m > 2 * FT) return launch_one<TR, TC, 0, 0, 0, 0, 3>(a, fd, fh.lds_bytes, st);
  if (fh.capped) return launch_one<TR, TC, 12, 8, 12, 8, 2>(a, fd, fh.lds_bytes, st);
  return launch_one<TR, TC, 0, 0, 0, 0, 2>(a, fd, fh.lds_bytes, st);
}

int fast_launch(const AdmmArgs &a, const FastHost &fh, const FastDev &fd, hipStream_t st) {
  switch (fh.TR * 100 + fh.TC) {
    case 102: return launch_tile<1, 2>(a, fh, fd, st);
    case 204: return launch_tile<2, 4>(a, fh, fd, st);
    case 306: return launch_tile<3, 6>(a, fh, fd, st);
    case 408: return launch_tile<4, 8>(a, fh, fd, st);
    case 509: return launch_tile<5, 9>(a, fh, fd, st);
    case 510: return launch_tile<5, 10>(a, fh, fd, st);
  }
  sco_set_error("fast_launch: unsupported tile");
  return SCO_ERR_CAPACITY;
}
