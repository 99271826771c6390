// sco_admm_wv.hip -- "wavefront" tier of the ADMM solve: ONE wavefront per problem, several problems per CU.
//
// Same algorithm as every other tier (OSQP's ADMM behind /root/reference/sco_py/sco_osqp/osqp_utils.py:195-216, reduced
// system K x~ = sigma x - q + A'(R z - y) with the hinge slacks eliminated in closed form, qp_plan.h).  What changes is
// the algebra of the core solve and, with it, how much of the chip one problem needs:
//
//   * the reduced matrix S of a trajectory penalty QP is BLOCK TRIDIAGONAL (one block per timestep, order bs = DOF;
//     the off-diagonal blocks come from the smoothness objective and are diagonal matrices).  The row-local tier
//     (sco_admm_rl.hip) applies the dense inverse W = S^-1 -- 157 KB of registers at 7 x 20, 71 % of its multiply-adds,
//     and the reason one problem fills a CU.  Here S is factored once per QP as a TWISTED block LDL' (blocks
//     eliminated from both ends towards the middle, explicit pivot inverses G_t = Sigma_t^-1: nb x bs x bs numbers,
//     9 KB) and every iteration runs two forward and two backward block sweeps, the two chains side by side in two
//     16-lane DPP rows, a bs x bs mat-vec = bs `v_fmac_f64_dpp row_newbcast` instructions;
//   * all row / variable state of a problem lives in the registers of ONE wavefront (lane = (timestep, part): NS hinge
//     row slots with their slack variable and its bound row, NV core-variable slots with their box row), the Jacobian
//     rows in lane-private LDS; partial column sums and the sweep's vectors go through 10 KB of LDS; no barrier ever
//     waits for another wavefront.  A problem needs <= 40 KB of LDS and one wavefront: 4 problems per CU.
//
// The tier takes patterns with this structure only (wv_plan_build), and per problem only VALUES with the structure of a
// penalty QP (hinge rows l = -inf with one common weight, slack rows [0, inf), everything on the base rho: checked by
// qp_wv_factor_kernel).  A problem that fails the value test is left to the row-local kernel, which the launcher runs
// right behind this one for exactly those problems -- so the tier never changes what is solved, only where.
//
// Block positions.  The sweeps are unrolled NSTEP times (a template argument); both chains are padded AT THEIR START
// with all-zero dummy blocks to exactly NSTEP blocks, so one instantiation serves every horizon up to 2 NSTEP + 1:
//   positions 0 .. NSTEP-1      chain A = blocks 0 .. mid-1 (the real ones last),
//   position  NSTEP             the middle block mid = nb / 2,
//   positions NSTEP+1 .. 2NSTEP chain B = blocks nb-1 .. mid+1 (the real ones last),
//   position  2NSTEP+1          dummy block of idle lanes.
// A dummy block has G = 0 and zero couplings: it maps zeros to zeros and the stores into it write zeros.
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "sco_internal.h"

#define WV_T 64
#ifndef WV_PD
#define WV_PD 3       // block steps between the LDS reads of a sweep step and its arithmetic
#endif
#define WV_MAXNS 4
#define WV_MAXNV 4
typedef double d2 __attribute__((ext_vector_type(2)));

// (a VALU write of the broadcast operand needs two wait states before a DPP instruction reads it: the first of a chain
// carries them itself, inline assembly is opaque to the compiler's hazard recogniser.  Plain `asm`, not `asm volatile`:
// a volatile statement is a scheduling barrier, and the loads of the next block step have to move above this one's chain)
#ifndef WV_KEEP
#define WV_KEEP 8        // blocks of G per chain the forward sweep keeps in registers for the backward sweep
#endif
#ifndef WV_JREG
#define WV_JREG 1
#endif
#ifndef WV_VARIANT
#define WV_VARIANT 0          // timing variants (scripts/build_ablate.py wv:WV_VARIANT=1): 1 = plain v_fmac_f64 instead of the DPP broadcasts (results wrong)
#endif
#if WV_VARIANT == 1
#define WV_FMAC_DPP(acc, w, g, k) asm("v_fmac_f64 %0, %1, %2" : "+v"(acc) : "v"(w), "v"(g))
#else
#define WV_FMAC_DPP(acc, w, g, k) asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #k " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(w), "v"(g))
#endif

__host__ __device__ __forceinline__ size_t wv_tri(int i, int j) { return (size_t)i * (i + 1) / 2 + j; }   // packed lower, j <= i

// LDS layouts chosen against bank conflicts (r04 PMC: 42 % of the LDS cycles of the first version were conflicts, the LDS
// pipe of a CU with four of these wavefronts was busy 68 % of the time).  A 64-byte row per lane at a 64-byte lane stride
// puts lanes i and i + 4 on the same banks for every 16-byte piece; instead the four pieces of a vector are stored
// PIECE-MAJOR, 16 bytes per block (or per lane slot) inside a piece:
//   block vectors (r, x~, x, extra)  index(pos, k) = (k >> 1) * 2 NPOS + 2 pos + (k & 1)
//   partial column sums              index(j, slot) = j * 2 NSLOT + 2 slot              (slot = pos * lpb + part)
//   rows of G, block pos             index(row i, col c) = 64 pos + 16 ((c >> 1) + shift & 3) + 2 i + (c & 1), shift = 1 for
//                                    the blocks of chain B (both chains read their piece j in the same instruction)
__host__ __device__ inline int wv_vidx(int npos, int pos, int k) { return (k >> 1) * 2 * npos + 2 * pos + (k & 1); }
__host__ __device__ inline int wv_gidx(int nstep, int pos, int i, int c) { return 64 * pos + 16 * (((c >> 1) + (pos > nstep ? 1 : 0)) & 3) + 2 * i + (c & 1); }

// int tables, [..][64] lane-minor; offsets in units of 64 ints
struct WvTab {
  int pos, hrow, hbrow, hevar, heidx, hepos, hbpos, hjpos, vvar, vrow, vpos, vpk, vpkm, vpkp, vpd, vpm, vpp, xrow, xpos, xpk, total;
};
__host__ __device__ inline WvTab wv_tab_layout(int NS, int NV) {
  WvTab t; int o = 0;
  t.pos = o; o += 1;
  t.hrow = o; o += NS; t.hbrow = o; o += NS; t.hevar = o; o += NS; t.heidx = o; o += NS; t.hepos = o; o += NS; t.hbpos = o; o += NS;
  t.hjpos = o; o += NS * 8;
  t.vvar = o; o += NV; t.vrow = o; o += NV; t.vpos = o; o += NV; t.vpk = o; o += NV; t.vpkm = o; o += NV; t.vpkp = o; o += NV;
  t.vpd = o; o += NV * 8; t.vpm = o; o += NV; t.vpp = o; o += NV;
  t.xrow = o; o += 1; t.xpos = o; o += 1; t.xpk = o; o += 1;
  t.total = o;
  return t;
}
// constants of the termination test, [slot][64] per problem: hinge slot q: 1/E_h, 1/E_b, 1/D_e; variable slot v: 1/D_c,
// 1/E_0, P(c, c-), P(c, c+), P(c, c), then P(c, block)[8] (read only by problems whose P has entries off these three
// diagonals: pflag); extra row: 1/E_x.  The scalings themselves (E, D) are formed from the reciprocals when the
// certificates' norms need them: the fetch of these constants is what a checked iteration costs most
// (every wavefront of the chip asks for them at the same time).
__host__ __device__ inline int wv_cst_h(int q) { return 3 * q; }
__host__ __device__ inline int wv_cst_v(int NS, int v) { return 3 * NS + 13 * v; }
__host__ __device__ inline int wv_cst_x(int NS, int NV) { return 3 * NS + 13 * NV; }
// (v_rcp_f64, one instruction: these reciprocals only scale the norms of the infeasibility tests)
__device__ __forceinline__ double wv_recip(double x) { return x != 0.0 ? __builtin_amdgcn_rcp(x) : 0.0; }

// ---------------------------------------------------------------------------------------------------------------------
// host plan
// ---------------------------------------------------------------------------------------------------------------------
bool wv_plan_build(const QpPlan &pl, WvHost &wh) {
  wh = WvHost();
  const int n_c = pl.n_c, n_e = pl.n_e, m = pl.m;
  if (n_e <= 0 || n_c <= 0) return false;
  // ---- block structure of S: within a block, or between neighbouring blocks at equal in-block index
  int hb = 0;
  for (int id = 0; id < pl.nS; id++) hb = std::max(hb, pl.s_a[id] - pl.s_b[id]);
  auto valid = [&](int bs) {
    if (bs <= 0 || bs > 8 || n_c % bs) return false;
    for (int id = 0; id < pl.nS; id++) {
      const int a = pl.s_a[id], b = pl.s_b[id], ta = a / bs, tb = b / bs;
      if (!(ta == tb || (ta == tb + 1 && a % bs == b % bs))) return false;
    }
    return true;
  };
  int bs = 0;
  if (valid(hb)) bs = hb; else if (valid(n_c)) bs = n_c; else return false;
  const int nb = n_c / bs;
  for (int e = 0; e < n_e; e++) if (pl.Pdiag[pl.elim_var[e]] >= 0) return false;      // no P entry on an eliminated variable
  // ---- rows
  std::vector<int> h_of(n_e, -1), b_of(n_e, -1), hblk(n_e, -1);
  std::vector<std::vector<int>> single(n_c);
  for (int i = 0; i < m; i++) {
    int ne = 0, nc = 0, e = -1, blk = -1; bool one_blk = true;
    for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
      const int j = pl.Rj[s];
      if (pl.elim_of[j] >= 0) { ne++; e = pl.elim_of[j]; }
      else { const int t = pl.core_of[j] / bs; if (nc && t != blk) one_blk = false; blk = t; nc++; }
    }
    if (ne == 1 && nc >= 1 && one_blk) { if (h_of[e] >= 0) return false; h_of[e] = i; hblk[e] = blk; }
    else if (ne == 1 && nc == 0) { if (b_of[e] >= 0) return false; b_of[e] = i; }
    else if (ne == 0 && nc == 1) single[pl.core_of[pl.Rj[pl.Rp[i]]]].push_back(i);
    else return false;
  }
  for (int e = 0; e < n_e; e++)
    if (h_of[e] < 0 || b_of[e] < 0 || pl.Ap[pl.elim_var[e] + 1] - pl.Ap[pl.elim_var[e]] != 2) return false;
  int n_extra = 0;
  for (int c = 0; c < n_c; c++) { if (single[c].size() > 2) return false; if (single[c].size() == 2) n_extra++; }
  if (n_extra > WV_T) return false;
  // ---- lanes
  int lpb = 0;
  for (int L = 8; L >= 1; L--) if (nb <= 4 * (16 / L)) { lpb = L; break; }
  if (!lpb) return false;
  const int bpr = 16 / lpb;
  std::vector<std::vector<int>> hin(nb);
  for (int e = 0; e < n_e; e++) hin[hblk[e]].push_back(e);          // elimination order = row order inside a block
  int maxh = 0;
  for (auto &v : hin) maxh = std::max(maxh, (int)v.size());
  const int NS = std::max(1, (maxh + lpb - 1) / lpb), NV = (bs + lpb - 1) / lpb;
  const int mid = nb / 2;
  // instantiations <BS, NS, NV, NSTEP>, the smallest that holds the shape first (row slots and padded block steps of a larger one
  // are executed all the same): <8, 1, 1, 4>, <8, 2, 2, 8>, <8, 2, 2, 10>, and <7, 4, 3, 10> for the 7-DOF x 20 shapes.  A shape that
  // would need more (e.g. 4-DOF x 24: 3 row slots, 12 steps) stays on the row-local tier: an instantiation with 4 x 4 slots and 16
  // steps (r04, dropped) held 496 VGPRs and 42 KB of LDS -- three problems per CU at half the row-local kernel's rate
  // (profiles/r04_tier_choice.txt).
  if (NS <= 1 && NV <= 1 && mid <= 4) { wh.BS = 8; wh.NS = 1; wh.NV = 1; wh.NSTEP = 4; }
  else if (NS <= 2 && NV <= 2 && mid <= 8) { wh.BS = 8; wh.NS = 2; wh.NV = 2; wh.NSTEP = 8; }
  else if (NS <= 2 && NV <= 2 && mid <= 10) { wh.BS = 8; wh.NS = 2; wh.NV = 2; wh.NSTEP = 10; }
  else if (bs == 7 && NS <= 4 && NV <= 3 && mid <= 10) { wh.BS = 7; wh.NS = 4; wh.NV = 3; wh.NSTEP = 10; }
  else return false;
  wh.bs = bs; wh.nb = nb; wh.mid = mid; wh.lpb = lpb; wh.n_extra = n_extra;
  const int NSTEP = wh.NSTEP, lenB = nb - 1 - mid, npos = 2 * NSTEP + 2, dummy = npos - 1;
  auto pos_of = [&](int t) { return t < mid ? NSTEP - mid + t : t == mid ? NSTEP : 2 * NSTEP + 1 - lenB + (nb - 1 - t); };
  auto apos = [&](int row, int var) {            // position of A(row, var) in the CSC value array
    for (int s = pl.Rp[row]; s < pl.Rp[row + 1]; s++) if (pl.Rj[s] == var) return pl.Rpos[s];
    return -1;
  };
  auto ppos = [&](int va, int vb) {              // position of P(va, vb) in the upper-triangular CSC array
    const int r = std::min(va, vb), c = std::max(va, vb);
    for (int t = pl.Pp[c]; t < pl.Pp[c + 1]; t++) if (pl.Pi[t] == r) return t;
    return -1;
  };
  const WvTab T = wv_tab_layout(wh.NS, wh.NV);
  wh.tab.assign((size_t)T.total * 64, -1);
  auto at = [&](int off, int lane) -> int & { return wh.tab[(size_t)off * 64 + lane]; };
  for (int l = 0; l < 64; l++) {
    at(T.pos, l) = dummy;                                           // idle lanes sit on the all-zero dummy block
    for (int v = 0; v < wh.NV; v++) { at(T.vpk + v, l) = wv_vidx(npos, dummy, v & 7); at(T.vpkm + v, l) = wv_vidx(npos, dummy, 0); at(T.vpkp + v, l) = wv_vidx(npos, dummy, 0); }
    at(T.xpk, l) = wv_vidx(npos, dummy, 0);
  }
  int xl = 0;
  for (int t = 0; t < nb; t++) {
    const int base = 16 * (t / bpr) + lpb * (t % bpr), p = pos_of(t);
    for (int sub = 0; sub < lpb; sub++) at(T.pos, base + sub) = p;
    for (int r = 0; r < (int)hin[t].size(); r++) {
      const int e = hin[t][r], l = base + r % lpb, q = r / lpb, ve = pl.elim_var[e];
      at(T.hrow + q, l) = h_of[e]; at(T.hbrow + q, l) = b_of[e]; at(T.hevar + q, l) = ve; at(T.heidx + q, l) = e;
      at(T.hepos + q, l) = apos(h_of[e], ve); at(T.hbpos + q, l) = apos(b_of[e], ve);
      for (int k = 0; k < bs; k++) at(T.hjpos + q * 8 + k, l) = apos(h_of[e], pl.core_var[t * bs + k]);
    }
    for (int k = 0; k < bs; k++) {
      const int l = base + k % lpb, v = k / lpb, c = t * bs + k, var = pl.core_var[c];
      at(T.vvar + v, l) = var; at(T.vpk + v, l) = wv_vidx(npos, p, k);
      at(T.vpkm + v, l) = t > 0 ? wv_vidx(npos, pos_of(t - 1), k) : wv_vidx(npos, dummy, 0);
      at(T.vpkp + v, l) = t + 1 < nb ? wv_vidx(npos, pos_of(t + 1), k) : wv_vidx(npos, dummy, 0);
      // the variable's own slot takes its LAST single row: the reference appends the bound / trust-region rows behind all
      // constraint rows (osqp_utils.py:185-189), so that is the box row, which sits on the base rho; an earlier one (a pin:
      // equality, rho_eq) goes to the general extra-row lanes
      if (!single[c].empty()) { at(T.vrow + v, l) = single[c].back(); at(T.vpos + v, l) = apos(single[c].back(), var); }
      for (int k2 = 0; k2 < bs; k2++) at(T.vpd + v * 8 + k2, l) = ppos(var, pl.core_var[t * bs + k2]);
      at(T.vpm + v, l) = t > 0 ? ppos(var, pl.core_var[(t - 1) * bs + k]) : -1;
      at(T.vpp + v, l) = t + 1 < nb ? ppos(var, pl.core_var[(t + 1) * bs + k]) : -1;
      if (single[c].size() == 2) { at(T.xrow, xl) = single[c][0]; at(T.xpos, xl) = apos(single[c][0], var); at(T.xpk, xl) = wv_vidx(npos, p, k); xl++; }
    }
  }
  // LDS of the ADMM kernel (doubles): G, ef, en, em | r, x~, x, extra | Jacobian rows (unless they live in registers) | partials
  const bool jreg = WV_JREG && wh.NS * wh.BS <= 28;
  wh.g_doubles = (size_t)npos * 64 + 2 * (size_t)npos * 8 + 8;
  // (+ the constants of the termination test, lane-minor: 3 NS + 5 NV + 1 slots of 64 doubles)
  wh.lds_doubles = wh.g_doubles + 4 * (size_t)npos * 8 + (jreg ? 0 : (size_t)wh.NS * 512) + (size_t)npos * lpb * 8 + (size_t)(3 * wh.NS + 5 * wh.NV + 1) * 64;
  wh.lds_bytes = wh.lds_doubles * sizeof(double);
  wh.cst_slots = 3 * wh.NS + 13 * wh.NV + 1;
  // the tier pays only with FOUR problems per CU (one wavefront per SIMD): 40 KB each at most
  return wh.lds_bytes <= 40 * 1024;
}

int wv_upload(const WvHost &wh, int batch, int n, int m, std::vector<void *> &allocs, WvDev &wd) {
  void *p = nullptr;
  SCO_HIP(hipMalloc(&p, wh.tab.size() * sizeof(int))); allocs.push_back(p);
  SCO_HIP(hipMemcpy(p, wh.tab.data(), wh.tab.size() * sizeof(int), hipMemcpyHostToDevice)); wd.tab = (const int *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * wh.g_doubles * sizeof(double))); allocs.push_back(p); wd.G = (double *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * wh.cst_slots * 64 * sizeof(double))); allocs.push_back(p); wd.cst = (double *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * sizeof(double))); allocs.push_back(p); wd.wc = (double *)p;
  SCO_HIP(hipMalloc(&p, (size_t)batch * sizeof(int))); allocs.push_back(p); wd.ok = (int *)p;
  SCO_HIP(hipMemset(p, 0, (size_t)batch * sizeof(int)));
  SCO_HIP(hipMalloc(&p, (size_t)batch * sizeof(int))); allocs.push_back(p); wd.rl_need = (int *)p;
  SCO_HIP(hipMemset(p, 0, (size_t)batch * sizeof(int)));
  SCO_HIP(hipMalloc(&p, sizeof(unsigned long long))); allocs.push_back(p); wd.it_count = (unsigned long long *)p;
  SCO_HIP(hipMemset(p, 0, sizeof(unsigned long long)));
  SCO_HIP(hipMalloc(&p, (size_t)batch * sizeof(int))); allocs.push_back(p); wd.pflag = (int *)p;
  SCO_HIP(hipMemset(p, 0, (size_t)batch * sizeof(int)));
  SCO_HIP(hipMalloc(&p, (size_t)batch * sizeof(int))); allocs.push_back(p); wd.w_ready = (int *)p;
  SCO_HIP(hipMemset(p, 0, (size_t)batch * sizeof(int)));
  SCO_HIP(hipMalloc(&p, (size_t)batch * (size_t)(n + m) * sizeof(double))); allocs.push_back(p); wd.scr = (double *)p;
  return SCO_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// kernel arguments
// ---------------------------------------------------------------------------------------------------------------------
struct WvArgs {
  int n, m, n_e, n_c, nnzA, nnzP, max_iter, check, slice;
  int b0; const int *list;
  int bs, nb, mid, lpb, n_extra, NS, NV, NSTEP, cst_slots; size_t g_doubles;
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  const int *tab;
  double *G, *cst, *wc, *scr; int *ok, *rl_need, *w_ready, *pflag; unsigned long long *it_count;
  const double *As, *Ps, *qs, *ls, *us, *rhov, *kee_inv, *cscale, *D, *E, *S;
  const int *w, *active, *setup_active;
  const int *Ap, *Ai, *Rp, *Rj, *Rpos, *Fp, *Fi, *Fpos;
  double *x, *y, *resid; int *status, *iters, *prog;
  double *sx, *sz, *sy, *st, *sg;
  int ablate;        // diagnostic (SCO_WV_ABLATE): 1 = no sweeps, 2 = no row passes, 16 = no termination test behind the checked iteration, 64 = no infeasibility certificates (timing only, results wrong)
};

// Wavefront-wide max / sum in registers: two DPP quad steps, two DPP mirror steps inside a 16-lane row, then the row and
// half swaps (v_permlane16_swap / v_permlane32_swap).  (__shfl_xor goes through ds_bpermute: an LDS round trip per step,
// 48 of them per termination test.)  Every lane ends with the result.
typedef unsigned int wv_u2 __attribute__((ext_vector_type(2)));
template <int CTRL>
__device__ __forceinline__ double wv_dpp(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true), __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true));
}
template <bool IS_MAX>
__device__ __forceinline__ double wv_wred(double v) {
  auto op = [](double a, double b) { return IS_MAX ? fmax(a, b) : a + b; };
  v = op(v, wv_dpp<0xb1>(v));       // quad_perm [1, 0, 3, 2]
  v = op(v, wv_dpp<0x4e>(v));       // quad_perm [2, 3, 0, 1]
  v = op(v, wv_dpp<0x141>(v));      // row_half_mirror
  v = op(v, wv_dpp<0x140>(v));      // row_mirror
  {
    const wv_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const wv_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = op(__hiloint2double((int)hi.x, (int)lo.x), __hiloint2double((int)hi.y, (int)lo.y));
  }
  {
    const wv_u2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(v), (unsigned)__double2loint(v), false, false);
    const wv_u2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(v), (unsigned)__double2hiint(v), false, false);
    v = op(__hiloint2double((int)hi.x, (int)lo.x), __hiloint2double((int)hi.y, (int)lo.y));
  }
  return v;
}
__device__ __forceinline__ double wv_wmax(double v) { return wv_wred<true>(v); }
__device__ __forceinline__ double wv_wsum(double v) { return wv_wred<false>(v); }
#define WV_BIG (SCO_INFTY * SCO_MIN_SCALING)

// ---------------------------------------------------------------------------------------------------------------------
// factor kernel: value test, twisted block factorisation G_t = Sigma_t^-1, constants of the termination test
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WV_T) void qp_wv_factor_kernel(WvArgs a) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int b = a.list ? a.list[g + a.b0] : g + a.b0;
  if (b < 0) return;
  if (a.setup_active && !a.setup_active[b]) { if (lane == 0) a.rl_need[b] = 0; return; }
  const int n = a.n, m = a.m, bs = a.bs, nb = a.nb, mid = a.mid, NS = a.NS, NV = a.NV, NSTEP = a.NSTEP;
  const WvTab T = wv_tab_layout(NS, NV);
  const int *tab = a.tab;
  const double *ls = a.ls + (size_t)b * m, *us = a.us + (size_t)b * m, *rho = a.rhov + (size_t)b * m;
  const int *w = a.w + (size_t)b * m;
  const double *D = a.D + (size_t)b * n, *E = a.E + (size_t)b * m, *Ps = a.Ps + (size_t)b * a.nnzP;
  double *cst = a.cst + (size_t)b * a.cst_slots * 64;
  // ---- value structure of a penalty QP (else the row-local kernel takes the problem)
  int bad = 0, wk = 0, offd = 0;      // offd: P has entries inside a block off its diagonal
  for (int q = 0; q < NS; q++) { const int h = tab[(T.hrow + q) * 64 + lane]; if (h >= 0) wk = max(wk, w[h]); }
  wk = (int)wv_wmax((double)wk);
  for (int q = 0; q < NS; q++) {
    const int h = tab[(T.hrow + q) * 64 + lane], br = tab[(T.hbrow + q) * 64 + lane], ev = tab[(T.hevar + q) * 64 + lane];
    double eh = 0.0, eb = 0.0, de = 0.0;
    if (h >= 0) {
      if (!(ls[h] < -WV_BIG) || rho[h] != a.rho || w[h] != wk) bad = 1;
      if (!(us[br] > WV_BIG) || ls[br] != 0.0 || rho[br] != a.rho || w[br] != 1) bad = 1;
      eh = E[h]; eb = E[br]; de = D[ev];
    }
    double *c = cst + (size_t)wv_cst_h(q) * 64 + lane;
    c[0] = h >= 0 ? 1.0 / eh : 0.0; c[64] = h >= 0 ? 1.0 / eb : 0.0; c[128] = h >= 0 ? 1.0 / de : 0.0;
  }
  for (int v = 0; v < NV; v++) {
    const int var = tab[(T.vvar + v) * 64 + lane], r0 = tab[(T.vrow + v) * 64 + lane];
    double *c = cst + (size_t)wv_cst_v(NS, v) * 64 + lane;
    if (r0 >= 0 && (rho[r0] != a.rho || w[r0] != 1)) bad = 1;
    c[0] = var >= 0 ? 1.0 / D[var] : 0.0;
    c[64] = r0 >= 0 ? 1.0 / E[r0] : 0.0;
    const int pm = tab[(T.vpm + v) * 64 + lane], pp = tab[(T.vpp + v) * 64 + lane];
    c[2 * 64] = pm >= 0 ? Ps[pm] : 0.0; c[3 * 64] = pp >= 0 ? Ps[pp] : 0.0;
    const int kown = (tab[(T.vpk + v) * 64 + lane] / (2 * (2 * NSTEP + 2)) << 1) | (tab[(T.vpk + v) * 64 + lane] & 1);   // in-block index of the variable
    double pdiag = 0.0;
    for (int k = 0; k < 8; k++) {
      const int pd = tab[(T.vpd + v * 8 + k) * 64 + lane];
      const double pv = pd >= 0 ? Ps[pd] : 0.0;
      c[(5 + k) * 64] = pv;
      if (var >= 0 && k == kown) pdiag = pv; else if (var >= 0 && pv != 0.0) offd = 1;
    }
    c[4 * 64] = pdiag;
  }
  {
    const int xr = tab[T.xrow * 64 + lane];
    double *c = cst + (size_t)wv_cst_x(NS, NV) * 64 + lane;
    c[0] = xr >= 0 ? 1.0 / E[xr] : 0.0;
  }
  bad = (int)wv_wmax((double)bad); offd = (int)wv_wmax((double)offd);
  if (lane == 0) a.pflag[b] = offd;
  // (a failed value test: the dense inverse is formed right behind this kernel, so W will be ready)
  if (lane == 0) { a.ok[b] = !bad; a.rl_need[b] = bad; a.w_ready[b] = bad; a.wc[b] = (double)wk; }
  if (bad) return;
  // ---- twisted block factorisation.  Lane (i, j) = (lane >> 3, lane & 7) holds entry (i, j) of the current 8 x 8 block
  // (entries beyond bs: identity).
  const double *S = a.S + (size_t)b * a.n_c * a.n_c;
  double *G = a.G + (size_t)b * a.g_doubles;
  const int npos = 2 * NSTEP + 2;
  double *ef = G + (size_t)npos * 64, *en = ef + npos * 8, *em = en + npos * 8;
  for (size_t t = lane; t < a.g_doubles; t += WV_T) G[t] = 0.0;
  __threadfence_block(); __syncthreads();
  const int i = lane >> 3, j = lane & 7;
  auto Sblk = [&](int t) -> double {            // entry (i, j) of diagonal block t
    if (i >= bs || j >= bs) return i == j ? 1.0 : 0.0;
    const int ca = t * bs + max(i, j), cb = t * bs + min(i, j);
    return S[wv_tri(ca, cb)];
  };
  auto Soff = [&](int t, int k) -> double {     // coupling between blocks t and t + 1 at in-block index k
    return (k < bs && t >= 0 && t + 1 < nb) ? S[wv_tri((t + 1) * bs + k, t * bs + k)] : 0.0;
  };
  auto invert = [&](double M) -> double {       // Gauss-Jordan without pivoting (S is positive definite)
    for (int p = 0; p < 8; p++) {
      const double piv = __shfl(M, p * 8 + p), rowp = __shfl(M, p * 8 + j), colp = __shfl(M, i * 8 + p);
      const double inv = 1.0 / piv;
      if (i == p && j == p) M = inv;
      else if (i == p) M = rowp * inv;
      else if (j == p) M = -colp * inv;
      else M = M - colp * rowp * inv;
    }
    return M;
  };
  double Gmid_corr = 0.0;                        // E G E of the two chains' last blocks, subtracted from the middle block
  for (int chain = 0; chain < 2; chain++) {
    const int len = chain ? nb - 1 - mid : mid, p0 = chain ? 2 * NSTEP + 1 - len : NSTEP - len;
    double Gp = 0.0;
    for (int s = 0; s < len; s++) {
      const int t = chain ? nb - 1 - s : s, p = p0 + s;
      const int tl = chain ? t : t - 1;           // link (tl, tl + 1) joins t with its predecessor in the chain
      const double ei = s > 0 ? Soff(tl, i) : 0.0, ej = s > 0 ? Soff(tl, j) : 0.0;
      const double M = Sblk(t) - ei * Gp * ej;
      Gp = invert(M);
      G[wv_gidx(NSTEP, p, i, j)] = Gp;
      const int tn = chain ? t - 1 : t;           // link (tn, tn + 1) joins t with the next block towards the middle
      if (lane < 8) { ef[p * 8 + lane] = s > 0 ? Soff(tl, lane) : 0.0; en[p * 8 + lane] = Soff(tn, lane); }
    }
    if (len > 0) {
      const int tl = chain ? mid : mid - 1;       // link between the chain's last block and the middle block
      Gmid_corr += Soff(tl, i) * Gp * Soff(tl, j);
    }
  }
  {
    const double Gm = invert(Sblk(mid) - Gmid_corr);
    G[wv_gidx(NSTEP, NSTEP, i, j)] = Gm;
    if (lane < 8) { ef[NSTEP * 8 + lane] = Soff(mid - 1, lane); em[lane] = Soff(mid, lane); }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// ADMM kernel
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ double wv_min(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double wv_max(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// a row of G: BS doubles (the eighth of a 7-wide row is not loaded: its register would be free at once and the allocator
// reuses it for the next load, which then has to wait for this one)
template <int BS> struct WvRow { d2 a, b, c; d2 d; };
// p: piece 0 of the lane's row (pieces 0 .. 2 follow at 16-double steps), pd: its piece 3 (wrapped for chain B)
template <int BS>
__device__ __forceinline__ WvRow<BS> wv_row(const double *p, const double *pd) {
  WvRow<BS> r;
  r.a = *(const d2 *)p; r.b = *(const d2 *)(p + 16); r.c = *(const d2 *)(p + 32);
  if (BS > 7) r.d = *(const d2 *)pd; else { r.d.x = pd[0]; r.d.y = 0.0; }
  return r;
}

template <int BS>
__device__ __forceinline__ double wv_matvec(double acc, double w, const WvRow<BS> &g) {
  const d2 g0 = g.a, g1 = g.b, g2 = g.c, g3 = g.d;
  // two accumulators: a dependent v_fmac_f64_dpp issues every 8.5 cycles, two interleaved chains every 6.5
  // (scripts/microbench/wave_cost.hip).  The FIRST instruction of both chains sits in one asm statement behind the wait
  // states: the two chains are independent, so as separate statements the scheduler was free to put the second chain's
  // first broadcast directly behind the instruction that writes w (seen as wrong results of one build, r04).
  double acc2 = 0.0;
#if WV_VARIANT == 1
  WV_FMAC_DPP(acc, w, g0.x, 0); WV_FMAC_DPP(acc2, w, g0.y, 1);
#else
  asm("s_nop 1\n\tv_fmac_f64_dpp %0, %2, %3 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
      "v_fmac_f64_dpp %1, %2, %4 row_newbcast:1 row_mask:0xf bank_mask:0xf"
      : "+v"(acc), "+v"(acc2) : "v"(w), "v"(g0.x), "v"(g0.y));
#endif
  WV_FMAC_DPP(acc, w, g1.x, 2); WV_FMAC_DPP(acc2, w, g1.y, 3);
  WV_FMAC_DPP(acc, w, g2.x, 4); WV_FMAC_DPP(acc2, w, g2.y, 5); WV_FMAC_DPP(acc, w, g3.x, 6);
  if (BS > 7) WV_FMAC_DPP(acc2, w, g3.y, 7);
  return acc + acc2;
}

// Hand-over of LDS data between lanes of the ONE wavefront of a workgroup: the LDS unit works through a wavefront's
// instructions in issue order, so a read issued after a write sees it; all that is needed is that the compiler keeps the
// program order (a wavefront-scope fence emits no instruction).  __syncthreads() would add s_waitcnt lgkmcnt(0) + s_barrier
// four times per iteration.
#define WV_SYNC() do { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); } while (0)

// opaque copy of a pointer: loads through it cannot be hoisted out of the iteration loop (the constants of the termination
// test and the lane tables are read on checked iterations only; hoisted, they would occupy ~170 registers for the whole solve)
template <typename T>
__device__ __forceinline__ const T *wv_opaque(const T *p) { asm volatile("" : "+v"(p)); return p; }

template <int BS, int NS, int NV, int NSTEP, int LPB>
__global__ __launch_bounds__(WV_T) void qp_admm_wv_kernel(WvArgs a) {
  const int g = blockIdx.x, lane = threadIdx.x;
  const int b = a.list ? a.list[g + a.b0] : g + a.b0;
  if (b < 0 || (a.active && !a.active[b])) return;
  if (!a.ok[b]) return;                          // not a penalty-QP value structure: the row-local kernel solves it
  const int n = a.n, m = a.m, lpb = LPB > 0 ? LPB : a.lpb;
  constexpr int NPOS = 2 * NSTEP + 2;
  // LDS (doubles), compile-time offsets: the vectors sit at fixed distances from each other
  // The lane's Jacobian rows: constants of the solve, read twice per iteration.  In REGISTERS where the instantiation has
  // room (NS x BS <= 28 doubles: every instantiation built today), else lane-private in LDS: the kernel is bound by the CU's LDS
  // pipe (four wavefronts share it), and these rows were a third of a wavefront's LDS bytes per iteration.
  constexpr bool JREG = WV_JREG && NS * BS <= 28;
  constexpr int oG = 0, oEF = oG + NPOS * 64, oEN = oEF + NPOS * 8, oEM = oEN + NPOS * 8, oR = oEM + 8, oXT = oR + NPOS * 8,
                oXC = oXT + NPOS * 8, oEX = oXC + NPOS * 8, oJ = oEX + NPOS * 8, oPART = oJ + (JREG ? 0 : NS * 512);
  extern __shared__ __attribute__((aligned(16))) double lds[];
  const int *tab0 = a.tab;
  constexpr int oPOS = 0, oHROW = 1, oHBROW = oHROW + NS, oHEVAR = oHBROW + NS, oHEIDX = oHEVAR + NS, oHEPOS = oHEIDX + NS,
                oHBPOS = oHEPOS + NS, oHJPOS = oHBPOS + NS, oVVAR = oHJPOS + 8 * NS, oVROW = oVVAR + NV, oVPOS = oVROW + NV,
                oVPK = oVPOS + NV, oVPKM = oVPK + NV, oVPKP = oVPKM + NV, oXROW = oVPKP + NV + 8 * NV + 2 * NV, oXPOS = oXROW + 1,
                oXPK = oXPOS + 1;
  const double *As = a.As + (size_t)b * a.nnzA, *qs = a.qs + (size_t)b * n, *lsg = a.ls + (size_t)b * m, *usg = a.us + (size_t)b * m;
  const double *cstg0 = a.cst + (size_t)b * a.cst_slots * 64 + lane;
  const double rho0 = a.rho, rinv0 = 1.0 / rho0, wc = a.wc[b], rw = wc * rho0;
  const double sigma = a.sigma, alpha = a.alpha, oma = 1.0 - alpha;
  const double cscale = a.cscale[b], cinv = 1.0 / cscale;
  const int it0 = a.slice > 0 ? a.prog[b] : 0;
  const bool resume = it0 > 0;
  const double *sxg = a.sx + (size_t)b * n, *szg = a.sz + (size_t)b * m, *syg = a.sy + (size_t)b * m;

  // ---- LDS: factor, zeroed vectors
  {
    const double *Gg = a.G + (size_t)b * a.g_doubles;
    for (int t = lane; t < oR; t += WV_T) lds[t] = Gg[t];
    for (int t = lane; t < 4 * NPOS * 8; t += WV_T) lds[oR + t] = 0.0;
    for (int t = lane; t < NPOS * lpb * 8; t += WV_T) lds[oPART + t] = 0.0;
  }
  // ---- LDS: the constants of the termination test (scalings of the lane's rows and variables, its P entries), lane-minor.
  // They are constants of the solve; as registers (fetched one iteration ahead of every test) they cost 70 VGPRs for the
  // whole launch -- beyond 256 every use of a register is a v_accvgpr move.
  const int oCK = oPART + NPOS * lpb * 8;
  double *const ckl = lds + oCK + lane;
  {
#pragma unroll
    for (int q = 0; q < NS; q++)
#pragma unroll
      for (int k = 0; k < 3; k++) ckl[(3 * q + k) * 64] = cstg0[(wv_cst_h(q) + k) * 64];
#pragma unroll
    for (int v = 0; v < NV; v++)
#pragma unroll
      for (int k = 0; k < 5; k++) ckl[(3 * NS + 5 * v + k) * 64] = cstg0[(wv_cst_v(NS, v) + k) * 64];
    ckl[(3 * NS + 5 * NV) * 64] = cstg0[wv_cst_x(NS, NV) * 64];
  }
#define CKH(q, k) ckl[(3 * (q) + (k)) * 64]
#define CKV(v, k) ckl[(3 * NS + 5 * (v) + (k)) * 64]
#define CKX ckl[(3 * NS + 5 * NV) * 64]
  // ---- lane roles (row layout): pointers into the vectors of the lane's block
  const int pos_r = tab0[oPOS * 64 + lane];
  double *const xt_p = lds + oXT + pos_r * 2;                                   // x~ of the block, piece 0 (xc: + (oXC - oXT))
  const int nslot2 = NPOS * lpb * 2;                                            // doubles of one piece of the partial sums
  double *const part_p = lds + oPART + (pos_r * lpb + (lane & 15) % lpb) * 2;   // the lane's partial column sums, piece 0
  double *const jl_p = lds + oJ + lane * 2;                                     // Jacobian rows, lane-private
  // hinge slots: constants and state
  bool h_on[NS];
  double h_ae[NS], h_ab[NS], h_u[NS], h_q[NS], h_kinv[NS], h_z[NS], h_y[NS], h_zb[NS], h_yb[NS], h_xe[NS], h_ge[NS];
  double hJ[JREG ? NS : 1][8];
#pragma unroll
  for (int q = 0; q < NS; q++) {
    const int h = tab0[(oHROW + q) * 64 + lane], br = tab0[(oHBROW + q) * 64 + lane], ev = tab0[(oHEVAR + q) * 64 + lane];
    const bool on = h >= 0;
    h_on[q] = on;
    h_ae[q] = on ? As[tab0[(oHEPOS + q) * 64 + lane]] : 0.0;
    h_ab[q] = on ? As[tab0[(oHBPOS + q) * 64 + lane]] : 0.0;
    h_u[q] = on ? usg[h] : 0.0;
    h_q[q] = on ? qs[ev] : 0.0;
    h_kinv[q] = on ? a.kee_inv[(size_t)b * a.n_e + tab0[(oHEIDX + q) * 64 + lane]] : 0.0;
    h_z[q] = on && resume ? szg[h] : 0.0; h_y[q] = on && resume ? syg[h] : 0.0;
    h_zb[q] = on && resume ? szg[br] : 0.0; h_yb[q] = on && resume ? syg[br] : 0.0;
    h_xe[q] = on && resume ? sxg[ev] : 0.0; h_ge[q] = 0.0;
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int j0 = on ? tab0[(oHJPOS + q * 8 + 2 * k) * 64 + lane] : -1, j1 = on ? tab0[(oHJPOS + q * 8 + 2 * k + 1) * 64 + lane] : -1;
      d2 v; v.x = j0 >= 0 ? As[j0] : 0.0; v.y = j1 >= 0 ? As[j1] : 0.0;
      if (JREG) { hJ[q][2 * k] = v.x; hJ[q][2 * k + 1] = v.y; }
      else *(d2 *)(jl_p + (q * 4 + k) * 128) = v;
    }
  }
  // core-variable slots
  double *v_p[NV];                       // address of x~ of the variable (r, x, extra: fixed distances)
  const double *v_pp[NV];                // its component in the first partial-sum slot of its block
  bool v_on[NV], v_r0on[NV];
  int v_pkm[NV], v_pkp[NV];               // positions of the variable's neighbours in time inside a block vector (P entries)
  double v_x[NV], v_q[NV], v_a[NV], v_l[NV], v_u[NV], v_z[NV], v_y[NV];
#pragma unroll
  for (int v = 0; v < NV; v++) {
    const int var = tab0[(oVVAR + v) * 64 + lane], r0 = tab0[(oVROW + v) * 64 + lane];
    const int vix = tab0[(oVPK + v) * 64 + lane], vk = ((vix / (2 * NPOS)) << 1) | (vix & 1), vpos = (vix % (2 * NPOS)) >> 1;
    v_on[v] = var >= 0; v_p[v] = lds + oXT + vix;
    v_pp[v] = lds + oPART + (vk >> 1) * nslot2 + vpos * lpb * 2 + (vk & 1);
    v_q[v] = var >= 0 ? qs[var] : 0.0; v_x[v] = var >= 0 && resume ? sxg[var] : 0.0;
    v_a[v] = r0 >= 0 ? As[tab0[(oVPOS + v) * 64 + lane]] : 0.0;
    v_l[v] = r0 >= 0 ? lsg[r0] : 0.0; v_u[v] = r0 >= 0 ? usg[r0] : 0.0;
    v_z[v] = r0 >= 0 && resume ? szg[r0] : 0.0; v_y[v] = r0 >= 0 && resume ? syg[r0] : 0.0;
    v_r0on[v] = r0 >= 0; v_pkm[v] = tab0[(oVPKM + v) * 64 + lane]; v_pkp[v] = tab0[(oVPKP + v) * 64 + lane];
  }
  // extra single rows (a second row on a core variable: pins), one per lane
  const int x_row = tab0[oXROW * 64 + lane];
  double *const x_p = lds + oXT + tab0[oXPK * 64 + lane];
  const double x_a = x_row >= 0 ? As[tab0[oXPOS * 64 + lane]] : 0.0, x_l = x_row >= 0 ? lsg[x_row] : 0.0, x_u = x_row >= 0 ? usg[x_row] : 0.0;
  const double x_rho = x_row >= 0 ? a.rhov[(size_t)b * m + x_row] : 1.0, x_rinv = 1.0 / x_rho, x_w = x_row >= 0 ? (double)a.w[(size_t)b * m + x_row] : 0.0;
  double x_z = x_row >= 0 && resume ? szg[x_row] : 0.0, x_y = x_row >= 0 && resume ? syg[x_row] : 0.0;
  // sweep roles: DPP row 0 (and its copy, row 2) runs chain A, row 1 (and 3) chain B; lane k of a row holds component k
  const int srow = lane >> 4, k8 = lane & 7, chain = srow & 1;
  const bool sw_store = srow < 2 && (lane & 15) < 8;
  const int pos0 = chain ? NSTEP + 1 : 0;
  const double *const sw_g = lds + oG + pos0 * 64 + k8 * 2 + (chain ? 16 : 0);        // row k8 of the chain's first block, piece 0
  const double *const sw_gd = lds + oG + pos0 * 64 + k8 * 2 + (chain ? 0 : 48);       // ... its piece 3
  double *const sw_v = lds + oR + wv_vidx(NPOS, pos0, k8);                              // r of the chain's first block (next block: + 2)
  const double *const sw_e = lds + oEF + pos0 * 8 + k8;                                // couplings of the chain's first block
  const double *const md_g = lds + oG + NSTEP * 64 + k8 * 2;
  double *const md_v = lds + oR + wv_vidx(NPOS, NSTEP, k8);
  const double *const md_e = lds + oEF + NSTEP * 8 + k8;
  WV_SYNC();

  // accumulators of the termination test that ride along in the checked step
  double c_ndy = 0.0, c_lhs = 0.0, c_ndx = 0.0, c_qdx = 0.0;
  // Row / variable indices and scaling constants that only a checked iteration needs.  They are fetched from global memory
  // at the START of that iteration, in front of the sweeps, so that the round trip (L2 or HBM: 68 x 512 B per problem) runs
  // under the iteration's own arithmetic; kept for the whole solve they would cost ~170 registers.
  const int pdense = (a.ablate & 32) ? 1 : a.pflag[b];         // P has entries off the three diagonals the compact constants hold
  // delta_y (clipped) / delta_x of the checked iteration: kept in registers for the infeasibility certificates
  struct { double h[NS], b[NS], e[NS], r0[NV], var[NV], x; } dsv;
  // One pass over the lane's rows and variables.  MODE 0: initialise (right-hand side of the first iteration from the
  // current x, z, y); 1: a plain iteration; 2: a checked iteration (delta_y / delta_x terms of the infeasibility tests).
  auto rows = [&](auto mode_tag) {
    constexpr int MODE = decltype(mode_tag)::value;
    // every LDS read of this phase is issued before its first store (the compiler keeps loads behind stores that may alias)
    double xt[8], xtx = 0.0, xtv[NV];
    if (MODE) {
      const d2 a0 = *(const d2 *)xt_p, a1 = *(const d2 *)(xt_p + 2 * NPOS), a2 = *(const d2 *)(xt_p + 4 * NPOS), a3 = *(const d2 *)(xt_p + 6 * NPOS);
      xtx = x_p[0];
#pragma unroll
      for (int v = 0; v < NV; v++) xtv[v] = v_p[v][0];
      xt[0] = a0.x; xt[1] = a0.y; xt[2] = a1.x; xt[3] = a1.y; xt[4] = a2.x; xt[5] = a2.y; xt[6] = a3.x; xt[7] = a3.y;
    }
    double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    d2 jn0, jn1, jn2, jn3;                       // (rows in LDS) the NEXT slot's row is in flight while this one's arithmetic runs
    if (!JREG) { const d2 *jp = (const d2 *)jl_p; jn0 = jp[0]; jn1 = jp[64]; jn2 = jp[128]; jn3 = jp[192]; }
#pragma unroll
    for (int q = 0; q < NS; q++) {
      double J[8];
      if (JREG) {
#pragma unroll
        for (int k = 0; k < 8; k++) J[k] = hJ[q][k];
      } else {
        J[0] = jn0.x; J[1] = jn0.y; J[2] = jn1.x; J[3] = jn1.y; J[4] = jn2.x; J[5] = jn2.y; J[6] = jn3.x; J[7] = jn3.y;
        if (q + 1 < NS) { const d2 *jp = (const d2 *)(jl_p + (q + 1) * 512); jn0 = jp[0]; jn1 = jp[64]; jn2 = jp[128]; jn3 = jp[192]; }
      }
      const double ae = h_ae[q], ab = h_ab[q], kinv = h_kinv[q];
      const double c2 = rw * ae;
      if (MODE) {
        double s = J[0] * xt[0];
#pragma unroll
        for (int k = 1; k < BS; k++) s = __builtin_fma(J[k], xt[k], s);
        const double xte = h_ge[q] - (kinv * c2) * s;
        const double zth = __builtin_fma(ae, xte, s), ztb = ab * xte;
        // hinge row: l = -inf
        {
          const double zr = alpha * zth + oma * h_z[q];
          const double zn = wv_min(zr + rinv0 * h_y[q], h_u[q]);
          const double dy = rho0 * (zr - zn);
          h_y[q] += dy; h_z[q] = zn;
          if (MODE == 2) {
            const double dc = fmax(dy, 0.0);              // clipped to the cone of the bounds (u finite, l = -inf)
            c_ndy = fmax(c_ndy, wv_recip(CKH(q, 0)) * fabs(dc)); c_lhs += wc * (h_u[q] * dc);
            dsv.h[q] = dc;
          }
        }
        // slack's bound row: [0, inf)
        {
          const double zr = alpha * ztb + oma * h_zb[q];
          const double zn = wv_max(zr + rinv0 * h_yb[q], 0.0);
          const double dy = rho0 * (zr - zn);
          h_yb[q] += dy; h_zb[q] = zn;
          if (MODE == 2) {
            const double dc = fmin(dy, 0.0);
            c_ndy = fmax(c_ndy, wv_recip(CKH(q, 1)) * fabs(dc));           // l = 0: nothing for u' dy+ + l' dy-
            dsv.b[q] = dc;
          }
        }
        const double xo = h_xe[q], xn = alpha * xte + oma * xo;
        h_xe[q] = xn;
        if (MODE == 2) {
          const double dx = xn - xo;
          c_ndx = fmax(c_ndx, wv_recip(CKH(q, 2)) * fabs(dx)); c_qdx += h_q[q] * dx;
          dsv.e[q] = dx;
        }
      }
      const double th = wc * (rho0 * h_z[q] - h_y[q]);
      const double tb = rho0 * h_zb[q] - h_yb[q];
      const double rhs = ae * th + ab * tb + (sigma * h_xe[q] - h_q[q]);
      const double ge = rhs * kinv;
      h_ge[q] = ge;
      const double tp = th - c2 * ge;
#pragma unroll
      for (int k = 0; k < BS; k++) part[k] = __builtin_fma(J[k], tp, part[k]);
    }
    {
      d2 v0, v1, v2, v3;
      v0.x = part[0]; v0.y = part[1]; v1.x = part[2]; v1.y = part[3]; v2.x = part[4]; v2.y = part[5]; v3.x = part[6]; v3.y = BS > 7 ? part[7] : 0.0;
      *(d2 *)part_p = v0; *(d2 *)(part_p + nslot2) = v1; *(d2 *)(part_p + 2 * nslot2) = v2; *(d2 *)(part_p + 3 * nslot2) = v3;
    }
    // extra rows
    {
      if (MODE) {
        const double xtc = xtx;
        const double zt = x_a * xtc, zr = alpha * zt + oma * x_z;
        const double zn = wv_min(wv_max(zr + x_rinv * x_y, x_l), x_u);
        const double dy = x_rho * (zr - zn);
        x_y += dy; x_z = zn;
        if (MODE == 2) {
          const double d1 = x_u > WV_BIG ? fmin(dy, 0.0) : dy, dc = x_l < -WV_BIG ? fmax(d1, 0.0) : d1;
          c_ndy = fmax(c_ndy, wv_recip(CKX) * fabs(dc)); c_lhs += x_w * (x_u * fmax(dc, 0.0) + x_l * fmin(dc, 0.0));
          dsv.x = dc;
        }
      }
      const double t = x_w * (x_rho * x_z - x_y);
      if (x_row >= 0) x_p[oEX - oXT] = x_a * t;
    }
    // core variables: box row, x update, own part of the right-hand side
    double own[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      if (MODE) {
        const double xtc = xtv[v];
        const double zt = v_a[v] * xtc, zr = alpha * zt + oma * v_z[v];
        const double zn = wv_min(wv_max(zr + rinv0 * v_y[v], v_l[v]), v_u[v]);
        const double dy = rho0 * (zr - zn);
        v_y[v] += dy; v_z[v] = zn;
        const double xo = v_x[v], xn = alpha * xtc + oma * xo;
        v_x[v] = xn;
        if (MODE == 2) {
          const double d1 = v_u[v] > WV_BIG ? fmin(dy, 0.0) : dy, dc = v_l[v] < -WV_BIG ? fmax(d1, 0.0) : d1;
          if (v_r0on[v]) { c_ndy = fmax(c_ndy, wv_recip(CKV(v, 1)) * fabs(dc)); c_lhs += v_u[v] * fmax(dc, 0.0) + v_l[v] * fmin(dc, 0.0); }
          dsv.r0[v] = dc;
          const double dx = xn - xo;
          if (v_on[v]) { c_ndx = fmax(c_ndx, wv_recip(CKV(v, 0)) * fabs(dx)); c_qdx += v_q[v] * dx; }
          dsv.var[v] = dx;
        }
      }
      const double t0 = rho0 * v_z[v] - v_y[v];
      own[v] = v_a[v] * t0 + (sigma * v_x[v] - v_q[v]);
    }
    WV_SYNC();
    // core right-hand side = own part + the extra row's + the block's partial column sums
    double rsum[NV];
#pragma unroll
    for (int v = 0; v < NV; v++) {
      double r = own[v] + v_p[v][oEX - oXT];
      const double *pp = v_pp[v];
      if (LPB > 0) {
#pragma unroll
        for (int s2 = 0; s2 < LPB; s2++) r += pp[s2 * 2];
      } else {
        for (int s2 = 0; s2 < lpb; s2++) r += pp[s2 * 2];
      }
      rsum[v] = r;
    }
#pragma unroll
    for (int v = 0; v < NV; v++) {
      v_p[v][oR - oXT] = v_on[v] ? rsum[v] : 0.0;
      if (MODE == 2) v_p[v][oXC - oXT] = v_x[v];
    }
    WV_SYNC();
  };

  // x~_C = S^-1 r by the twisted block factorisation: forward sweeps of both chains, the middle block, backward sweeps.
  // Nothing is stored inside the sweeps: the NSTEP intermediate vectors of a chain stay in registers.
  auto sweep = [&]() {
    // only the 16 lanes that hold a component of one of the two chains take part: the others would fetch the same rows of
    // G again (four wavefronts share a CU's LDS pipe).
    // A block step is ~60 cycles of arithmetic (seven dependent v_fmac_f64_dpp in two chains, one fma, one add) and an LDS
    // read comes back after ~130 (scripts/microbench/wave_cost.hip): the rows of G, the right-hand sides and the couplings
    // are asked for WV_PD block steps ahead (r04: one step ahead left the sweeps waiting for LDS, 176 cycles per step).
    constexpr int PD = WV_PD < NSTEP ? WV_PD : NSTEP;
    // The sweeps are bound by the CU's LDS pipe, not by their arithmetic (r04: plain v_fmac_f64 in place of the DPP broadcasts
    // changes nothing, scripts/build_ablate.py wv:WV_VARIANT=1; four wavefronts x 21 steps x 16 lanes x 64 B of G per pass):
    // rows of G the forward pass has read stay in registers for the backward pass (16 VGPRs per block) where the
    // instantiation has room.
    constexpr int KEEPN = NSTEP <= 10 ? (WV_KEEP < NSTEP ? WV_KEEP : NSTEP) : 0;      // the forward pass's last KEEPN blocks
    constexpr int K0 = NSTEP - KEEPN;
    WvRow<BS> gk[KEEPN > 0 ? KEEPN : 1];
    double vs[NSTEP];
    WvRow<BS> gq[PD]; double rq[PD], eq[PD];
    double emk = 0.0, vlast = 0.0;
    if (sw_store) {
      auto fetch_fwd = [&](int st, WvRow<BS> &g, double &r, double &e) {      // block st of the chain; st = NSTEP: the middle block
        if (st < NSTEP) { g = wv_row<BS>(sw_g + st * 64, sw_gd + st * 64); r = sw_v[st * 2]; e = sw_e[st * 8]; }
        else if (st == NSTEP) { g = wv_row<BS>(md_g, md_g + 48); r = md_v[0]; e = md_e[0]; }
      };
#pragma unroll
      for (int i = 0; i < PD; i++) fetch_fwd(i, gq[i], rq[i], eq[i]);
      double vprev = 0.0;
#pragma unroll
      for (int st = 0; st < NSTEP; st++) {
        const WvRow<BS> g = gq[st % PD]; const double rr = rq[st % PD], ee = eq[st % PD];
        fetch_fwd(st + PD, gq[st % PD], rq[st % PD], eq[st % PD]);
        const double w = __builtin_fma(-ee, vprev, rr);
        vprev = wv_matvec<BS>(0.0, w, g);
        vs[st] = vprev;
        if (st >= K0) gk[st - K0] = g;
      }
      emk = lds[oEM + k8];
      vlast = vs[NSTEP - 1];
    }
    // the middle block needs the last vector of BOTH chains: DPP row 0 holds chain A's, row 1 chain B's.  One
    // v_permlane16_swap per register half hands each row the other's (through LDS the exchange was a store, a wait and a
    // read: ~250 cycles of the ~3000 of a sweep).  All lanes take part in the swap; only the sweep lanes use the result.
    double vother;
    {
      const wv_u2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(vlast), (unsigned)__double2loint(vlast), false, false);
      const wv_u2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(vlast), (unsigned)__double2hiint(vlast), false, false);
      // .x: odd rows now hold the even rows' values; .y: even rows now hold the odd rows' values
      vother = chain ? __hiloint2double((int)hi.x, (int)lo.x) : __hiloint2double((int)hi.y, (int)lo.y);
    }
    if (sw_store) {
      // (slot NSTEP % PD of the ring holds the middle block: fetched PD steps before the end of the forward sweep)
      const WvRow<BS> gm = gq[NSTEP % PD]; const double rm = rq[NSTEP % PD], em_ = eq[NSTEP % PD];
      const double vA = chain ? vother : vlast, vB = chain ? vlast : vother;
      // backward sweep: block st of the chain and its coupling towards the middle, again PD steps ahead
      WvRow<BS> gb[PD]; double eb[PD];
      auto fetch_bwd = [&](int st, WvRow<BS> &g, double &e) {
        if (st >= 0) { if (st < K0) g = wv_row<BS>(sw_g + st * 64, sw_gd + st * 64); e = sw_e[st * 8 + (oEN - oEF)]; }
      };
#pragma unroll
      for (int i = 0; i < PD; i++) fetch_bwd(NSTEP - 1 - i, gb[i], eb[i]);
      double w = __builtin_fma(-em_, vA, rm);
      w = __builtin_fma(-emk, vB, w);
      double xn = wv_matvec<BS>(0.0, w, gm);
      const double xmid = xn;
#pragma unroll
      for (int st = NSTEP - 1; st >= 0; st--) {
        const int slot = (NSTEP - 1 - st) % PD;
        const WvRow<BS> g = st >= K0 ? gk[st - K0] : gb[slot]; const double ee = eb[slot];
        fetch_bwd(st - PD, gb[slot], eb[slot]);
        const double u = -(ee * xn);
        xn = wv_matvec<BS>(vs[st], u, g);
        vs[st] = xn;
      }
#pragma unroll
      for (int st = 0; st < NSTEP; st++) sw_v[st * 2 + (oXT - oR)] = vs[st];
      if (srow == 0) md_v[oXT - oR] = xmid;
    }
    WV_SYNC();
  };

  rows(std::integral_constant<int, 0>{});

  int status = 0, iter = it0;
  double pri = 0.0, dua = 0.0;
  const int stop = (a.slice > 0 && it0 + a.slice < a.max_iter) ? it0 + a.slice : a.max_iter;
  while (!status && iter < stop) {
    int next = stop;
    if (a.check > 0) { next = (iter / a.check + 1) * a.check; if (next > stop) next = stop; }
    while (iter + 2 < next) { iter++; if (!(a.ablate & 1)) sweep(); if (!(a.ablate & 2)) rows(std::integral_constant<int, 1>{}); }
    if (iter + 1 < next) { iter++; if (!(a.ablate & 1)) sweep(); if (!(a.ablate & 2)) rows(std::integral_constant<int, 1>{}); }
    iter++;
    c_ndy = 0.0; c_lhs = 0.0; c_ndx = 0.0; c_qdx = 0.0;
    sweep(); rows(std::integral_constant<int, 2>{});
    if (a.ablate & 16) continue;
    // ---- termination test (formulas of admm_check in sco_qp.hip) on the structured layout
    for (int approximate = 0; approximate < 2 && !status; approximate++) {
      if (approximate && iter < a.max_iter) break;
      double ea = a.eps_abs, er = a.eps_rel, epi = a.eps_prim_inf, edi = a.eps_dual_inf;
      if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
      double w_pri = 0.0, w_pn = 0.0, w_dua = 0.0, w_dn = 0.0;
      // rows of the hinge slots; partial sums of A' (w y) over the block's columns
      {
        const double *xq = xt_p + (oXC - oXT);
        const d2 a0 = *(const d2 *)xq, a1 = *(const d2 *)(xq + 2 * NPOS), a2 = *(const d2 *)(xq + 4 * NPOS), a3 = *(const d2 *)(xq + 6 * NPOS);
        const double xc[8] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y};
        double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < NS; q++) {
          double J[8];
          if (JREG) {
#pragma unroll
            for (int k = 0; k < 8; k++) J[k] = hJ[q][k];
          } else {
            const d2 *jp = (const d2 *)(jl_p + q * 512);
            const d2 j0 = jp[0], j1 = jp[64], j2 = jp[128], j3 = jp[192];
            J[0] = j0.x; J[1] = j0.y; J[2] = j1.x; J[3] = j1.y; J[4] = j2.x; J[5] = j2.y; J[6] = j3.x; J[7] = j3.y;
          }
          double s = J[0] * xc[0];
#pragma unroll
          for (int k = 1; k < BS; k++) s = __builtin_fma(J[k], xc[k], s);
          const double ax = s + h_ae[q] * h_xe[q], axb = h_ab[q] * h_xe[q];
          const double eh = CKH(q, 0), eb = CKH(q, 1), de = CKH(q, 2);
          w_pri = fmax(w_pri, fmax(eh * fabs(ax - h_z[q]), eb * fabs(axb - h_zb[q])));
          w_pn = fmax(w_pn, fmax(eh * fmax(fabs(h_z[q]), fabs(ax)), eb * fmax(fabs(h_zb[q]), fabs(axb))));
          const double wy = wc * h_y[q];
          const double aty = h_ae[q] * wy + h_ab[q] * h_yb[q];
          w_dua = fmax(w_dua, de * fabs(h_q[q] + aty));
          w_dn = fmax(w_dn, de * fmax(fabs(h_q[q]), fabs(aty)));
#pragma unroll
          for (int k = 0; k < BS; k++) part[k] = __builtin_fma(J[k], wy, part[k]);
        }
        d2 v0, v1, v2, v3;
        v0.x = part[0]; v0.y = part[1]; v1.x = part[2]; v1.y = part[3]; v2.x = part[4]; v2.y = part[5]; v3.x = part[6]; v3.y = BS > 7 ? part[7] : 0.0;
        *(d2 *)part_p = v0; *(d2 *)(part_p + nslot2) = v1; *(d2 *)(part_p + 2 * nslot2) = v2; *(d2 *)(part_p + 3 * nslot2) = v3;
      }
      {
        const double xcv = x_p[oXC - oXT], ax = x_a * xcv, ex = CKX;
        w_pri = fmax(w_pri, ex * fabs(ax - x_z)); w_pn = fmax(w_pn, ex * fmax(fabs(x_z), fabs(ax)));
        if (x_row >= 0) x_p[oEX - oXT] = x_a * (x_w * x_y);
      }
      WV_SYNC();
#pragma unroll
      for (int v = 0; v < NV; v++) {
        const double c[5] = {CKV(v, 0), CKV(v, 1), CKV(v, 2), CKV(v, 3), CKV(v, 4)};
        const int vix = (int)(v_p[v] - (lds + oXT)), p = (vix % (2 * NPOS)) >> 1;
        const double dj = c[0], e0 = c[1];
        const double ax = v_a[v] * v_x[v];
        w_pri = fmax(w_pri, e0 * fabs(ax - v_z[v])); w_pn = fmax(w_pn, e0 * fmax(fabs(v_z[v]), fabs(ax)));
        double aty = v_a[v] * v_y[v] + v_p[v][oEX - oXT];
        const double *pp = v_pp[v];
        if (LPB > 0) {
#pragma unroll
          for (int s2 = 0; s2 < LPB; s2++) aty += pp[s2 * 2];
        } else {
          for (int s2 = 0; s2 < lpb; s2++) aty += pp[s2 * 2];
        }
        double px = c[2] * lds[oXC + v_pkm[v]] + c[3] * lds[oXC + v_pkp[v]];
        if (pdense) {
          const double *cd = wv_opaque(cstg0) + (size_t)(wv_cst_v(NS, v) + 5) * 64;
#pragma unroll
          for (int k2 = 0; k2 < BS; k2++) px = __builtin_fma(cd[k2 * 64], lds[oXC + wv_vidx(NPOS, p, k2)], px);
        } else {
          px = __builtin_fma(c[4], v_x[v], px);
        }
        if (v_on[v]) {
          w_dua = fmax(w_dua, dj * fabs(v_q[v] + px + aty));
          w_dn = fmax(w_dn, dj * fmax(fabs(v_q[v]), fmax(fabs(aty), fabs(px))));
        }
      }
      WV_SYNC();       // (the next pass over the rows rewrites the two buffers the test has borrowed)
      w_pri = wv_wmax(w_pri); w_pn = wv_wmax(w_pn); w_dua = wv_wmax(w_dua); w_dn = wv_wmax(w_dn);
      const double ndy = wv_wmax(c_ndy), ndx = wv_wmax(c_ndx), lhs = wv_wsum(c_lhs), qdx = wv_wsum(c_qdx);
      pri = w_pri; dua = cinv * w_dua;
      if (!(pri <= SCO_INFTY) || !(dua <= SCO_INFTY)) { status = SCO_QP_NON_CVX; break; }
      const double eps_p = ea + er * w_pn, eps_d = ea + er * cinv * w_dn;
      const bool prim_ok = (m == 0) || (pri < eps_p), dual_ok = dua < eps_d;
      if (prim_ok && dual_ok) { status = approximate ? SCO_QP_SOLVED_INACCURATE : SCO_QP_SOLVED; break; }
      // The two infeasibility certificates, on the structured layout as well (r04: stiff penalty QPs -- the compounded
      // penalty of quirk Q1 -- meet their cheap preconditions at most checks; as generic loops over the CSC pattern in
      // global memory they cost 8 % of the solve, scripts/gpu_wv_time.py).  Same quantities as admm_check in sco_qp.hip:
      //   primal:  || D^-1 A' (w dy) ||inf < eps_prim_inf || E dy ||inf                  (dy clipped to the cone of the bounds)
      //   dual:    || D^-1 P dx ||inf < c eps_dual_inf || D dx ||inf  and no row of E^-1 A dx leaves its finite bounds' cone
      // delta_y / delta_x of the checked iteration are in registers (dsv); column sums go through the two buffers the
      // test above has borrowed already.
      if ((a.ablate & 64) == 0 && !prim_ok && ndy > epi && lhs < -epi * ndy) {
        double nat = 0.0;
        {
          double part[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
          for (int q = 0; q < NS; q++) {
            double J[8];
            if (JREG) {
#pragma unroll
              for (int k = 0; k < 8; k++) J[k] = hJ[q][k];
            } else {
              const d2 *jp = (const d2 *)(jl_p + q * 512);
              const d2 j0 = jp[0], j1 = jp[64], j2 = jp[128], j3 = jp[192];
              J[0] = j0.x; J[1] = j0.y; J[2] = j1.x; J[3] = j1.y; J[4] = j2.x; J[5] = j2.y; J[6] = j3.x; J[7] = j3.y;
            }
            const double wd = wc * dsv.h[q];
            nat = fmax(nat, CKH(q, 2) * fabs(h_ae[q] * wd + h_ab[q] * dsv.b[q]));      // the slack's column
#pragma unroll
            for (int k = 0; k < BS; k++) part[k] = __builtin_fma(J[k], wd, part[k]);
          }
          d2 v0, v1, v2, v3;
          v0.x = part[0]; v0.y = part[1]; v1.x = part[2]; v1.y = part[3]; v2.x = part[4]; v2.y = part[5]; v3.x = part[6]; v3.y = BS > 7 ? part[7] : 0.0;
          *(d2 *)part_p = v0; *(d2 *)(part_p + nslot2) = v1; *(d2 *)(part_p + 2 * nslot2) = v2; *(d2 *)(part_p + 3 * nslot2) = v3;
        }
        if (x_row >= 0) x_p[oEX - oXT] = x_a * (x_w * dsv.x);
        WV_SYNC();
#pragma unroll
        for (int v = 0; v < NV; v++) {
          double aty = v_a[v] * dsv.r0[v] + v_p[v][oEX - oXT];
          const double *pp = v_pp[v];
          if (LPB > 0) {
#pragma unroll
            for (int s2 = 0; s2 < LPB; s2++) aty += pp[s2 * 2];
          } else {
            for (int s2 = 0; s2 < lpb; s2++) aty += pp[s2 * 2];
          }
          if (v_on[v]) nat = fmax(nat, CKV(v, 0) * fabs(aty));
        }
        WV_SYNC();
        nat = wv_wmax(nat);
        if (nat < epi * ndy) { status = approximate ? SCO_QP_PRIMAL_INFEASIBLE_INACCURATE : SCO_QP_PRIMAL_INFEASIBLE; break; }
      }
      if ((a.ablate & 64) == 0 && !dual_ok && ndx > edi && qdx < -cscale * edi * ndx) {
        // delta_x of the core variables takes the place of x in its block vector (the next checked pass rewrites it)
#pragma unroll
        for (int v = 0; v < NV; v++) v_p[v][oXC - oXT] = dsv.var[v];
        WV_SYNC();
        double npx = 0.0;
#pragma unroll
        for (int v = 0; v < NV; v++) {
          const double c[5] = {CKV(v, 0), CKV(v, 1), CKV(v, 2), CKV(v, 3), CKV(v, 4)};
          const int vix = (int)(v_p[v] - (lds + oXT)), p = (vix % (2 * NPOS)) >> 1;
          double px = c[2] * lds[oXC + v_pkm[v]] + c[3] * lds[oXC + v_pkp[v]];
          if (pdense) {
            const double *cd = wv_opaque(cstg0) + (size_t)(wv_cst_v(NS, v) + 5) * 64;
#pragma unroll
            for (int k2 = 0; k2 < BS; k2++) px = __builtin_fma(cd[k2 * 64], lds[oXC + wv_vidx(NPOS, p, k2)], px);
          } else {
            px = __builtin_fma(c[4], dsv.var[v], px);
          }
          if (v_on[v]) npx = fmax(npx, c[0] * fabs(px));
        }
        npx = wv_wmax(npx);
        if (npx < cscale * edi * ndx) {
          const double thr = edi * ndx;
          double badv = 0.0;
          const double *xq = xt_p + (oXC - oXT);
          const d2 a0 = *(const d2 *)xq, a1 = *(const d2 *)(xq + 2 * NPOS), a2 = *(const d2 *)(xq + 4 * NPOS), a3 = *(const d2 *)(xq + 6 * NPOS);
          const double dxb[8] = {a0.x, a0.y, a1.x, a1.y, a2.x, a2.y, a3.x, a3.y};
#pragma unroll
          for (int q = 0; q < NS; q++) {
            double J[8];
            if (JREG) {
#pragma unroll
              for (int k = 0; k < 8; k++) J[k] = hJ[q][k];
            } else {
              const d2 *jp = (const d2 *)(jl_p + q * 512);
              const d2 j0 = jp[0], j1 = jp[64], j2 = jp[128], j3 = jp[192];
              J[0] = j0.x; J[1] = j0.y; J[2] = j1.x; J[3] = j1.y; J[4] = j2.x; J[5] = j2.y; J[6] = j3.x; J[7] = j3.y;
            }
            double s2 = J[0] * dxb[0];
#pragma unroll
            for (int k = 1; k < BS; k++) s2 = __builtin_fma(J[k], dxb[k], s2);
            const double adh = CKH(q, 0) * (s2 + h_ae[q] * dsv.e[q]), adb = CKH(q, 1) * (h_ab[q] * dsv.e[q]);
            if (h_on[q] && ((h_u[q] < WV_BIG && adh > thr) || adb < -thr)) badv = 1.0;     // hinge: l = -inf; its slack's bound row: [0, inf)
          }
#pragma unroll
          for (int v = 0; v < NV; v++) {
            const double adx = CKV(v, 1) * (v_a[v] * dsv.var[v]);
            if (v_r0on[v] && ((v_u[v] < WV_BIG && adx > thr) || (v_l[v] > -WV_BIG && adx < -thr))) badv = 1.0;
          }
          if (x_row >= 0) {
            const double adx = CKX * (x_a * x_p[oXC - oXT]);
            if ((x_u < WV_BIG && adx > thr) || (x_l > -WV_BIG && adx < -thr)) badv = 1.0;
          }
          badv = wv_wmax(badv);
          if (badv == 0.0) { status = approximate ? SCO_QP_DUAL_INFEASIBLE_INACCURATE : SCO_QP_DUAL_INFEASIBLE; break; }
        }
        WV_SYNC();
#pragma unroll
        for (int v = 0; v < NV; v++) v_p[v][oXC - oXT] = v_x[v];       // x back in its place (the test at max_iter runs twice)
        WV_SYNC();
      }
    }
  }
  if (!status && iter < a.max_iter) {
    // the slice is used up: park the solve (scaled x, z, y in natural order; t', g_e and the right-hand side are rebuilt
    // from them by the next launch with the same formulas, i.e. bit for bit).  t' and g_e are written too, in the form
    // the row-local kernel resumes from (sco_admm_rl.hip: t'_i = t_i - rw_i a_ie g_e): the SQP loop hands the tail of a
    // step, when fewer problems are alive than this tier needs to fill the chip, to that kernel.
    double *sx = a.sx + (size_t)b * n, *sz = a.sz + (size_t)b * m, *sy = a.sy + (size_t)b * m;
    double *stp = a.st + (size_t)b * m, *sgp = a.sg + (size_t)b * a.n_e;
    const int *tab = wv_opaque(tab0);
#pragma unroll
    for (int q = 0; q < NS; q++) {
      const int h = tab[(oHROW + q) * 64 + lane], br = tab[(oHBROW + q) * 64 + lane], ev = tab[(oHEVAR + q) * 64 + lane];
      if (h >= 0) {
        sz[h] = h_z[q]; sy[h] = h_y[q]; sz[br] = h_zb[q]; sy[br] = h_yb[q]; sx[ev] = h_xe[q];
        const double th = wc * (rho0 * h_z[q] - h_y[q]), tb = rho0 * h_zb[q] - h_yb[q];
        stp[h] = th - (rw * h_ae[q]) * h_ge[q]; stp[br] = tb - (rho0 * h_ab[q]) * h_ge[q];
        sgp[tab[(oHEIDX + q) * 64 + lane]] = h_ge[q];
      }
    }
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int var = tab[(oVVAR + v) * 64 + lane], r0 = tab[(oVROW + v) * 64 + lane];
      if (var >= 0) sx[var] = v_x[v];
      if (r0 >= 0) { sz[r0] = v_z[v]; sy[r0] = v_y[v]; stp[r0] = rho0 * v_z[v] - v_y[v]; }
    }
    if (x_row >= 0) { sz[x_row] = x_z; sy[x_row] = x_y; stp[x_row] = x_w * (x_rho * x_z - x_y); }
    if (lane == 0) { a.prog[b] = iter; a.status[b] = 0; a.iters[b] = iter; atomicAdd(a.it_count, (unsigned long long)(iter - it0)); }
    return;
  }
  if (a.slice > 0 && lane == 0) a.prog[b] = 0;
  if (!status) status = SCO_QP_MAX_ITER_REACHED;
  if (iter > a.max_iter) iter = a.max_iter;
  {
    const double *Dg = a.D + (size_t)b * n, *Eg = a.E + (size_t)b * m;
    double *xo = a.x + (size_t)b * n, *yo = a.y + (size_t)b * m;
    const int *tab = wv_opaque(tab0);
#pragma unroll
    for (int q = 0; q < NS; q++) {
      const int h = tab[(oHROW + q) * 64 + lane], br = tab[(oHBROW + q) * 64 + lane], ev = tab[(oHEVAR + q) * 64 + lane];
      if (h >= 0) {
        xo[ev] = Dg[ev] * h_xe[q];
        yo[h] = cinv * Eg[h] * h_y[q] * wc;
        yo[br] = cinv * Eg[br] * h_yb[q];
      }
    }
#pragma unroll
    for (int v = 0; v < NV; v++) {
      const int var = tab[(oVVAR + v) * 64 + lane], r0 = tab[(oVROW + v) * 64 + lane];
      if (var >= 0) xo[var] = Dg[var] * v_x[v];
      if (r0 >= 0) yo[r0] = cinv * Eg[r0] * v_y[v];
    }
    if (x_row >= 0) yo[x_row] = cinv * Eg[x_row] * x_y * x_w;
    if (lane == 0) {
      a.status[b] = status; a.iters[b] = iter; a.resid[2 * (size_t)b] = pri; a.resid[2 * (size_t)b + 1] = dua;
      atomicAdd(a.it_count, (unsigned long long)(iter - it0));
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// launch
// ---------------------------------------------------------------------------------------------------------------------
static void wv_fill_args(const AdmmArgs &aa, const WvHost &wh, const WvDev &wd, const int *setup_mask, WvArgs &a) {
  const QpDev &d = aa.d;
  a.n = d.n; a.m = d.m; a.n_e = d.n_e; a.n_c = d.n_c; a.nnzA = d.nnzA; a.nnzP = d.nnzP;
  a.max_iter = aa.max_iter; a.check = aa.check; a.slice = aa.slice; a.b0 = d.b0; a.list = d.list;
  a.bs = wh.bs; a.nb = wh.nb; a.mid = wh.mid; a.lpb = wh.lpb; a.n_extra = wh.n_extra; a.NS = wh.NS; a.NV = wh.NV; a.NSTEP = wh.NSTEP;
  a.cst_slots = wh.cst_slots; a.g_doubles = wh.g_doubles;
  a.rho = aa.rho; a.sigma = aa.sigma; a.alpha = aa.alpha; a.eps_abs = aa.eps_abs; a.eps_rel = aa.eps_rel;
  a.eps_prim_inf = aa.eps_prim_inf; a.eps_dual_inf = aa.eps_dual_inf;
  a.tab = wd.tab; a.G = wd.G; a.cst = wd.cst; a.wc = wd.wc; a.scr = wd.scr; a.ok = wd.ok; a.rl_need = wd.rl_need; a.w_ready = wd.w_ready; a.pflag = wd.pflag; a.it_count = wd.it_count;
  a.As = d.As; a.Ps = d.Ps; a.qs = d.qs; a.ls = d.ls; a.us = d.us; a.rhov = d.rho; a.kee_inv = d.kee_inv; a.cscale = d.cscale;
  a.D = d.D; a.E = d.E; a.S = d.W; a.w = d.w; a.active = d.active; a.setup_active = setup_mask;
  a.Ap = d.Ap; a.Ai = d.Ai; a.Rp = d.Rp; a.Rj = d.Rj; a.Rpos = d.Rpos; a.Fp = d.Fp; a.Fi = d.Fi; a.Fpos = d.Fpos;
  a.x = d.x; a.y = d.y; a.resid = d.resid; a.status = d.status; a.iters = d.iters; a.prog = d.prog;
  a.sx = d.sx; a.sz = d.sz; a.sy = d.sy; a.st = d.st; a.sg = d.sg;
  { const char *ab = getenv("SCO_WV_ABLATE"); a.ablate = ab ? atoi(ab) : 0; }
}

// need[b] = the problem starts a QP, or it is active on a QP the wavefront tier factored (W buffer still holds S)
__global__ void qp_wv_need_kernel(int b0, const int *list, int nb, const int *setup_mask, const int *active, int *w_ready, int *need) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= nb) return;
  const int b = list ? list[g + b0] : g + b0;
  if (b < 0) return;
  const int nd = (setup_mask ? setup_mask[b] != 0 : 1) || ((!active || active[b]) && !w_ready[b]);
  need[b] = nd;
  if (nd) w_ready[b] = 1;
}

int wv_launch_need(const AdmmArgs &aa, const int *setup_mask, const WvDev &wd, hipStream_t st) {
  const int nwg = aa.d.nb > 0 ? aa.d.nb : aa.d.batch;
  hipLaunchKernelGGL(qp_wv_need_kernel, dim3((nwg + 255) / 256), dim3(256), 0, st, aa.d.b0, aa.d.list, nwg, setup_mask, aa.d.active,
                     wd.w_ready, wd.rl_need);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}

int wv_launch_factor(const AdmmArgs &aa, const int *setup_mask, const WvHost &wh, const WvDev &wd, hipStream_t st) {
  WvArgs a; wv_fill_args(aa, wh, wd, setup_mask, a);
  const int nwg = aa.d.nb > 0 ? aa.d.nb : aa.d.batch;
  hipLaunchKernelGGL(qp_wv_factor_kernel, dim3(nwg), dim3(WV_T), 0, st, a);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}

template <int BS, int NS, int NV, int NSTEP, int LPB>
static int wv_launch_k(const WvArgs &a, int nwg, size_t lds, hipStream_t st) {
  static bool attr_done[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  dev_ &= 63;
  if (!attr_done[dev_]) {
    SCO_HIP(hipFuncSetAttribute((const void *)qp_admm_wv_kernel<BS, NS, NV, NSTEP, LPB>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
    attr_done[dev_] = true;
  }
  hipLaunchKernelGGL((qp_admm_wv_kernel<BS, NS, NV, NSTEP, LPB>), dim3(nwg), dim3(WV_T), lds, st, a);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}

int wv_launch(const AdmmArgs &aa, const WvHost &wh, const WvDev &wd, hipStream_t st) {
  WvArgs a; wv_fill_args(aa, wh, wd, nullptr, a);
  const int nwg = aa.d.nb > 0 ? aa.d.nb : aa.d.batch;
  // the LDS request also fixes how many problems share a CU: at most 4 (one wavefront per SIMD)
  const size_t lds = std::max(wh.lds_bytes, (size_t)36 * 1024);      // (36 KB: never a fifth workgroup on a CU)
  if (wh.NSTEP == 10 && wh.NS == 4 && wh.lpb == 3) return wv_launch_k<7, 4, 3, 10, 3>(a, nwg, lds, st);    // 7-DOF, 17 .. 20 timesteps
  if (wh.NSTEP == 10 && wh.NS == 4) return wv_launch_k<7, 4, 3, 10, 0>(a, nwg, lds, st);
  if (wh.NSTEP == 10) return wv_launch_k<8, 2, 2, 10, 0>(a, nwg, lds, st);
  if (wh.NSTEP == 4) return wv_launch_k<8, 1, 1, 4, 0>(a, nwg, lds, st);
  return wv_launch_k<8, 2, 2, 8, 0>(a, nwg, lds, st);
}
