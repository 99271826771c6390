// sco_admm_rl.hip -- "row-local" ADMM kernel for gfx950 (MI355X): the fastest tier.
//
// Same ADMM as the other tiers (OSQP's algorithm, the third-party call behind
// /root/reference/sco_py/sco_osqp/osqp_utils.py:216) with the two-level solve
//     x~_C = W (rhs_C - K_CE g),  g = K_EE^-1 rhs_E,  x~_E = g - K_EE^-1 K_EC x~_C
// rewritten so that the coupling block K_CE never materialises.  K's graph gives
// every row at most ONE eliminated variable e(i) (E is an independent set), hence
//     K_CE g      = A_C' u,          u_i = rw_i A_{i,e(i)} g_{e(i)}
//     K_EC x~_C   = sum_{i has e} rw_i A_{i,e} (A_{i,C} x~_C)
// so with  t'_i = t_i - u_i  the core right-hand side is one gather-dot over a core
// variable's column,  r_c = sigma x_c - q_c + sum_i A_ic t'_i,  and x~_e falls out of
// the core part of the row dot products its owner computes anyway.  All rows of an
// eliminated variable live in the SAME thread, so everything about e (x_e, q_e,
// 1/K_ee, g_e, u_i) is thread-local.
//
// One iteration = 3 phases / 3 barriers (the profile of the 6-phase kernel,
// profiles/r01_v4_pmc_summary.txt, showed 54 % of wave time in s_waitcnt/s_barrier):
//   (1)  core-variable owners:  r_c                     (CW-entry gather-dot)
//   (3)  all 512 threads:       register-tile W mat-vec; the 16 partial sums of a row are
//                                 folded with lane swaps (rl_reduce_rows), no LDS hop
//   (Y)  row owners:            core part of A x~, x~_e, z / y / x updates, t'
// Data placement as in sco_admm_reg.hip: W tile and packed byte offsets in
// registers, sparse values in LDS sliced-ELL images built in thread order
// (conflict-free, immediate offsets), padded slots gather an always-zero element.
#include "sco_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <type_traits>

// Diagnostic builds only (scripts/build_ablate.py): -DRL_ABLATE=<mask> removes one piece of the iteration so that its
// cost can be read off the per-iteration time (results are wrong; never compiled into the product).
//   1 column sums (1)   2 row dots of (Y)   4 W mat-vec FMAs   8 lane-swap reduction   16 the barriers
#ifndef RL_ABLATE
#define RL_ABLATE 0
#endif
// experiment switches of diagnostic builds (scripts/build_ablate.py --variant): 1 static priority for wavefronts 4-7,
// 2 register pin on the packed offsets (the r01 form), 4 unpadded right-hand side, 8 eight iterations per loop trip,
// 16 no termination test (the loop structure around it stays), 32 the checked iteration writes no test scratch,
// 64 wave-uniform row-slot masks in phase (Y) (r03: slower, 0.955 against 0.933 us), 128 values of the gather-dots read
// before the barrier in front of their phase, 512 the guard around the eliminated variable's update put back, 4096 the guard
// around the t' store put back (r03 removed both: profiles/r03_ab.txt section 7; tried there and dropped: gather-dot dispatch
// widest first -- slower --, no barrier behind the termination test and unguarded stores of the W totals -- no change)
#ifndef RL_VARIANT
#define RL_VARIANT 0
#endif
#define LT 512
#define LWV (LT / 64)
#define LGJ 16
#define LGI (LT / LGJ)
#define LCAP_M 1280
#define LCAP_NC 160
#define LCW 12          // value slots of a core variable's column = 2 x operand pairs (default; 16 for the widest instantiation)
#define LCW_MAX 16
#define LCW_MAX3 20       // three-row-slot instantiation: 10 operand pairs per column
#define LNS_MAX 3
#define LRW 8           // value slots of a row's core entries (4 aligned pairs); the wide instantiation (CW = 16) takes 10
#define LRW_MAX 10
#define RL_ROLES 29
// aligned closed assignment: core index of the thread's W tile rows (3) and of the mat-vec total it stores (-1: none)
#define ROLE_WROW(rr) (25 + (rr))
#define ROLE_XOUT 28
// tile of the aligned layout: 8 column groups (lane bits 0, 4, 5) x 8 row groups per wavefront (lane bits 1-3)
#define AL_TR 3
#define AL_TC 18
#define AL_ROWS (8 * AL_TR)       // W rows (core columns) a wavefront can own
// role-table rows of row slot q (slots 0, 1 keep their r01 places; slot 2 follows)
#define ROLE_ROW(q) ((q) < 2 ? 4 + (q) : 20)
#define ROLE_EPOS(q) ((q) < 2 ? 6 + (q) : 21)
#define ROLE_RBASE(q) ((q) < 2 ? 9 + (q) : 22)
#define ROLE_RWID(q) ((q) < 2 ? 14 + (q) : 23)
#define ROLE_POS(q) ((q) < 2 ? 16 + (q) : 24)

// Paired sliced-ELL: slices of 64 items (one wavefront); entries 2h and 2h+1 of lane l
// sit next to each other at base[s] + (64 h + l) * 2, so one ds_read_b128 per lane
// (contiguous 1 KiB per wave instruction, conflict-free) fetches two values.
static void build_sell2(int nitems, const std::vector<int> &ptr, const std::vector<int> &src, SellHost &out) {
  out.nitems = nitems;
  const int ns = (nitems + 63) / 64;
  out.base.assign(ns + 1, 0); out.width.assign(std::max(ns, 1), 0);
  for (int s = 0; s < ns; s++) {
    int w = 0;
    for (int it = s * 64; it < std::min(nitems, s * 64 + 64); it++) w = std::max(w, ptr[it + 1] - ptr[it]);
    w = (w + 1) & ~1;
    out.width[s] = w; out.base[s + 1] = out.base[s] + 64 * w;
  }
  out.total = out.base[ns];
  out.idx.clear(); out.src.assign(std::max(out.total, 1), -1);
  for (int it = 0; it < nitems; it++) {
    const int s = it / 64, l = it % 64;
    for (int k = 0; k < ptr[it + 1] - ptr[it]; k++)
      out.src[out.base[s] + (64 * (k / 2) + l) * 2 + (k & 1)] = src[ptr[it] + k];
  }
}

// --------------------------------------------------------------------------
// host: thread assignment and per-thread programs
// --------------------------------------------------------------------------
// Thread assignment in which every core column sits in the SAME wavefront as all of its rows ("closed"): the
// connected components of the row / core-column graph (rows of one eliminated variable tied together) are packed
// whole into wavefronts.  Then the column sums A_C' t' of phase (1) only read t' values written by their own
// wavefront, phase (1) follows phase (Y) without a workgroup barrier, and all eight wavefronts share the column
// work (in a trajectory QP a component is a timestep).  False if a component needs more than 64 threads or the
// components do not pack: the caller keeps the barrier then.
// cols_cap > 0 (aligned form): a wavefront also owns the W rows of its core columns, at most cols_cap of them; the core
// columns of wavefront w, in tile-row order, come back in wave_cols[w].
static bool rl_assign_closed(const QpPlan &pl, const std::vector<int> &row_elim, const std::vector<std::vector<int>> &erows,
                             std::vector<int> &slot_row, std::vector<int> &thr_core, std::vector<int> &thr_elim,
                             const int cols_cap = 0, std::vector<std::vector<int>> *wave_cols = nullptr) {
  const int m = pl.m, nc = pl.n_c, ne = pl.n_e;
  std::vector<int> uf(m + nc + ne);
  for (size_t i = 0; i < uf.size(); i++) uf[i] = (int)i;
  auto find = [&](int a) { while (uf[a] != a) { uf[a] = uf[uf[a]]; a = uf[a]; } return a; };
  auto unite = [&](int a, int b) { a = find(a); b = find(b); if (a != b) uf[a] = b; };
  for (int i = 0; i < m; i++)
    for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
      const int c = pl.core_of[pl.Rj[s]];
      if (c >= 0) unite(i, m + c);
    }
  for (int e = 0; e < ne; e++) for (int i : erows[e]) unite(m + nc + e, i);
  struct Comp { std::vector<int> cols, elims, free_rows; int need = 0; };
  std::vector<Comp> comps;
  std::vector<int> comp_of(uf.size(), -1);
  auto comp = [&](int node) -> Comp & {
    const int r = find(node);
    if (comp_of[r] < 0) { comp_of[r] = (int)comps.size(); comps.emplace_back(); }
    return comps[comp_of[r]];
  };
  for (int c = 0; c < nc; c++) comp(m + c).cols.push_back(c);
  for (int e = 0; e < ne; e++) comp(m + nc + e).elims.push_back(e);
  for (int i = 0; i < m; i++) if (row_elim[i] < 0) comp(i).free_rows.push_back(i);
  for (Comp &k : comps) {
    int spare = 0;
    for (int e : k.elims) spare += 2 - (int)erows[e].size();
    const int extra = std::max(0, (int)k.free_rows.size() - spare);
    k.need = std::max((int)k.elims.size() + (extra + 1) / 2, (int)k.cols.size());
    if (getenv("SCO_DEBUG_PLAN")) fprintf(stderr, "comp: cols %zu elims %zu free %zu need %d\n", k.cols.size(), k.elims.size(), k.free_rows.size(), k.need);
    if (k.need > 64) return false;
    if (k.need == 0) k.need = 1;
  }
  std::vector<int> order(comps.size());
  for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return comps[a].need > comps[b].need; });
  int load[LWV] = {0}, ncols[LWV] = {0};
  std::vector<std::vector<int>> bin(LWV);
  for (int k : order) {
    int best = -1;
    for (int w = 0; w < LWV; w++)
      if (load[w] + comps[k].need <= 64 && (cols_cap <= 0 || ncols[w] + (int)comps[k].cols.size() <= cols_cap) &&
          (best < 0 || load[w] < load[best])) best = w;
    if (best < 0) return false;
    load[best] += comps[k].need; ncols[best] += (int)comps[k].cols.size(); bin[best].push_back(k);
  }
  if (wave_cols) {
    wave_cols->assign(LWV, std::vector<int>());
    for (int w = 0; w < LWV; w++)
      for (int k : bin[w]) for (int c : comps[k].cols) (*wave_cols)[w].push_back(c);
  }
  slot_row.assign(2 * LT, -1); thr_core.assign(LT, -1); thr_elim.assign(LT, -1);
  for (int w = 0; w < LWV; w++) {
    int base = 64 * w;
    for (int k : bin[w]) {
      const Comp &K = comps[k];
      int t = base;
      for (int e : K.elims) {
        thr_elim[t] = e;
        for (size_t q = 0; q < erows[e].size(); q++) slot_row[q * LT + t] = erows[e][q];
        t++;
      }
      for (size_t j = 0; j < K.cols.size(); j++) thr_core[base + (int)j] = K.cols[j];
      // rows without an eliminated variable: the threads behind the eliminated ones first (slot 0, then 1), then
      // the slots the eliminated variables left empty
      std::vector<int> slots;
      for (int q = 0; q < 2; q++) for (int u = t; u < base + K.need; u++) slots.push_back(q * LT + u);
      for (int q = 0; q < 2; q++) for (int u = base; u < t; u++) if (slot_row[q * LT + u] < 0) slots.push_back(q * LT + u);
      if (slots.size() < K.free_rows.size()) return false;
      for (size_t j = 0; j < K.free_rows.size(); j++) slot_row[slots[j]] = K.free_rows[j];
      base += K.need;
    }
  }
  return true;
}

static bool rl_plan_build_ns(const QpPlan &pl, RlHost &rh, const int ns_min, const bool allow_aligned = false) {
  if (pl.n_c + 4 > LCAP_NC || pl.m >= LNS_MAX * LT || pl.n_e > LT || pl.n_c > LT || pl.nnzA >= 65536 || pl.n > 2 * LT) return false;
  const int m = pl.m;
  // rows of every eliminated variable; every row's eliminated variable
  std::vector<int> row_elim(m, -1), row_epos(m, -1);
  std::vector<std::vector<int>> erows(pl.n_e);
  for (int i = 0; i < m; i++)
    for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
      const int e = pl.elim_of[pl.Rj[s]];
      if (e >= 0) {
        if (row_elim[i] >= 0) return false;          // two eliminated variables in one row
        row_elim[i] = e; row_epos[i] = pl.Rpos[s]; erows[e].push_back(i);
      }
    }
  for (int e = 0; e < pl.n_e; e++) if (erows[e].size() > 2) return false;
  // ---- thread assignment: slot (q, t) -> row, thread -> core / eliminated variable
  std::vector<int> slot_row, thr_core, thr_elim;
  // The closed assignment is measured slower on the trajectory QPs (every wavefront then runs the widest row AND
  // column code, 957 against 871 ms per bench step), so it is opt-in: SCO_QP_RL_CLOSED=1.
  const char *closed = getenv("SCO_QP_RL_CLOSED");
  rh.NS = 2;
  // The ALIGNED closed assignment (r03; opt-in with SCO_QP_RL_ALIGNED=1, measured slower, see below): the wavefront
  // that holds a connected component's rows and columns also holds the W rows of those columns, so the mat-vec totals a
  // row needs are produced by its own wavefront: (3) -> (Y) -> (1) run without a workgroup barrier, ONE barrier per
  // iteration is left (before (3): the whole right-hand side), and the two wavefronts of a SIMD drift apart: the
  // latency-bound phases of one overlap the multiply-adds of the other.  For cores of 129 .. 144 variables (the tile
  // of that layout is 3 x 18 with 8 column groups); a pattern whose components do not pack keeps the open assignment.
  // Measured (profiles/r03_ab.txt): 1.057 against 0.932 us per iteration.  Every wavefront then runs every kind of work
  // at a third of its lanes (21 columns, ~45 row threads per wavefront): 176 instead of ~140 VALU and 39 instead of ~20
  // LDS instructions per wavefront and iteration, and an iteration costs what its instructions cost, not what its
  // barriers cost.
  const char *al = getenv("SCO_QP_RL_ALIGNED");
  const bool want_closed = (al && al[0] == '1') || (closed && closed[0] == '1');     // SCO_QP_RL_CLOSED (r02) = the aligned form now
  std::vector<std::vector<int>> wave_cols;
  rh.aligned = allow_aligned && ns_min == 2 && want_closed && pl.n_c > 128 && pl.n_c <= 8 * AL_TC &&
               rl_assign_closed(pl, row_elim, erows, slot_row, thr_core, thr_elim, AL_ROWS, &wave_cols);
  // The 8-column-group W layout with the OPEN assignment (r03): the W phase costs 3 swap folds + one quad step per
  // wavefront instead of 5 + 4 (a lane-swap instruction costs ~12 cycles, a multiply-add 4.5: scripts/microbench/inst_cost.hip),
  // six wavefronts carry W rows instead of eight.  Measured neutral without and slower with the termination test
  // (0.929 / 1.048 against 0.932 / 1.009 us per iteration, profiles/r03_ab.txt): 54 instead of 45 multiply-adds and 9
  // instead of 5 reads of the right-hand side eat what the reduction saves.  Opt-in: SCO_QP_RL_LAY8=1.
  const char *l8 = getenv("SCO_QP_RL_LAY8");
  rh.lay8 = !rh.aligned && ns_min == 2 && l8 && l8[0] == '1' && pl.n_c > 128 && pl.n_c <= 8 * AL_TC;
  rh.merged = rh.aligned;
  // every open plan puts a core column on a lane pair (2 n_c <= 312 lanes: n_c <= LCAP_NC - 4); SCO_QP_RL_SPLIT=0 keeps
  // all of a column's pairs on the owner and leaves the helper lane empty (the r02 distribution of the work)
  const char *sp = getenv("SCO_QP_RL_SPLIT");
  const bool split_pairs = !(sp && sp[0] == '0');
  rh.split = !rh.merged;
  if (!rh.merged) {
    // two row slots per thread; a pattern with more rows than that (velocity + joint limits at 7-DOF x 20: 1100) takes
    // the three-slot instantiation.  The rows of an eliminated variable (<= 2) always sit in slots 0 and 1 of its thread.
    for (int ns = ns_min; ns <= LNS_MAX; ns++) {
      rh.NS = ns;
      slot_row.assign((size_t)ns * LT, -1); thr_core.assign(LT, -1); thr_elim.assign(LT, -1);
      for (int e = 0; e < pl.n_e; e++) {               // eliminated variable e -> thread e
        thr_elim[e] = e;
        for (size_t q = 0; q < erows[e].size(); q++) slot_row[q * LT + e] = erows[e][q];
      }
      // far from the eliminated-variable threads.  Split form (r03): the entries of a core column go to TWO neighbouring
      // lanes (owner = even lane, helper = odd lane: half of the operand pairs each), their partial sums meet in one
      // quad_perm step -- phase (1) reads half as many operands per lane and runs on twice as many wavefronts
      for (int c = 0; c < pl.n_c; c++) thr_core[rh.split ? LT - 2 - 2 * c : LT - 1 - c] = c;
      int cur = 0;
      std::vector<int> order;                        // free slots: threads without an eliminated variable first
      for (int q = 0; q < ns; q++) for (int t = pl.n_e; t < LT; t++) order.push_back(q * LT + t);
      for (int q = 0; q < ns; q++) for (int t = 0; t < pl.n_e; t++) order.push_back(q * LT + t);
      bool ok = true;
      for (int i = 0; i < m && ok; i++) {
        if (row_elim[i] >= 0) continue;
        while (cur < (int)order.size() && slot_row[order[cur]] >= 0) cur++;
        if (cur >= (int)order.size()) ok = false;
        else slot_row[order[cur]] = i;
      }
      if (ok) break;
      if (ns == LNS_MAX) return false;
    }
  }
  if (rh.NS != 2) rh.lay8 = false;
  // ---- LDS positions of the row vectors (t', and w y / dy of the termination test).  A gather-dot reads its
  // operands two at a time (one ds_read_b128 per aligned PAIR of positions), so rows that one column reads together
  // should sit next to each other: rows with several core entries grouped by their set of columns (the rows of one
  // constraint block; a velocity limit and its negated copy), then the rows with a single core entry grouped by that column
  // (trust-region row, pin, joint limit of the same variable), every group on an even position.  Rows without a
  // core entry are never gathered and get no position.
  std::vector<int> pos(m, -1);
  int npos = 0;
  {
    std::vector<int> ncore(m, 0), onecol(m, -1);
    for (int i = 0; i < m; i++)
      for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
        const int c = pl.core_of[pl.Rj[s]];
        if (c >= 0) { ncore[i]++; onecol[i] = c; }
      }
    {
      // rows with several core entries: rows with the SAME set of core columns next to each other (the rows of a
      // constraint block already are; a velocity limit and its negated copy are d (T - 1) rows apart), groups in order
      // of first appearance, every group on an even position
      std::map<std::vector<int>, int> group_of;
      std::vector<std::vector<int>> groups;
      for (int i = 0; i < m; i++) {
        if (ncore[i] < 2) continue;
        std::vector<int> sig;
        for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) { const int c = pl.core_of[pl.Rj[s]]; if (c >= 0) sig.push_back(c); }
        std::sort(sig.begin(), sig.end());
        auto it = group_of.find(sig);
        if (it == group_of.end()) { group_of[sig] = (int)groups.size(); groups.emplace_back(1, i); }
        else groups[it->second].push_back(i);
      }
      for (const auto &g : groups) {
        npos = (npos + 1) & ~1;
        for (int i : g) pos[i] = npos++;
      }
    }
    std::vector<std::vector<int>> singles(pl.n_c);
    for (int i = 0; i < m; i++) if (ncore[i] == 1) singles[onecol[i]].push_back(i);
    for (int c = 0; c < pl.n_c; c++) {
      if (singles[c].empty()) continue;
      npos = (npos + 1) & ~1;
      for (int i : singles[c]) pos[i] = npos++;
    }
    npos = (npos + 1) & ~1;                            // the always-zero pair sits at npos, npos + 1
    rh.zpos = npos;
    npos += 2;
    for (int i = 0; i < m; i++) if (ncore[i] == 0) pos[i] = npos++;    // written, never gathered (no predicate in the loop)
    if (npos > LCAP_M - 1) return false;                 // position LCAP_M - 1 takes the stores of empty row slots
  }
  const int zcore = (pl.n_c + 1) & ~1;                 // always-zero pair of the core vectors
  if (zcore + 2 > LCAP_NC) return false;
  // pair lists: item -> [(pair index, source of the even element, source of the odd element)], -1 = structural zero
  struct Pair { int p, s0, s1; };
  auto make_pairs = [](std::vector<std::pair<int, int>> ent) {       // (position, source) -> pairs by position / 2
    std::sort(ent.begin(), ent.end());
    std::vector<Pair> out;
    for (auto &e : ent) {
      if (out.empty() || out.back().p != e.first / 2) out.push_back({e.first / 2, -1, -1});
      ((e.first & 1) ? out.back().s1 : out.back().s0) = e.second;
    }
    return out;
  };
  const int NSv = rh.NS;
  std::vector<std::vector<Pair>> colp(LT), rowp[LNS_MAX];
  for (int q = 0; q < NSv; q++) rowp[q].resize(LT);
  size_t maxc = 0, maxr = 0;
  for (int t = 0; t < LT; t++) {
    const int c = thr_core[t];
    if (c >= 0) {
      const int j = pl.core_var[c];
      std::vector<std::pair<int, int>> ent;
      for (int p = pl.Ap[j]; p < pl.Ap[j + 1]; p++) ent.push_back({pos[pl.Ai[p]], p});
      colp[t] = make_pairs(ent);
      if (rh.split && split_pairs) {                     // the second half of the pairs: the helper lane t + 1
        const size_t keep = (colp[t].size() + 1) / 2;
        colp[t + 1].assign(colp[t].begin() + keep, colp[t].end());
        colp[t].resize(keep);
      }
      maxc = std::max(maxc, colp[t].size());
    }
    for (int q = 0; q < NSv; q++) {
      const int i = slot_row[q * LT + t];
      if (i < 0) continue;
      std::vector<std::pair<int, int>> ent;
      for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
        const int c2 = pl.core_of[pl.Rj[s]];
        if (c2 >= 0) ent.push_back({c2, pl.Rpos[s]});
      }
      rowp[q][t] = make_pairs(ent);
      if (rowp[q][t].size() > LRW_MAX / 2) return false;
      maxr = std::max(maxr, rowp[q][t].size());
    }
  }
  if (maxc > (NSv > 2 ? LCW_MAX3 : LCW_MAX) / 2) return false;
  if (maxc > LCW / 2 || maxr > LRW / 2) rh.CW = LCW_MAX;      // the wide instantiation: 8 pairs per column, 5 per row
  // three row slots: 10 pairs per column thread, 5 per row -- unless the split plan leaves every column thread <= 6 pairs and
  // every row <= 4 (velocity + joint limits at 7-DOF x 20: 5 + 5 pairs): then the narrow offsets do, with far fewer registers
  if (NSv > 2) rh.CW = (maxc <= LCW / 2 && maxr <= LRW / 2) ? LCW : LCW_MAX3;
  // ---- sliced-ELL images in thread order: two value slots per pair
  // Open assignment: item = thread.  Closed assignments spread every kind of work over all wavefronts (a wavefront holds
  // ~21 of the 140 columns), so an image with one item per THREAD would be 8 full slices per operator and overflow LDS:
  // there the items are the threads that have entries, in thread order (`ipos[t]`: image position of thread t's first
  // value, -1 = none: such a thread reads the zero block behind the images), and a wavefront's trip count is the
  // widest of its own threads (`iwid`).  Its active lanes still read consecutive 16-byte pieces: conflict-free.
  const bool compact = rh.merged;
  auto image = [&](const std::vector<std::vector<Pair>> &items, SellHost &out, std::vector<int> &ipos, std::vector<int> &iwid) {
    std::vector<int> ptr(1, 0), src, item_of(LT, -1);
    int nit = 0;
    for (int t = 0; t < LT; t++) {
      if (compact && items[t].empty()) continue;
      for (const Pair &pr : items[t]) { src.push_back(pr.s0); src.push_back(pr.s1); }
      ptr.push_back((int)src.size());
      item_of[t] = nit++;
    }
    if (nit == 0) { ptr.push_back(0); nit = 1; }
    build_sell2(nit, ptr, src, out);
    ipos.assign(LT, -1); iwid.assign(LT, 0);
    for (int t = 0; t < LT; t++) {
      if (item_of[t] >= 0) ipos[t] = out.base[item_of[t] / 64] + 2 * (item_of[t] % 64);
      if (!compact) iwid[t] = out.width[t / 64];
    }
    if (compact)
      for (int w = 0; w < LWV; w++) {
        int mx = 0;
        for (int t = 64 * w; t < 64 * w + 64; t++) mx = std::max(mx, 2 * (int)items[t].size());
        for (int t = 64 * w; t < 64 * w + 64; t++) iwid[t] = mx;
      }
  };
  std::vector<int> cpos, cwid, rpos[LNS_MAX], rwid[LNS_MAX];
  image(colp, rh.Ac, cpos, cwid);
  size_t rtot = 0;
  for (int q = 0; q < NSv; q++) { image(rowp[q], rh.Ar[q], rpos[q], rwid[q]); rtot += rh.Ar[q].total; }
  const int zblock = rh.Ac.total + (int)rtot;          // 64 x 16 zeros behind the images (kernel prologue)
  rh.lds_bytes = 8 * ((size_t)rh.Ac.total + rtot + 64 * 16 + (2 * NSv + 5) * LT) + 12 * 4 * LCAP_NC;   // + check constants
  // ---- per-thread tables: packed pair offsets (bytes) and roles
  const int CPv = rh.CW / 2;
  const int RPv = (rh.CW > LCW ? LRW_MAX : LRW) / 2;          // pairs per row slot
  const int slots = CPv + NSv * RPv + 1;                         // + 1: the last register of a slot is packed from two table rows
  rh.off.assign((size_t)slots * LT, 0);
  rh.role.assign((size_t)RL_ROLES * LT, -1);
  for (int t = 0; t < LT; t++) {
    for (int k = 0; k < CPv; k++) rh.off[(size_t)k * LT + t] = (unsigned short)(8 * rh.zpos);               // zero pair of t'
    for (int k = 0; k < NSv * RPv + 1; k++) rh.off[(size_t)(CPv + k) * LT + t] = (unsigned short)(8 * zcore);   // zero pair of x_C
    // role table: 0 core idx, 1 core var, 2 elim idx, 3 elim var, 4/5 row of slot 0/1,
    //             6/7 CSC position of the row's eliminated coefficient, 8 col base, 9/10 row bases,
    //             11/12 position of P_jj for the core / eliminated variable, 13-15 trip counts,
    //             16/17 LDS position of the row of slot 0/1
    rh.role[(size_t)8 * LT + t] = cpos[t] >= 0 ? cpos[t] : zblock + 2 * (t % 64);
    for (int q = 0, acc = rh.Ac.total; q < NSv; acc += rh.Ar[q].total, q++) {
      rh.role[(size_t)ROLE_RBASE(q) * LT + t] = rpos[q][t] >= 0 ? acc + rpos[q][t] : zblock + 2 * (t % 64);
      rh.role[(size_t)ROLE_RWID(q) * LT + t] = rwid[q][t];
    }
    rh.role[(size_t)13 * LT + t] = cwid[t];                  // wave-uniform trip counts (value slots = 2 x pairs)
    const int c = thr_core[t];
    if (c >= 0) {
      const int j = pl.core_var[c];
      rh.role[t] = c; rh.role[(size_t)LT + t] = j; rh.role[(size_t)11 * LT + t] = pl.Pdiag[j];
    }
    for (size_t k = 0; k < colp[t].size(); k++) rh.off[k * LT + t] = (unsigned short)(16 * colp[t][k].p);   // owner or helper
    const int e = thr_elim[t];
    if (e >= 0) {
      rh.role[(size_t)2 * LT + t] = e; rh.role[(size_t)3 * LT + t] = pl.elim_var[e];
      rh.role[(size_t)12 * LT + t] = pl.Pdiag[pl.elim_var[e]];
    }
    if (rh.aligned || rh.lay8) {
      const int wv = t / 64, lane = t % 64, g = (lane >> 1) & 7, h = (lane >> 4) & 3;
      for (int rr = 0; rr < AL_TR; rr++) {
        const int lr = g * AL_TR + rr;
        // open assignment: wavefront w carries the W rows 24 w .. 24 w + 23 (six wavefronts at 140 core variables)
        rh.role[(size_t)ROLE_WROW(rr) * LT + t] = rh.aligned ? (lr < (int)wave_cols[wv].size() ? wave_cols[wv][lr] : -1)
                                                             : (wv * AL_ROWS + lr < pl.n_c ? wv * AL_ROWS + lr : -1);
      }
      // after the folds (rl_reduce_al) the total of tile row 0 / 1 / 2 sits in the lanes with (bit 5, bit 4) = 00 / 10 / 01
      const int rr_out = h == 0 ? 0 : h == 2 ? 1 : h == 1 ? 2 : -1;
      if ((lane & 1) == 0 && rr_out >= 0) rh.role[(size_t)ROLE_XOUT * LT + t] = rh.role[(size_t)ROLE_WROW(rr_out) * LT + t];
    }
    for (int q = 0; q < NSv; q++) {
      const int i = slot_row[q * LT + t];
      rh.role[(size_t)ROLE_ROW(q) * LT + t] = i;
      if (i < 0) continue;
      rh.role[(size_t)ROLE_POS(q) * LT + t] = pos[i];
      if (row_elim[i] >= 0) {
        if (row_elim[i] != e) return false;
        if (q >= 2) return false;                    // an eliminated variable's rows live in slots 0 and 1
        rh.role[(size_t)ROLE_EPOS(q) * LT + t] = row_epos[i];
      }
      for (size_t k = 0; k < rowp[q][t].size(); k++)
        rh.off[(size_t)(CPv + q * RPv + k) * LT + t] = (unsigned short)(16 * rowp[q][t][k].p);
    }
  }
  // P restricted to the core (for the dual residual): core c -> (position in triu values, core index)
  rh.pc_ptr.assign(pl.n_c + 1, 0);
  for (int c = 0; c < pl.n_c; c++) {
    const int j = pl.core_var[c];
    for (int p = pl.Fp[j]; p < pl.Fp[j + 1]; p++) {
      const int c2 = pl.core_of[pl.Fi[p]];
      if (c2 < 0) return false;                    // an eliminated variable has no off-diagonal P entry
      rh.pc_pos.push_back(pl.Fpos[p]); rh.pc_core.push_back(c2);
    }
    rh.pc_ptr[c + 1] = (int)rh.pc_pos.size();
    rh.pcw = std::max(rh.pcw, rh.pc_ptr[c + 1] - rh.pc_ptr[c]);
  }
  rh.TR = std::max(1, (pl.n_c + LGI - 1) / LGI);
  rh.TC = 2 * rh.TR;
  if (rh.TR == 5 && pl.n_c <= LGJ * 9) rh.TC = 9;
  if (rh.CW > LCW && rh.TR != 5) return false;        // the wide variants exist for the 5-row tiles only
  if (rh.NS != 2) rh.lay8 = false;
  if (rh.aligned || rh.lay8) { rh.TR = AL_TR; rh.TC = AL_TC; }
  return rh.lds_bytes + 40 * 1024 <= 160 * 1024;   // + 37.5 KB of static LDS
}

// Two row slots per thread where the pattern allows it; otherwise (more than 1024 rows, or more than 8 operand pairs in
// a column) the three-slot instantiation with ten pairs per column.
bool rl_plan_build(const QpPlan &pl, RlHost &rh) {
  {
    RlHost al;
    if (rl_plan_build_ns(pl, al, 2, true) && al.aligned) { rh = al; return true; }
  }
  {
    RlHost two;
    if (rl_plan_build_ns(pl, two, 2)) { rh = two; return true; }
  }
  RlHost three;
  if (!rl_plan_build_ns(pl, three, 3)) return false;
  rh = three;
  return true;
}

// --------------------------------------------------------------------------
// device
// --------------------------------------------------------------------------
struct RlArgs {
  int n, m, n_e, n_c, nnzA, nnzP, max_iter, check;
  int b0;            // launch window: workgroup g solves problem b0 + g, or list[b0 + g] (< 0: none)
  const int *list;
  double sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  const unsigned short *off; const int *role;
  const int *srcAc, *srcAr[LNS_MAX], *pc_ptr, *pc_pos, *pc_core;
  int pcw;           // most P entries in a core column; <= 4: the termination test reads them from LDS
  int zpos;          // LDS position of the always-zero pair of the row vectors
  int totAc, totAr[LNS_MAX];
  const double *As, *W, *qs, *kee_inv, *ls, *us, *rho, *cscale, *Ps, *D, *E;
  const int *w, *active;
  const int *skip;   // or null: skip[b] != 0 = the wavefront tier solves this problem
  double *x, *y, *resid;
  int *status, *iters;
  int warm;          // start from the previous (unscaled) solution held in x / y instead of zero
  // time slicing (slice > 0): at most `slice` iterations per launch; an unfinished solve leaves status 0,
  // its iteration count in prog[b] and its scaled state in the s* arrays, and the next launch resumes it
  // bit-exactly (nothing is recomputed)
  int slice;
  // adaptive rho (ADAPT instantiations only): every ad_interval iterations, after the termination test, OSQP's
  // estimate from the scaled residual norms the test has just formed; when it leaves [rho / tol, rho tol] the solve
  // parks, leaves the new rho in rho_b[b] and raises smask[b] (setup must refactor) and rflag[b] (the resume
  // rebuilds the cached right-hand sides t', g_e from x, z, y)
  int ad_interval;
  double ad_tol;
  double *rho_b;
  int *rflag, *smask, *nupd;
  int *prog;
  double *sx, *sz, *sy, *st, *sg;
  double *stamp;     // diagnostic build only (SCO_STAMP), else unused
};

__device__ __forceinline__ double lwmax(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
  return v;
}
__device__ __forceinline__ double lwsum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
// maximum over the wavefront, valid in lane 63: DPP row shifts and row broadcasts (VALU only; the shuffle version
// goes through the LDS permute path six dependent times per value and made the termination test's reductions cost
// 3.7 k cycles each, profiles/r01_check_stamps.txt).  A lane without a source keeps its own value, so a NaN
// survives exactly when every lane holds one, as with the shuffles.
__device__ __forceinline__ double lwmax63(double v) {
  int lo, hi, lo2, hi2;
#define RL_DPP_MAX(ctrl, rmask)                                                              \
  lo = __double2loint(v); hi = __double2hiint(v);                                            \
  lo2 = __builtin_amdgcn_update_dpp(lo, lo, ctrl, rmask, 0xf, false);                        \
  hi2 = __builtin_amdgcn_update_dpp(hi, hi, ctrl, rmask, 0xf, false);                        \
  v = fmax(v, __hiloint2double(hi2, lo2));
  RL_DPP_MAX(0x111, 0xf) RL_DPP_MAX(0x112, 0xf) RL_DPP_MAX(0x114, 0xf) RL_DPP_MAX(0x118, 0xf)
  RL_DPP_MAX(0x142, 0xa) RL_DPP_MAX(0x143, 0xc)
#undef RL_DPP_MAX
  return v;
}
typedef unsigned int rl_uint2 __attribute__((ext_vector_type(2)));
// Block maximum of six values with the lane-swap folds of the W reduction (fmax instead of add; fmax is idempotent, so
// an odd value folds with itself): 17 swap / max instructions, then two row_shr scans instead of six.  After the folds
// lane 15 of DPP row r holds  t0: v0, v2, v1, v3 (r = 0..3)  and  t1: v4 (r = 0, 1), v5 (r = 2, 3).
__device__ __forceinline__ double rl_fmax32(double a, double b) {
  const rl_uint2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const rl_uint2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return fmax(__hiloint2double((int)hi.x, (int)lo.x), __hiloint2double((int)hi.y, (int)lo.y));
}
__device__ __forceinline__ double rl_fmax16(double a, double b) {
  const rl_uint2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const rl_uint2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return fmax(__hiloint2double((int)hi.x, (int)lo.x), __hiloint2double((int)hi.y, (int)lo.y));
}
__device__ __forceinline__ double rl_rowmax15(double v) {        // maximum of a DPP row, valid in its lane 15
  int lo, hi, lo2, hi2;
#define RL_ROW_MAX(ctrl)                                                                     \
  lo = __double2loint(v); hi = __double2hiint(v);                                            \
  lo2 = __builtin_amdgcn_update_dpp(lo, lo, ctrl, 0xf, 0xf, false);                          \
  hi2 = __builtin_amdgcn_update_dpp(hi, hi, ctrl, 0xf, 0xf, false);                          \
  v = fmax(v, __hiloint2double(hi2, lo2));
  RL_ROW_MAX(0x111) RL_ROW_MAX(0x112) RL_ROW_MAX(0x114) RL_ROW_MAX(0x118)
#undef RL_ROW_MAX
  return v;
}
__device__ __forceinline__ double rl_sum32(double a, double b) {
  const rl_uint2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const rl_uint2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double rl_sum16(double a, double b) {
  const rl_uint2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const rl_uint2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
// one step of a scan inside the DPP rows: v op (v shifted right by the row_shr control), lanes without a source keep v
#define RL_ROW_STEP_MAX(v, ctrl)                                                             \
  { const int lo_ = __double2loint(v), hi_ = __double2hiint(v);                              \
    v = fmax(v, __hiloint2double(__builtin_amdgcn_update_dpp(hi_, hi_, ctrl, 0xf, 0xf, false),   \
                                 __builtin_amdgcn_update_dpp(lo_, lo_, ctrl, 0xf, 0xf, false))); }
#define RL_ROW_STEP_ADD(v, ctrl)                                                             \
  { const int lo_ = __double2loint(v), hi_ = __double2hiint(v);                              \
    v += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi_, ctrl, 0xf, 0xf, true),         \
                          __builtin_amdgcn_update_dpp(0, lo_, ctrl, 0xf, 0xf, true)); }
__device__ __forceinline__ double rl_bcast_lane(double v, int lane) {      // lane: compile-time constant
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// Block reduction of the termination test: maxima of v[0..5] and sums of v[6], v[7] over the workgroup, result in
// every thread.  Inside a wavefront the lane-swap folds of the W reduction (fmax is idempotent, so an odd value folds
// with itself; a sum folds with zero): after them lane 15 of DPP row r holds  t0: v0, v2, v1, v3 (r = 0..3),
// t1: v4 (r = 0, 1), v5 (r = 2, 3),  t2: v6 (r = 0), v7 (r = 2).  Across the wavefronts: the eight partial results of
// value k sit in red[8 k .. 8 k + 7], ONE 8-byte read per lane fetches all 64, three row_shr steps combine the eight
// of a value in lane 8 k + 7 and v_readlane hands it to every lane.  (r02: every thread used to read all 48 partial
// results back: 24 KB of LDS reads per wavefront, 1.5 k cycles of a 6.4 k-cycle test.)  Fixed association order.
__device__ __forceinline__ void lblock_max6_sum2(double (&v)[8], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const double u0 = rl_fmax32(v[0], v[1]), u1 = rl_fmax32(v[2], v[3]), u2 = rl_fmax32(v[4], v[5]), u3 = rl_sum32(v[6], v[7]);
  const double t0 = rl_rowmax15(rl_fmax16(u0, u1)), t1 = rl_rowmax15(rl_fmax16(u2, u2));
  double t2 = rl_sum16(u3, 0.0);
  RL_ROW_STEP_ADD(t2, 0x111) RL_ROW_STEP_ADD(t2, 0x112) RL_ROW_STEP_ADD(t2, 0x114) RL_ROW_STEP_ADD(t2, 0x118)
  __syncthreads();
  if ((lane & 15) == 15) {
    const int r = lane >> 4;
    red[8 * (r == 0 ? 0 : r == 1 ? 2 : r == 2 ? 1 : 3) + wv] = t0;
    if ((r & 1) == 0) { red[8 * (4 + (r >> 1)) + wv] = t1; red[8 * (6 + (r >> 1)) + wv] = t2; }
  }
  __syncthreads();
  double mx = red[lane], sm = mx;
  RL_ROW_STEP_MAX(mx, 0x111) RL_ROW_STEP_MAX(mx, 0x112) RL_ROW_STEP_MAX(mx, 0x114)
  RL_ROW_STEP_ADD(sm, 0x111) RL_ROW_STEP_ADD(sm, 0x112) RL_ROW_STEP_ADD(sm, 0x114)
  v[0] = rl_bcast_lane(mx, 7); v[1] = rl_bcast_lane(mx, 15); v[2] = rl_bcast_lane(mx, 23); v[3] = rl_bcast_lane(mx, 31);
  v[4] = rl_bcast_lane(mx, 39); v[5] = rl_bcast_lane(mx, 47); v[6] = rl_bcast_lane(sm, 55); v[7] = rl_bcast_lane(sm, 63);
}

template <int NR, bool IS_MAX>
__device__ __forceinline__ void lblock_reduce(double (&v)[NR], double *red) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NR; k++) v[k] = IS_MAX ? lwmax63(v[k]) : lwsum(v[k]);
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
    for (int w = 1; w < LWV; w++) r = IS_MAX ? fmax(r, red[w * NR + k]) : r + red[w * NR + k];
    v[k] = r;
  }
}

__device__ __forceinline__ double lgat(const double *base, unsigned int byte_off) {
  return *(const double *)((const char *)base + byte_off);
}

typedef double dbl2 __attribute__((ext_vector_type(2)));

// Gather-dot over N value slots = N / 2 operand PAIRS: pair h multiplies the two values V[128 h], V[128 h + 1] of
// the paired image (one 16-byte LDS read, contiguous per wavefront) with the two operands at byte offset o[h] of
// `vec` (one 16-byte gather; o[h] is a multiple of 16).  A padded pair has zero values and gathers the always-zero
// pair.  Offsets stay packed two per VGPR.  (With per-entry offsets an empty asm had to pin them packed or the compiler
// hoisted the unpack out of the ADMM loop and spilled; with half as many pair offsets the pin only costs a move per
// register: 1.095 -> 1.074 us per iteration without it.)  Even and odd elements run in two accumulation chains.
template <int N>
__device__ __forceinline__ double rl_dot(const double *V, unsigned int *o, const double *vec) {
  constexpr int NP = N / 2;
  if constexpr (NP > 6) {
    // the wide instantiations (8 and 10 pairs per column): two halves, so that at most half of the operands are in
    // registers at once (the three-row-slot kernel spilled 141 VGPRs with all twenty loaded ahead of the arithmetic)
    constexpr int H = (NP / 2 + 1) & ~1;          // even: the packed offsets of the second half start on a register
    return rl_dot<2 * H>(V, o, vec) + rl_dot<2 * (NP - H)>(V + 128 * H, o + H / 2, vec);
  }
  dbl2 val[NP], g[NP];
#pragma unroll
  for (int h = 0; h < NP; h++) {
    if ((h & 1) == 0 && (RL_VARIANT & 2)) asm volatile("" : "+v"(o[h / 2]));
    val[h] = *(const dbl2 *)(V + 128 * h);
    const unsigned int off = (h & 1) ? (o[h / 2] >> 16) : (o[h / 2] & 0xffffu);
    g[h] = *(const dbl2 *)((const char *)vec + off);
  }
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int h = 0; h < NP; h++) { a0 += val[h].x * g[h].x; a1 += val[h].y * g[h].y; }
  return a0 + a1;
}

// The gather-dots of the loop, dispatched on the wave-uniform trip count with NESTED tiers (r03) --
// `if (w > 0) { pairs of tier 0; if (w > t1) { ...; if (w > t2) { .. }}}`, once for the loads and once for the multiply-adds
// (pair order as in rl_dot: the results are bit-identical).  The widest path -- the wavefronts on the critical path -- only
// falls through branches that are not taken; the chain of `if (w <= t) return rl_dot<t>` it replaces left three TAKEN
// branches behind every dot (0.973 -> 0.956 us per iteration, profiles/r03_ab.txt section 8;
// RL_VARIANT & 131072 puts the chain back)
#define RL_LD(h) { val[h] = *(const dbl2 *)(V + 128 * (h)); g[h] = *(const dbl2 *)((const char *)vec + (((h) & 1) ? (o[(h) / 2] >> 16) : (o[(h) / 2] & 0xffffu))); }
#define RL_FM(h) { a0 += val[h].x * g[h].x; a1 += val[h].y * g[h].y; }
template <int CW>
__device__ __forceinline__ double rl_dot_col_nested(int w, const double *V, unsigned int *o, const double *vec) {
  dbl2 val[CW / 2], g[CW / 2];
  double a0 = 0.0, a1 = 0.0;
  if (w > 0) { RL_LD(0) RL_LD(1)
    if (w > 4) { RL_LD(2)
      if (w > 6) { RL_LD(3)
        if (w > 8) { RL_LD(4) RL_LD(5)
          if constexpr (CW > 12) { if (w > 12) { RL_LD(6) RL_LD(7)
            if constexpr (CW > 16) { if (w > 16) { RL_LD(8) RL_LD(9) } } } } } } } }
  if (w > 0) { RL_FM(0) RL_FM(1)
    if (w > 4) { RL_FM(2)
      if (w > 6) { RL_FM(3)
        if (w > 8) { RL_FM(4) RL_FM(5)
          if constexpr (CW > 12) { if (w > 12) { RL_FM(6) RL_FM(7)
            if constexpr (CW > 16) { if (w > 16) { RL_FM(8) RL_FM(9) } } } } } } } }
  return a0 + a1;
}
template <int RW>
__device__ __forceinline__ double rl_dot_row_nested(int w, const double *V, unsigned int *o, const double *vec) {
  dbl2 val[RW / 2], g[RW / 2];
  double a0 = 0.0, a1 = 0.0;
  if (w > 0) { RL_LD(0)
    if (w > 2) { RL_LD(1)
      if (w > 4) { RL_LD(2) RL_LD(3)
        if constexpr (RW > LRW) { if (w > LRW) { RL_LD(4) } } } } }
  if (w > 0) { RL_FM(0)
    if (w > 2) { RL_FM(1)
      if (w > 4) { RL_FM(2) RL_FM(3)
        if constexpr (RW > LRW) { if (w > LRW) { RL_FM(4) } } } } }
  return a0 + a1;
}
#undef RL_LD
#undef RL_FM

// dispatch on the wave-uniform trip count so padded slots cost nothing
// NEST: nested tiers (the two-row-slot kernels); the three-row-slot kernels keep the chain -- with every pair of a dot loaded
// ahead of the multiply-adds they spill inside the loop (velocity + joint limits 1.39 -> 1.53, reach + velocity + joint limits
// 1.91 -> 2.72 us per problem-iteration, profiles/r03_family_speed.txt)
template <int CW, bool NEST = true>
__device__ __forceinline__ double rl_dot_col(int w, const double *V, unsigned int *o, const double *vec) {
  if constexpr (NEST && (RL_VARIANT & 131072) == 0) return rl_dot_col_nested<CW>(w, V, o, vec);
  if (w <= 0) return 0.0;
  if (w <= 4) return rl_dot<4>(V, o, vec);
  if (w <= 6) return rl_dot<6>(V, o, vec);          // half of a 6-pair column (split form)
  if (w <= 8) return rl_dot<8>(V, o, vec);
  if constexpr (CW > 12) { if (w <= 12) return rl_dot<12>(V, o, vec); }
  if constexpr (CW > 16) { if (w <= 16) return rl_dot<16>(V, o, vec); }
  return rl_dot<CW>(V, o, vec);
}
template <int RW, bool NEST = true>
__device__ __forceinline__ double rl_dot_row_w(int w, const double *V, unsigned int *o, const double *vec) {
  if constexpr (NEST && (RL_VARIANT & 131072) == 0) return rl_dot_row_nested<RW>(w, V, o, vec);
  if (w <= 0) return 0.0;
  if (w <= 2) return rl_dot<2>(V, o, vec);
  if (w <= 4) return rl_dot<4>(V, o, vec);
  if constexpr (RW > LRW) { if (w <= LRW) return rl_dot<LRW>(V, o, vec); }
  return rl_dot<RW>(V, o, vec);
}

// The same gather-dots in two halves (RL_VARIANT & 128): the VALUES of a thread's entries never change during a solve,
// so they can be read before the barrier that precedes the phase; only the gathered operands have to wait for it.
template <int NP, bool COL>
__device__ __forceinline__ void rl_vals(int w, const double *V, dbl2 (&val)[NP]) {
  // pairs the dispatch below will multiply for this (wave-uniform) trip count
  const int need = w <= 0 ? 0 : COL ? (w <= 4 ? 2 : w <= 8 ? 4 : w <= 12 ? 6 : w <= 16 ? 8 : NP) : (w <= 2 ? 1 : w <= 4 ? 2 : w <= 8 ? 4 : NP);
#pragma unroll
  for (int h = 0; h < NP; h++)
    if (h < need) val[h] = *(const dbl2 *)(V + 128 * h);
}
template <int N, int NPMAX>
__device__ __forceinline__ double rl_dot_pre(const dbl2 (&val)[NPMAX], unsigned int *o, const double *vec) {
  constexpr int NP = N / 2;
  dbl2 g[NP];
#pragma unroll
  for (int h = 0; h < NP; h++) {
    const unsigned int off = (h & 1) ? (o[h / 2] >> 16) : (o[h / 2] & 0xffffu);
    g[h] = *(const dbl2 *)((const char *)vec + off);
  }
  double a0 = 0.0, a1 = 0.0;
#pragma unroll
  for (int h = 0; h < NP; h++) { a0 += val[h].x * g[h].x; a1 += val[h].y * g[h].y; }
  return a0 + a1;
}
template <int CW>
__device__ __forceinline__ double rl_dot_col_pre(int w, const dbl2 (&val)[CW / 2], unsigned int *o, const double *vec) {
  if (w <= 0) return 0.0;
  if (w <= 4) return rl_dot_pre<4>(val, o, vec);
  if (w <= 8) return rl_dot_pre<8>(val, o, vec);
  if constexpr (CW > 12) { if (w <= 12) return rl_dot_pre<12>(val, o, vec); }
  if constexpr (CW > 16) { if (w <= 16) return rl_dot_pre<16>(val, o, vec); }
  return rl_dot_pre<CW>(val, o, vec);
}
template <int RW>
__device__ __forceinline__ double rl_dot_row_pre(int w, const dbl2 (&val)[RW / 2], unsigned int *o, const double *vec) {
  if (w <= 0) return 0.0;
  if (w <= 2) return rl_dot_pre<2>(val, o, vec);
  if (w <= 4) return rl_dot_pre<4>(val, o, vec);
  if constexpr (RW > LRW) { if (w <= LRW) return rl_dot_pre<LRW>(val, o, vec); }
  return rl_dot_pre<RW>(val, o, vec);
}

// ---- phase (3) reduction --------------------------------------------------------------
// The 16 partial sums of a row of the W mat-vec sit in the lanes with the same (lane >> 2) & 3:
// 4 DPP rows (lane >> 4) x 4 lanes of a quad (lane & 3).  Two rows' partial sums are folded per
// lane-swap instruction pair (v_permlane32_swap across the wavefront halves, v_permlane16_swap across
// the DPP rows: after swapping register a of the upper lanes with register b of the lower lanes one
// add leaves the pair sum of a in the lower and of b in the upper lanes), the last two levels run
// inside the quad with quad_perm moves: 27 VALU instructions for 5 rows instead of the 60 of five
// row_shr scans.  Fixed association order.
__device__ __forceinline__ double rl_fold32(double a, double b) {
  const rl_uint2 lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const rl_uint2 hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
__device__ __forceinline__ double rl_fold16(double a, double b) {
  const rl_uint2 lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(a), (unsigned)__double2loint(b), false, false);
  const rl_uint2 hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(a), (unsigned)__double2hiint(b), false, false);
  return __hiloint2double((int)hi.x, (int)lo.x) + __hiloint2double((int)hi.y, (int)lo.y);
}
// both lanes of a pair (lane ^ 1) end up with the pair's sum
__device__ __forceinline__ double rl_pair_sum(double v) {
  const int lo = __double2loint(v), hi = __double2hiint(v);
  return v + __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true),     // quad_perm [1,0,3,2]
                              __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true));
}
// all four lanes of a quad end up with the quad's sum
__device__ __forceinline__ double rl_quad_sum(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true),     // quad_perm [1,0,3,2]
                        __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true));
  lo = __double2loint(v); hi = __double2hiint(v);
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0x4e, 0xf, 0xf, true),     // quad_perm [2,3,0,1]
                        __builtin_amdgcn_update_dpp(0, lo, 0x4e, 0xf, 0xf, true));
  return v;
}
// acc[rr] = this lane's partial sum of tile row rr.  Returns t[n]: in the lanes of DPP row h the total of
// tile row 4 n + 2 (h & 1) + (h >> 1) (rl_tile_row below); garbage where that index is >= TR.
#define RL_NT(TR) (((TR) + 3) / 4)
template <int TR>
__device__ __forceinline__ void rl_reduce_rows(const double (&acc)[TR], double (&t)[RL_NT(TR)]) {
  constexpr int NU = (TR + 1) / 2;
  double u[NU];
#pragma unroll
  for (int m = 0; m < NU; m++) u[m] = rl_fold32(acc[2 * m], 2 * m + 1 < TR ? acc[2 * m + 1] : 0.0);
#pragma unroll
  for (int n = 0; n < RL_NT(TR); n++) t[n] = rl_quad_sum(rl_fold16(u[2 * n], 2 * n + 1 < NU ? u[2 * n + 1] : 0.0));
}
__device__ __forceinline__ int rl_tile_row(int n, int h) { return 4 * n + 2 * (h & 1) + (h >> 1); }
// Aligned layout: 8 column groups = lane bits 5, 4, 0.  Three tile rows: the swap folds over bits 5 and 4 leave the
// total (over those bits) of row 0 / 1 / 2 in the lanes with (bit 5, bit 4) = 00 / 10 / 01, one quad_perm step adds the
// lane pair: 12 VALU instructions for 3 rows.  Fixed association order.
__device__ __forceinline__ double rl_reduce_al(const double (&acc)[AL_TR]) {
  double v = rl_fold16(rl_fold32(acc[0], acc[1]), rl_fold32(acc[2], 0.0));
  const int lo = __double2loint(v), hi = __double2hiint(v);
  v += __hiloint2double(__builtin_amdgcn_update_dpp(0, hi, 0xb1, 0xf, 0xf, true),     // quad_perm [1,0,3,2]
                        __builtin_amdgcn_update_dpp(0, lo, 0xb1, 0xf, 0xf, true));
  return v;
}
// wavefront-local hand-over through LDS (aligned / closed assignment): the LDS operations of one wavefront complete
// in order, so no s_barrier is needed; this only keeps the compiler from moving accesses across the hand-over
#define RL_WAVE_SYNC() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); \
                         __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }

// LAY: 0 = 16 column groups of the W tile (5 x 9 at 140 core variables), 1 = 8 column groups (3 x 18, cheaper reduction, six
// of the eight wavefronts carry W), 2 = 8 column groups with the aligned closed assignment (one barrier per iteration)
template <int TR, int TC, int CW, bool ADAPT, int NS = 2, int LAY = 0>
__global__ __launch_bounds__(LT) void qp_admm_rl_kernel(RlArgs a) {
  constexpr bool L8 = LAY != 0, AL = LAY == 2;
  static_assert(!L8 || (TR == AL_TR && TC == AL_TC && NS == 2), "8-column-group layout: 3 x 18 tiles, two row slots");
  const int b = a.list ? a.list[blockIdx.x + a.b0] : (int)blockIdx.x + a.b0, tid = threadIdx.x;
  if (b < 0 || (a.active && !a.active[b]) || (a.skip && a.skip[b])) return;
  const int n = a.n, m = a.m, n_e = a.n_e, n_c = a.n_c;

  // row vectors are indexed by the planner's LDS position of a row (rl_plan_build), core vectors by core index;
  // both keep an always-zero aligned pair for padded gathers (never written after the prologue)
  __shared__ __attribute__((aligned(16))) double s_tv[LCAP_M];                 // t'
  // core right-hand side: the TC entries of column group gj start at gj * TCP (TCP = TC rounded up to even), so a
  // thread reads its tile's entries with 16-byte LDS reads (half the LDS cycles of 8-byte pairs)
  constexpr int TCP = (RL_VARIANT & 4) ? TC : ((TC + 1) & ~1);
  static_assert((L8 ? 8 : LGJ) * TCP <= LCAP_NC + 16, "padded right-hand side");
  // aligned form: two buffers used in turn (one barrier per iteration: a fast wavefront writes the next right-hand side
  // while a slow one still reads the current one)
  constexpr int RVS = AL ? LCAP_NC + 16 : 0;
  __shared__ __attribute__((aligned(16))) double s_rv[(AL ? 2 : 1) * (LCAP_NC + 16)];
  __shared__ __attribute__((aligned(16))) double s_xc[LCAP_NC];                // x~_C
  __shared__ __attribute__((aligned(16))) double s_chk[2 * LCAP_M + 2 * LCAP_NC];   // check scratch: w*y, dy (rows), x_C, dx_C
  __shared__ double s_red[LWV * 8];
  double *swy = s_chk, *sdy = swy + LCAP_M, *sxc = sdy + LCAP_M, *sdxc = sxc + LCAP_NC;
  extern __shared__ double s_val[];

  // ---- prologue ----------------------------------------------------------------
  const double *gAs = a.As + (size_t)b * a.nnzA;
  {
    double *V = s_val;
    for (int p = tid; p < a.totAc; p += LT) { const int s = a.srcAc[p]; V[p] = s >= 0 ? gAs[s] : 0.0; }
    V += a.totAc;
#pragma unroll
    for (int q = 0; q < NS; q++) {
      for (int p = tid; p < a.totAr[q]; p += LT) { const int s = a.srcAr[q][p]; V[p] = s >= 0 ? gAs[s] : 0.0; }
      V += a.totAr[q];
    }
    for (int p = tid; p < 64 * 16; p += LT) V[p] = 0.0;
  }
  auto pack = [&](int slot) -> unsigned int {
    return (unsigned int)a.off[(size_t)slot * LT + tid] | ((unsigned int)a.off[(size_t)(slot + 1) * LT + tid] << 16);
  };
  constexpr int RW = CW > LCW ? LRW_MAX : LRW;     // value slots per row: the wide instantiation takes 5 operand pairs
  constexpr int RO = (RW / 2 + 1) / 2;
  unsigned int co[CW / 4], ro[NS][RO];             // pair offsets, two per register
  auto rl_dot_row = [&](int w, const double *V, unsigned int *o, const double *vec) __attribute__((always_inline)) { return rl_dot_row_w<RW, NS == 2>(w, V, o, vec); };
#pragma unroll
  for (int k = 0; k < CW / 4; k++) co[k] = pack(2 * k);
#pragma unroll
  for (int q = 0; q < NS; q++)
#pragma unroll
    for (int k = 0; k < RO; k++) ro[q][k] = pack(CW / 2 + q * (RW / 2) + 2 * k);
  const double *vcol = s_val + a.role[(size_t)8 * LT + tid];
  const double *vr[NS]; int wr[NS];
#pragma unroll
  for (int q = 0; q < NS; q++) {
    vr[q] = s_val + a.role[(size_t)ROLE_RBASE(q) * LT + tid];
    wr[q] = __builtin_amdgcn_readfirstlane(a.role[(size_t)ROLE_RWID(q) * LT + tid]);
  }
  const int wcol = __builtin_amdgcn_readfirstlane(a.role[(size_t)13 * LT + tid]);
  // W tile of this thread: row group gi (4 per wavefront: bits 2-3 of the lane), column group gj (bits 0-1 and
  // 4-5 of the lane: the 16 lanes whose partial sums rl_reduce_rows adds)
  // (aligned layout: 8 column groups = lane bits 0, 4, 5; the tile rows come from the role table)
  const int gi = (tid >> 6) * 4 + ((tid >> 2) & 3), gj = L8 ? (tid & 1) + 2 * ((tid >> 4) & 3) : (tid & 3) + 4 * ((tid >> 4) & 3);
  const int wrow_h = (tid >> 4) & 3;                 // DPP row of the lane: which tile rows' totals it receives
  double wreg[TR][TC];
  {
    const double *W = a.W + (size_t)b * n_c * n_c;
#pragma unroll
    for (int rr = 0; rr < TR; rr++)
#pragma unroll
      for (int cc = 0; cc < TC; cc++) {
        const int row = L8 ? a.role[(size_t)ROLE_WROW(rr) * LT + tid] : gi * TR + rr, col = gj * TC + cc;
        wreg[rr][cc] = (row >= 0 && row < n_c && col < n_c) ? W[(size_t)row * n_c + col] : 0.0;
      }
  }
  const int xout = L8 ? a.role[(size_t)ROLE_XOUT * LT + tid] : -1;     // 8 column groups: core index of the total this lane stores
  const bool wwave = !L8 || __builtin_amdgcn_readfirstlane(__any(a.role[(size_t)ROLE_WROW(0) * LT + tid] >= 0) ? 1 : 0) != 0;   // this wavefront carries W rows
  // core-variable state
  const int cown = a.role[tid];
  const int cvar = cown >= 0 ? a.role[(size_t)LT + tid] : 0;
  const int rvpos = cown >= 0 ? (cown / TC) * TCP + cown % TC : 0;     // slot of r_c in the padded right-hand side
  double *const rvw = s_rv + rvpos;                     // aligned form: an odd step uses the second buffer, RVS further on
  const dbl2 *const rvr = (const dbl2 *)(s_rv + gj * TCP);
  double xcv = 0.0, qc = 0.0;
  if (cown >= 0) qc = a.qs[(size_t)b * n + cvar];
  // eliminated-variable state
  const int eown = a.role[(size_t)2 * LT + tid];
  const int evar = eown >= 0 ? a.role[(size_t)3 * LT + tid] : 0;
  double xe = 0.0, qe = 0.0, kinv = 0.0, ge = 0.0;
  if (eown >= 0) { qe = a.qs[(size_t)b * n + evar]; kinv = a.kee_inv[(size_t)b * n_e + eown]; ge = -qe * kinv; }
  // row state
  int r_i[NS], r_p[NS]; double r_ls[NS], r_us[NS], r_rho[NS], r_rinv[NS], r_z[NS], r_y[NS], r_w[NS], r_ae[NS];
#pragma unroll
  for (int q = 0; q < NS; q++) {
    r_i[q] = a.role[(size_t)ROLE_ROW(q) * LT + tid];
    r_p[q] = a.role[(size_t)ROLE_POS(q) * LT + tid];       // LDS position of the row in t' / w y / dy
    r_ls[q] = r_us[q] = r_z[q] = r_y[q] = r_ae[q] = 0.0; r_rho[q] = r_rinv[q] = r_w[q] = 1.0;
    if (r_i[q] >= 0) {
      const int i = r_i[q];
      r_ls[q] = a.ls[(size_t)b * m + i]; r_us[q] = a.us[(size_t)b * m + i];
      r_rho[q] = a.rho[(size_t)b * m + i]; r_rinv[q] = 1.0 / r_rho[q];
      r_w[q] = (double)a.w[(size_t)b * m + i];
      // (a third slot never holds a row of an eliminated variable -- rl_plan_build -- so its a_e is the constant 0 and the
      // compiler drops the register and the terms that carry it)
      const int ep = q < 2 ? a.role[(size_t)ROLE_EPOS(q) * LT + tid] : -1;
      if (q < 2 && ep >= 0) r_ae[q] = gAs[ep];
    }
  }
  // which row slots this wavefront uses at all, and whether it owns an eliminated variable (wave-uniform)
  int rmask = 0;
  {
    bool any_row = __any(eown >= 0) != 0, upper = false;
#pragma unroll
    for (int q = 0; q < NS; q++) { const bool h = __any(r_p[q] >= 0) != 0; any_row |= h; if (q > 0) upper |= h; }
    // two slots: 0 = no rows, 1 = slot 0 only, 3 = both; three slots: 0 or all
    rmask = __builtin_amdgcn_readfirstlane(!any_row ? 0 : (NS == 2 && !upper) ? 1 : (1 << NS) - 1);
  }
  // per-thread constants of the termination test, parked in LDS (slot k of thread t at [k * LT + t]): the scalings
  // E of its two rows, D of its core / eliminated variable, the eliminated variable's P_ee.  Read back by the
  // same thread only; from global memory they cost the test ~2 us of exposed L2 latency every 25 iterations.
  // slots: CS_E + q = E of row slot q, CS_DC / CS_DE = D of the core / eliminated variable, CS_PEE = P_ee, then the
  // reciprocals CS_RE + q, CS_RDC, CS_RDE (the test divides by E and D in every row and column norm)
  constexpr int CS_E = 0, CS_DC = NS, CS_DE = NS + 1, CS_PEE = NS + 2, CS_RE = NS + 3, CS_RDC = 2 * NS + 3, CS_RDE = 2 * NS + 4, CS_N = 2 * NS + 5;
  int atot = a.totAc;
#pragma unroll
  for (int q = 0; q < NS; q++) atot += a.totAr[q];
  double *s_cst = s_val + atot + 64 * 16;
  {
    const double *Dg0 = a.D + (size_t)b * n, *Eg0 = a.E + (size_t)b * m;
#pragma unroll
    for (int q = 0; q < NS; q++) {
      s_cst[(CS_E + q) * LT + tid] = r_i[q] >= 0 ? Eg0[r_i[q]] : 1.0;
      s_cst[(CS_RE + q) * LT + tid] = 1.0 / s_cst[(CS_E + q) * LT + tid];
    }
    s_cst[CS_DC * LT + tid] = cown >= 0 ? Dg0[cvar] : 1.0;
    s_cst[CS_DE * LT + tid] = eown >= 0 ? Dg0[evar] : 1.0;
    const int pd0 = eown >= 0 ? a.role[(size_t)12 * LT + tid] : -1;
    s_cst[CS_PEE * LT + tid] = pd0 >= 0 ? (a.Ps + (size_t)b * a.nnzP)[pd0] : 0.0;
    s_cst[CS_RDC * LT + tid] = 1.0 / s_cst[CS_DC * LT + tid]; s_cst[CS_RDE * LT + tid] = 1.0 / s_cst[CS_DE * LT + tid];
  }
  // ... and the P entries of a core variable's column (value, core index; padded with 0 * x_C[n_c] = 0)
  double *s_pcv = s_cst + CS_N * LT;
  int *s_pci = (int *)(s_pcv + 4 * LCAP_NC);
  if (a.pcw <= 4 && cown >= 0) {
    const double *Ps0 = a.Ps + (size_t)b * a.nnzP;
    const int t0 = a.pc_ptr[cown], t1 = a.pc_ptr[cown + 1];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const bool on = t0 + k < t1;
      s_pcv[k * LCAP_NC + cown] = on ? Ps0[a.pc_pos[t0 + k]] : 0.0;
      s_pci[k * LCAP_NC + cown] = on ? a.pc_core[t0 + k] : n_c;
    }
  }
  if (tid < 4) { s_pcv[tid * LCAP_NC + n_c] = 0.0; s_pci[tid * LCAP_NC + n_c] = n_c; }     // the zero column: threads without a core variable
  for (int i = tid; i < LCAP_M; i += LT) s_tv[i] = 0.0;
  for (int i = tid; i < LCAP_NC; i += LT) s_xc[i] = 0.0;
  for (int i = tid; i < (AL ? 2 : 1) * (LCAP_NC + 16); i += LT) s_rv[i] = 0.0;
  for (int i = tid; i < 2 * LCAP_M + 2 * LCAP_NC; i += LT) s_chk[i] = 0.0;
  __syncthreads();
  const double cscale = a.cscale[b];
  // 1 / c for the termination test, formed once and kept in scalar registers (wave-uniform)
  const double cinv_v = 1.0 / cscale;
  const double cinv = __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(cinv_v)), __builtin_amdgcn_readfirstlane(__double2loint(cinv_v)));
  const double alpha = a.alpha, sigma = a.sigma;
  // compile-time since r03 (a wave-uniform RUN-TIME test of either in the loop cost 2.2 % and 1.1 % of an iteration,
  // profiles/r03_ab.txt section 8): only the aligned closed assignment drops barriers, and every other plan puts a core
  // column on a lane PAIR (owner + helper; SCO_QP_RL_SPLIT=0 leaves the helper without entries)
  constexpr bool merged = AL, colsplit = !AL;
  const int it0 = a.slice > 0 ? a.prog[b] : 0;
  if (it0 > 0) {
    // resume an unfinished solve: every loop-carried value comes back from memory
    if (cown >= 0) xcv = a.sx[(size_t)b * n + cvar];
    if (eown >= 0) { xe = a.sx[(size_t)b * n + evar]; ge = a.sg[(size_t)b * n_e + eown]; }
#pragma unroll
    for (int q = 0; q < NS; q++)
      if (r_i[q] >= 0) {
        r_z[q] = a.sz[(size_t)b * m + r_i[q]]; r_y[q] = a.sy[(size_t)b * m + r_i[q]];
        s_tv[r_p[q]] = a.st[(size_t)b * m + r_i[q]];
      }
    if (ADAPT && a.rflag[b]) {
      double tq[NS] = {};
#pragma unroll
      for (int q = 0; q < NS; q++)
        if (r_i[q] >= 0) tq[q] = r_w[q] * (r_rho[q] * r_z[q] - r_y[q]);
      if (eown >= 0) ge = ((sigma * xe - qe) + r_ae[0] * tq[0] + r_ae[1] * tq[1]) * kinv;
#pragma unroll
      for (int q = 0; q < NS; q++)
        if (r_i[q] >= 0) s_tv[r_p[q]] = tq[q] - (r_w[q] * r_rho[q]) * r_ae[q] * ge;
    }
  } else if (a.warm) {
    // OSQP-style warm start from the previous solution of this handle (x, y unscaled in a.x / a.y):
    //   x_s = x / D,  y_s = c y / (E w),  z = A_s x_s;  then t, g_e, t' as after any iteration
    const double *Dg = a.D + (size_t)b * n, *Eg = a.E + (size_t)b * m;
    if (cown >= 0) { xcv = a.x[(size_t)b * n + cvar] / Dg[cvar]; sxc[cown] = xcv; }
    if (eown >= 0) xe = a.x[(size_t)b * n + evar] / Dg[evar];
    __syncthreads();
    double axc[NS];
#pragma unroll
    for (int q = 0; q < NS; q++) axc[q] = rl_dot_row(wr[q], vr[q], ro[q], sxc);
    double tq[NS] = {};
#pragma unroll
    for (int q = 0; q < NS; q++)
      if (r_i[q] >= 0) {
        r_z[q] = axc[q] + r_ae[q] * xe;
        r_y[q] = a.y[(size_t)b * m + r_i[q]] * cscale / (Eg[r_i[q]] * r_w[q]);
        tq[q] = r_w[q] * (r_rho[q] * r_z[q] - r_y[q]);
      }
    if (eown >= 0) ge = ((sigma * xe - qe) + r_ae[0] * tq[0] + r_ae[1] * tq[1]) * kinv;
#pragma unroll
    for (int q = 0; q < NS; q++)
      if (r_i[q] >= 0) s_tv[r_p[q]] = tq[q] - (r_w[q] * r_rho[q]) * r_ae[q] * ge;
  } else {
    // t' of the start point x = z = y = 0:  t = 0, g_e = -q_e / K_ee, t'_i = -rw_i a_ie g_e
#pragma unroll
    for (int q = 0; q < NS; q++)
      if (r_i[q] >= 0) s_tv[r_p[q]] = -(r_w[q] * r_rho[q]) * r_ae[q] * ge;
  }
  __syncthreads();
  if (ADAPT && tid == 0) { a.rflag[b] = 0; a.smask[b] = 0; }

  // Indices that only the prologue, the rare branches of the termination test and the epilogue need are re-read from
  // the role table there instead of occupying registers for the whole solve (the loop had a scratch reload in the
  // column phase of every iteration).  A slot holds a row exactly when it has an LDS position.
#define RL_ROW(q) (a.role[(size_t)ROLE_ROW(q) * LT + tid])
#define RL_CVAR (a.role[(size_t)LT + tid])
#define RL_EVAR (a.role[(size_t)3 * LT + tid])
  int status = 0, iter = 0;
  double pri = 0.0, dua = 0.0;
  double rho_new = 0.0;       // ADAPT: > 0 = park now, this is the rho to continue with
#ifdef SCO_STAMP
  // diagnostic build only: cycles per phase per wavefront (never compiled into the product)
  long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_t = __builtin_readcyclecounter();
#define STAMP(k) { const long long now_ = __builtin_readcyclecounter(); st_acc[k] += now_ - st_t; st_t = now_; }
  long long ck_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ck_t = 0;     // segments of the termination test
#define CSTAMP0 { ck_t = __builtin_readcyclecounter(); ck_acc[7] += 1; }
#define CSTAMP(k) { const long long now_ = __builtin_readcyclecounter(); ck_acc[k] += now_ - ck_t; ck_t = now_; }
#else
#define STAMP(k)
#define CSTAMP0
#define CSTAMP(k)
#endif
  // One ADMM iteration.  `chk` is a compile-time constant at both call sites: the
  // unchecked copy forms a tight inner loop with its latch right behind it (the
  // far jump over the termination test cost ~900 cycles per iteration,
  // profiles/r01_v5_stamps.txt), the checked copy runs every `check`-th iteration.
  double dxe = 0.0, dxc = 0.0;
  // `k` (0 / 1, a constant at every call site): which right-hand-side buffer the step uses (aligned form)
  dbl2 pcol[CW / 2];          // RL_VARIANT & 128: the column values of the next phase (1), read ahead of its barrier
  auto step = [&](const bool chk, const int k) __attribute__((always_inline)) {
    STAMP(6)
    // (1) core right-hand side
    {
      double dv = (RL_ABLATE & 1) ? 0.0 : (RL_VARIANT & 128) ? rl_dot_col_pre<CW>(wcol, pcol, co, s_tv) : rl_dot_col<CW, NS == 2>(wcol, vcol, co, s_tv);
      if (colsplit) dv = rl_pair_sum(dv);
      if (cown >= 0) rvw[k * RVS] = (sigma * xcv - qc) + dv;
    }
    STAMP(0)
    if (!(RL_ABLATE & 16)) __syncthreads();
    STAMP(1)
    // (3) register-tile mat-vec; the 16 partial sums of a row sit in the 16 lanes of a
    //     DPP row and are added with row shifts (no LDS round trip, no extra barrier)
    if (wwave) {          // (8 column groups: two of the eight wavefronts hold no W rows and go straight to the barrier)
      double rr_[TC], acc[TR], tot[RL_NT(TR)];
      if constexpr (L8) {
#pragma unroll
        for (int cc = 0; cc < TC; cc += 2) { const dbl2 v2 = rvr[(k * RVS + cc) / 2]; rr_[cc] = v2.x; rr_[cc + 1] = v2.y; }
      } else {
#pragma unroll
        for (int cc = 0; cc < TC; cc++) rr_[cc] = s_rv[gj * TCP + cc];
      }
#pragma unroll
      for (int rr = 0; rr < TR; rr++) {
        acc[rr] = 0.0;
#pragma unroll
        for (int cc = 0; cc < ((RL_ABLATE & 4) ? 1 : TC); cc++) acc[rr] += wreg[rr][cc] * rr_[cc];
      }
      if constexpr (L8) {
        const double t = (RL_ABLATE & 8) ? acc[0] : rl_reduce_al(acc);
        if (xout >= 0) s_xc[xout] = t;
        (void)tot;
      } else {
      if (RL_ABLATE & 8) {
#pragma unroll
        for (int nn = 0; nn < RL_NT(TR); nn++) tot[nn] = acc[nn];
      } else
      rl_reduce_rows<TR>(acc, tot);
      if ((tid & 3) == 0) {
#pragma unroll
        for (int nn = 0; nn < RL_NT(TR); nn++) {
          const int rr = rl_tile_row(nn, wrow_h);
          if (rr < TR && gi * TR + rr < n_c) s_xc[gi * TR + rr] = tot[nn];
        }
      }
      }
    }
    dbl2 prow[NS][RW / 2];
    if (RL_VARIANT & 128) {
#pragma unroll
      for (int q = 0; q < NS; q++) rl_vals<RW / 2, false>(wr[q], vr[q], prow[q]);
    }
    STAMP(2)
    // aligned form: the totals a row gathers were stored by its own wavefront
    if constexpr (AL) RL_WAVE_SYNC() else if (!(RL_ABLATE & 16)) __syncthreads();
    STAMP(3)
    // (Y) rows, eliminated variable, updates, next t'
    {
      // `M`: the row slots this WAVEFRONT uses at all (wave-uniform; r03): a wavefront that only holds columns skips
      // the whole update chain, one whose rows all sit in slot 0 the second slot's half of it.  A skipped slot holds no
      // row in any lane: a_e = z = y = 0 there, so its terms are exact zeros and its state never changes.
      auto rows = [&](auto M) __attribute__((always_inline)) {
        constexpr int msk = decltype(M)::value;
        double zc[NS];
#pragma unroll
        for (int q = 0; q < NS; q++) zc[q] = !((msk >> q) & 1) ? 0.0 : (RL_ABLATE & 2) ? s_xc[q] : (RL_VARIANT & 128) ? rl_dot_row_pre<RW>(wr[q], prow[q], ro[q], s_xc) : rl_dot_row(wr[q], vr[q], ro[q], s_xc);
        // x~_e = g_e - (1/K_ee) sum_i rw_i a_ie (A_iC x~_C)
        const double xte = (msk & 2) ? ge - kinv * ((r_w[0] * r_rho[0]) * r_ae[0] * zc[0] + (r_w[1] * r_rho[1]) * r_ae[1] * zc[1])
                                     : ge - kinv * ((r_w[0] * r_rho[0]) * r_ae[0] * zc[0]);
        double tq[NS], dyq[NS];
#pragma unroll
        for (int q = 0; q < NS; q++) {
          tq[q] = dyq[q] = 0.0;
          if (!((msk >> q) & 1)) continue;
          const double zt = zc[q] + r_ae[q] * xte;
          const double zr = alpha * zt + (1.0 - alpha) * r_z[q];
          double zn = zr + r_rinv[q] * r_y[q];
          zn = fmin(fmax(zn, r_ls[q]), r_us[q]);
          dyq[q] = r_rho[q] * (zr - zn);
          r_y[q] += dyq[q]; r_z[q] = zn;
          tq[q] = r_w[q] * (r_rho[q] * zn - r_y[q]);
        }
        // No guard (r03; RL_VARIANT & 512 puts `if (eown >= 0)` back): a thread without an eliminated variable has
        // x_e = q_e = 1 / K_ee = a_e = 0, so every line below leaves exact zeros -- and the loop loses two selects and an
        // exec-mask region (1.045 -> 1.027 us per iteration, profiles/r03_ab.txt section 7)
        if (!(RL_VARIANT & 512) || eown >= 0) {
          const double xn = alpha * xte + (1.0 - alpha) * xe;
          if (chk) dxe = xn - xe;           // only the termination test reads the steps
          xe = xn;
          const double rhs_e = (msk & 2) ? (sigma * xe - qe) + r_ae[0] * tq[0] + r_ae[1] * tq[1] : (sigma * xe - qe) + r_ae[0] * tq[0];
          ge = rhs_e * kinv;
        }
#pragma unroll
        for (int q = 0; q < NS; q++) {
          if constexpr ((RL_VARIANT & 4096) == 0) {
            // no exec-mask region around the store of every iteration (r03): an empty slot writes its t' to position
            // LCAP_M - 1, which no row owns (rl_plan_build) and nobody gathers: 1.027 -> 1.011 us per iteration
            if ((msk >> q) & 1) s_tv[r_p[q] >= 0 ? r_p[q] : LCAP_M - 1] = tq[q] - (r_w[q] * r_rho[q]) * r_ae[q] * ge;
            if (chk && ((msk >> q) & 1) && r_p[q] >= 0) { swy[r_p[q]] = r_w[q] * r_y[q]; sdy[r_p[q]] = dyq[q]; }
          } else
          if (((msk >> q) & 1) && r_p[q] >= 0) {
            s_tv[r_p[q]] = tq[q] - (r_w[q] * r_rho[q]) * r_ae[q] * ge;
            if (chk) { swy[r_p[q]] = r_w[q] * r_y[q]; sdy[r_p[q]] = dyq[q]; }
          }
        }
      };
      if (!(RL_VARIANT & 64)) rows(std::integral_constant<int, (1 << NS) - 1>());
      else if (NS == 2 && rmask == 1) rows(std::integral_constant<int, 1>());
      else if (rmask != 0) rows(std::integral_constant<int, (1 << NS) - 1>());
      if (cown >= 0) {
        const double xn = alpha * s_xc[cown] + (1.0 - alpha) * xcv;
        if (chk) dxc = xn - xcv;
        xcv = xn;
        if (chk) { sxc[cown] = xn; sdxc[cown] = dxc; }
      }
      if ((RL_VARIANT & 128) && !chk) rl_vals<CW / 2, true>(wcol, vcol, pcol);      // for the next step's phase (1)
      STAMP(4)
      // closed assignment: the next phase (1) reads only t' written by its own wavefront (LDS operations of one
      // wavefront complete in order), every other hazard is covered by the two remaining barriers; the
      // termination test after a checked step reads what all wavefronts have just written
      if ((chk || !merged) && (chk || !(RL_ABLATE & 16))) __syncthreads();
      else RL_WAVE_SYNC()
      STAMP(5)
    }
  };
  iter = it0;
  if ((RL_VARIANT & 1) && tid >= LT / 2) __builtin_amdgcn_s_setprio(1);
  const int stop = (a.slice > 0 && it0 + a.slice < a.max_iter) ? it0 + a.slice : a.max_iter;
  while (!status && iter < stop && !(ADAPT && rho_new > 0.0)) {
    int next = stop;
    if (a.check > 0) { next = (iter / a.check + 1) * a.check; if (next > stop) next = stop; }
    if (RL_VARIANT & 128) rl_vals<CW / 2, true>(wcol, vcol, pcol);
    // twelve iterations per trip (two trips between termination tests at the default cadence of 25): a loop trip
    // costs several hundred cycles of instruction refetch (profiles/r01_v6_stamps.txt); 24 copies overflow the
    // instruction cache and are slower (1.33 against 1.26 us per iteration)
    if (!(RL_VARIANT & 8))
    while (iter + 12 < next) { iter += 12; step(false, 0); step(false, 1); step(false, 0); step(false, 1); step(false, 0); step(false, 1); step(false, 0); step(false, 1); step(false, 0); step(false, 1); step(false, 0); step(false, 1); }
    while (iter + 8 < next) { iter += 8; step(false, 0); step(false, 1); step(false, 0); step(false, 1); step(false, 0); step(false, 1); step(false, 0); step(false, 1); }
    while (iter + 4 < next) { iter += 4; step(false, 0); step(false, 1); step(false, 0); step(false, 1); }
    // the steps come in pairs (first buffer, second buffer); a single step is followed by a barrier, so that the first
    // buffer can be written again at once (the checked step ends with one anyway)
    while (iter + 1 < next) { iter++; step(false, 0); if constexpr (AL) __syncthreads(); }
    iter++; if (RL_VARIANT & 32) { step(false, 0); if constexpr (AL) __syncthreads(); } else step(true, 0);
    {
      // ---- termination test (formulas of admm_check in sco_qp.hip) ---------------------
      CSTAMP0
      const bool adapt_pt = ADAPT && iter % a.ad_interval == 0 && iter < a.max_iter;
      double vs[7] = {0, 0, 0, 0, 0, 0, 0};       // ADAPT: the same norms of the SCALED iterates
      for (int approximate = 0; approximate < 2 && !status && !(RL_VARIANT & 16); approximate++) {
        if (approximate && iter < a.max_iter) break;
        const double *Ps = a.Ps + (size_t)b * a.nnzP;
        const double *Dg = a.D + (size_t)b * n, *Eg = a.E + (size_t)b * m;
        double ea = a.eps_abs, er = a.eps_rel, epi = a.eps_prim_inf, edi = a.eps_dual_inf;
        if (approximate) { ea *= 10; er *= 10; epi *= 10; edi *= 10; }
        // Branch-free: a slot without a row has a_e = z = y = 0, zero gather offsets and scaling 1, a thread without a
        // core / eliminated variable has x = q = 0, scaling 1 and the zero entries of the P table, so every term below is
        // an exact zero there (r02: the predicated form spent 180 of its 730 instructions on 64-bit register moves).
        // |c x| = c |x| and max(c a, c b) = c max(a, b) hold exactly for c > 0, so the pairs of norms that the tolerances
        // only use through their maximum are formed with one multiplication.
        double w4[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        double dyp[NS];
        {
          double axc[NS];
#pragma unroll
          for (int q = 0; q < NS; q++) axc[q] = rl_dot_row(wr[q], vr[q], ro[q], sxc);
          CSTAMP(0)
#pragma unroll
          for (int q = 0; q < NS; q++) {
            const double ax = axc[q] + r_ae[q] * xe;
            const double ei = s_cst[(CS_RE + q) * LT + tid];
            w4[0] = fmax(w4[0], ei * fabs(ax - r_z[q]));
            w4[1] = fmax(w4[1], ei * fmax(fabs(r_z[q]), fabs(ax)));
            if (ADAPT && adapt_pt) {
              vs[0] = fmax(vs[0], fabs(ax - r_z[q])); vs[1] = fmax(vs[1], fabs(r_z[q])); vs[2] = fmax(vs[2], fabs(ax));
            }
            // dy clipped to the cone of the bounds (a slot without a row reads the always-zero position)
            const double dy0 = sdy[r_p[q] >= 0 ? r_p[q] : a.zpos];
            const double dy1 = r_us[q] > SCO_INFTY * SCO_MIN_SCALING ? fmin(dy0, 0.0) : dy0;
            const double dy = r_ls[q] < -SCO_INFTY * SCO_MIN_SCALING ? fmax(dy1, 0.0) : dy1;
            dyp[q] = dy;
            w4[4] = fmax(w4[4], s_cst[(CS_E + q) * LT + tid] * fabs(dy));
            w4[6] += r_w[q] * (r_us[q] * fmax(dy, 0.0) + r_ls[q] * fmin(dy, 0.0));
          }
        }
        double aty_c = rl_dot_col<CW, NS == 2>(wcol, vcol, co, swy);     // whole wave: the trip count is wave-uniform
        if (colsplit) { const double both = rl_pair_sum(aty_c); aty_c = cown >= 0 ? both : 0.0; }   // a helper lane owns no column
        CSTAMP(1)
        {
          const int cix = cown >= 0 ? cown : n_c;        // the table's zero column
          double px = 0.0;
          if (a.pcw <= 4) {
#pragma unroll
            for (int k = 0; k < 4; k++) px += s_pcv[k * LCAP_NC + cix] * sxc[s_pci[k * LCAP_NC + cix]];
          } else if (cown >= 0) {
            for (int t = a.pc_ptr[cown]; t < a.pc_ptr[cown + 1]; t++) px += Ps[a.pc_pos[t]] * sxc[a.pc_core[t]];
          }
          const double dj = s_cst[CS_RDC * LT + tid];
          w4[2] = dj * fabs(qc + px + aty_c);
          w4[3] = dj * fmax(fabs(qc), fmax(fabs(aty_c), fabs(px)));
          if (ADAPT && adapt_pt) { vs[3] = fabs(qc + px + aty_c); vs[4] = fabs(qc); vs[5] = fabs(aty_c); vs[6] = fabs(px); }
        }
        {
          const double px = s_cst[CS_PEE * LT + tid] * xe;
          const double aty = r_ae[0] * (r_w[0] * r_y[0]) + r_ae[1] * (r_w[1] * r_y[1]);
          const double dj = s_cst[CS_RDE * LT + tid];
          w4[2] = fmax(w4[2], dj * fabs(qe + px + aty));
          w4[3] = fmax(w4[3], dj * fmax(fabs(qe), fmax(fabs(aty), fabs(px))));
          if (ADAPT && adapt_pt) {
            vs[3] = fmax(vs[3], fabs(qe + px + aty)); vs[4] = fmax(vs[4], fabs(qe));
            vs[5] = fmax(vs[5], fabs(aty)); vs[6] = fmax(vs[6], fabs(px));
          }
        }
        CSTAMP(2)
        // Everything that opens the two infeasibility tests rides along in the same block reduction (two barriers in
        // all): the norms |E dy| and |D dx| and the two sums u' dy+ + l' dy- and q' dx.  On an unconverged iterate the
        // test used to make three reductions one after the other.
        w4[5] = fmax(s_cst[CS_DC * LT + tid] * fabs(dxc), s_cst[CS_DE * LT + tid] * fabs(dxe));
        w4[7] = qc * dxc + qe * dxe;
        CSTAMP(3)
        lblock_max6_sum2(w4, s_red);
        CSTAMP(4)
        pri = w4[0]; dua = cinv * w4[2];
        if (!(pri <= SCO_INFTY) || !(dua <= SCO_INFTY)) { status = SCO_QP_NON_CVX; break; }
        const double eps_p = ea + er * w4[1];
        const double eps_d = ea + er * cinv * w4[3];
        const bool prim_ok = (m == 0) || (pri < eps_p), dual_ok = dua < eps_d;
        if (prim_ok && dual_ok) { status = approximate ? SCO_QP_SOLVED_INACCURATE : SCO_QP_SOLVED; break; }
        if (!prim_ok) {            // primal infeasibility certificate from delta_y
          const double ndy = w4[4];
          if (ndy > epi && w4[6] < -epi * ndy) {
            {
              __syncthreads();
#pragma unroll
              for (int q = 0; q < NS; q++) if (r_p[q] >= 0) swy[r_p[q]] = r_w[q] * dyp[q];
              __syncthreads();
              double nat[1] = {0.0};
              {
                double dv = rl_dot_col<CW, NS == 2>(wcol, vcol, co, swy);
                if (colsplit) dv = rl_pair_sum(dv);
                if (cown >= 0) nat[0] = fabs(dv / Dg[RL_CVAR]);
              }
              if (eown >= 0) nat[0] = fmax(nat[0], fabs((r_ae[0] * (r_w[0] * dyp[0]) + r_ae[1] * (r_w[1] * dyp[1])) / Dg[RL_EVAR]));
              lblock_reduce<1, true>(nat, s_red);
#pragma unroll
              for (int q = 0; q < NS; q++) if (r_p[q] >= 0) swy[r_p[q]] = r_w[q] * r_y[q];
              __syncthreads();
              if (nat[0] < epi * ndy) { status = approximate ? SCO_QP_PRIMAL_INFEASIBLE_INACCURATE : SCO_QP_PRIMAL_INFEASIBLE; break; }
            }
          }
        }
        CSTAMP(5)
        if (!dual_ok) {            // dual infeasibility certificate from delta_x
          const double ndx = w4[5];
          if (ndx > edi) {
            if (w4[7] < -cscale * edi * ndx) {
              double npx[1] = {0.0};
              if (cown >= 0) {
                double px = 0.0;
                for (int t = a.pc_ptr[cown]; t < a.pc_ptr[cown + 1]; t++) px += Ps[a.pc_pos[t]] * sdxc[a.pc_core[t]];
                npx[0] = fabs(px / Dg[RL_CVAR]);
              }
              if (eown >= 0) {
                const int pd = a.role[(size_t)12 * LT + tid];
                if (pd >= 0) npx[0] = fmax(npx[0], fabs(Ps[pd] * dxe / Dg[RL_EVAR]));
              }
              lblock_reduce<1, true>(npx, s_red);
              if (npx[0] < cscale * edi * ndx) {
                double bad[1] = {0.0};
                double adc[NS];
#pragma unroll
                for (int q = 0; q < NS; q++) adc[q] = rl_dot_row(wr[q], vr[q], ro[q], sdxc);
#pragma unroll
                for (int q = 0; q < NS; q++)
                  if (r_p[q] >= 0) {
                    const double adx = (adc[q] + r_ae[q] * dxe) / Eg[RL_ROW(q)];
                    if ((r_us[q] < SCO_INFTY * SCO_MIN_SCALING && adx > edi * ndx) ||
                        (r_ls[q] > -SCO_INFTY * SCO_MIN_SCALING && adx < -edi * ndx)) bad[0] = 1.0;
                  }
                lblock_reduce<1, true>(bad, s_red);
                if (bad[0] == 0.0) { status = approximate ? SCO_QP_DUAL_INFEASIBLE_INACCURATE : SCO_QP_DUAL_INFEASIBLE; break; }
              }
            }
          }
        }
      }
      __syncthreads();
      CSTAMP(6)
      if (ADAPT && adapt_pt && !status) {
        // OSQP's rho estimate (compute_rho_estimate / adapt_rho of osqp 0.6, as recalled; oracle/osqp_ref.c)
        lblock_reduce<7, true>(vs, s_red);
        const double rho = a.rho_b[b];
        const double pn = vs[0] / (fmax(vs[1], vs[2]) + 1e-10);
        const double dn = vs[3] / (fmax(vs[4], fmax(vs[5], vs[6])) + 1e-10);
        const double est = fmin(fmax(rho * sqrt(pn / (dn + 1e-10)), SCO_RHO_MIN), 1e6);
        if (est > rho * a.ad_tol || est < rho / a.ad_tol) rho_new = est;
      }
    }
  }
  if (!status && iter < a.max_iter) {
    // the slice is used up (or rho changes): park the solve
    if (ADAPT && rho_new > 0.0 && tid == 0) { a.rho_b[b] = rho_new; a.rflag[b] = 1; a.smask[b] = 1; a.nupd[b] += 1; }
    if (cown >= 0) a.sx[(size_t)b * n + RL_CVAR] = xcv;
    if (eown >= 0) { a.sx[(size_t)b * n + RL_EVAR] = xe; a.sg[(size_t)b * n_e + eown] = ge; }
#pragma unroll
    for (int q = 0; q < NS; q++)
      if (r_p[q] >= 0) {
        a.sz[(size_t)b * m + RL_ROW(q)] = r_z[q]; a.sy[(size_t)b * m + RL_ROW(q)] = r_y[q];
        a.st[(size_t)b * m + RL_ROW(q)] = s_tv[r_p[q]];
      }
    if (tid == 0) { a.prog[b] = iter; a.status[b] = 0; a.iters[b] = iter; }
    return;
  }
  if (a.slice > 0 && tid == 0) a.prog[b] = 0;
  if (!status) status = SCO_QP_MAX_ITER_REACHED;
  if (iter > a.max_iter) iter = a.max_iter;
#ifdef SCO_STAMP
  if ((tid & 63) == 0 && b == 0 && a.stamp) {
    for (int k = 0; k < 7; k++) a.stamp[(tid >> 6) * 8 + k] = (double)st_acc[k];
    for (int k = 0; k < 8; k++) a.stamp[64 + (tid >> 6) * 8 + k] = (double)ck_acc[k];
    a.stamp[(tid >> 6) * 8 + 7] = (double)iter;
  }
#endif
  {
    const double *Dg = a.D + (size_t)b * n, *Eg = a.E + (size_t)b * m;
    const double cinv = 1.0 / cscale;
    if (cown >= 0) a.x[(size_t)b * n + RL_CVAR] = Dg[RL_CVAR] * xcv;
    if (eown >= 0) a.x[(size_t)b * n + RL_EVAR] = Dg[RL_EVAR] * xe;
#pragma unroll
    for (int q = 0; q < NS; q++)
      if (r_p[q] >= 0) a.y[(size_t)b * m + RL_ROW(q)] = cinv * Eg[RL_ROW(q)] * r_y[q] * r_w[q];
    if (tid == 0) {
      a.status[b] = status; a.iters[b] = iter;
      a.resid[2 * (size_t)b] = pri; a.resid[2 * (size_t)b + 1] = dua;
    }
  }
}

#undef RL_ROW
#undef RL_CVAR
#undef RL_EVAR

// --------------------------------------------------------------------------
// host glue
// --------------------------------------------------------------------------
#ifdef SCO_STAMP
double *sco_debug_stamp_ptr = nullptr;
extern "C" int sco_debug_stamps(double *out) {      // diagnostic build only
  if (!sco_debug_stamp_ptr) return -1;
  return hipMemcpy(out, sco_debug_stamp_ptr, 128 * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess ? 0 : -2;
}
#endif
template <typename T>
static int upl(std::vector<void *> &allocs, const std::vector<T> &v, const T **out) {
  void *p = nullptr;
  size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
  SCO_HIP(hipMalloc(&p, bytes));
  allocs.push_back(p);
  if (!v.empty()) SCO_HIP(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = (const T *)p;
  return SCO_OK;
}

int rl_upload(const RlHost &rh, std::vector<void *> &allocs, RlDev &rd) {
  int rc;
  if ((rc = upl(allocs, rh.off, &rd.off))) return rc;
  if ((rc = upl(allocs, rh.role, &rd.role))) return rc;
  if ((rc = upl(allocs, rh.Ac.src, &rd.srcAc))) return rc;
  for (int q = 0; q < LNS_MAX; q++)
    if ((rc = upl(allocs, rh.Ar[q].src, &rd.srcAr[q]))) return rc;
  if ((rc = upl(allocs, rh.pc_ptr, &rd.pc_ptr))) return rc;
  if ((rc = upl(allocs, rh.pc_pos, &rd.pc_pos))) return rc;
  if ((rc = upl(allocs, rh.pc_core, &rd.pc_core))) return rc;
  return SCO_OK;
}

template <int TR, int TC, int CW, bool ADAPT, int NS = 2, int LAY = 0>
static int rl_launch_k(const RlArgs &ra, int batch, size_t lds, hipStream_t st) {
  // hipFuncSetAttribute applies to the current device only
  static bool attr_done[64] = {};
  int dev_ = 0;
  (void)hipGetDevice(&dev_);
  dev_ &= 63;
  if (!attr_done[dev_]) {
    SCO_HIP(hipFuncSetAttribute((const void *)qp_admm_rl_kernel<TR, TC, CW, ADAPT, NS, LAY>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                108 * 1024));
    attr_done[dev_] = true;
  }
  hipLaunchKernelGGL((qp_admm_rl_kernel<TR, TC, CW, ADAPT, NS, LAY>), dim3(batch), dim3(LT), lds, st, ra);
  SCO_HIP(hipGetLastError());
  return SCO_OK;
}
template <int TR, int TC, int CW = LCW, int NS = 2, int LAY = 0>
static int rl_launch_one(const RlArgs &ra, int batch, size_t lds, hipStream_t st) {
  // the adaptive-rho variant is a separate instantiation: the default kernel's code is untouched by it
  return ra.ad_interval > 0 ? rl_launch_k<TR, TC, CW, true, NS, LAY>(ra, batch, lds, st) : rl_launch_k<TR, TC, CW, false, NS, LAY>(ra, batch, lds, st);
}

int rl_launch(const AdmmArgs &a, const RlHost &rh, const RlDev &rd, hipStream_t st) {
  const QpDev &d = a.d;
  RlArgs ra;
  ra.n = d.n; ra.m = d.m; ra.n_e = d.n_e; ra.n_c = d.n_c; ra.nnzA = d.nnzA; ra.nnzP = d.nnzP;
  ra.max_iter = a.max_iter; ra.check = a.check; ra.b0 = d.b0; ra.list = d.list;
  const int nwg = d.nb > 0 ? d.nb : d.batch;
  ra.sigma = a.sigma; ra.alpha = a.alpha; ra.eps_abs = a.eps_abs; ra.eps_rel = a.eps_rel;
  ra.eps_prim_inf = a.eps_prim_inf; ra.eps_dual_inf = a.eps_dual_inf;
  ra.off = rd.off; ra.role = rd.role; ra.srcAc = rd.srcAc;
  for (int q = 0; q < LNS_MAX; q++) { ra.srcAr[q] = rd.srcAr[q]; ra.totAr[q] = q < rh.NS ? rh.Ar[q].total : 0; }
  ra.pc_ptr = rd.pc_ptr; ra.pc_pos = rd.pc_pos; ra.pc_core = rd.pc_core; ra.pcw = rh.pcw; ra.zpos = rh.zpos;
  ra.totAc = rh.Ac.total;
  ra.As = d.As; ra.W = d.W; ra.qs = d.qs; ra.kee_inv = d.kee_inv; ra.ls = d.ls; ra.us = d.us; ra.rho = d.rho;
  ra.cscale = d.cscale; ra.Ps = d.Ps; ra.D = d.D; ra.E = d.E; ra.w = d.w; ra.active = d.active;
  ra.x = d.x; ra.y = d.y; ra.resid = d.resid; ra.status = d.status; ra.iters = d.iters;
  ra.warm = a.warm; ra.skip = a.skip;
  ra.slice = a.slice; ra.prog = d.prog;
  ra.ad_interval = a.adaptive ? a.ad_interval : 0; ra.ad_tol = a.ad_tol;
  ra.rho_b = d.rho_b; ra.rflag = d.rflag; ra.smask = d.smask; ra.nupd = d.nupd; ra.sx = d.sx; ra.sz = d.sz; ra.sy = d.sy; ra.st = d.st; ra.sg = d.sg;
  ra.stamp = nullptr;
#ifdef SCO_STAMP
  {
    static double *g_stamp = nullptr;
    if (!g_stamp) { SCO_HIP(hipMalloc((void **)&g_stamp, 128 * sizeof(double))); SCO_HIP(hipMemset(g_stamp, 0, 128 * sizeof(double))); }
    ra.stamp = g_stamp;
    extern double *sco_debug_stamp_ptr; sco_debug_stamp_ptr = g_stamp;
  }
#endif
  if (rh.aligned) {
    // aligned closed assignment: 3 x 18 tiles, one barrier per iteration
    if (rh.CW > LCW) return rl_launch_one<AL_TR, AL_TC, LCW_MAX, 2, 2>(ra, nwg, rh.lds_bytes, st);
    return rl_launch_one<AL_TR, AL_TC, LCW, 2, 2>(ra, nwg, rh.lds_bytes, st);
  }
  if (rh.lay8) {
    // 8 column groups, open assignment (three barriers)
    if (rh.CW > LCW) return rl_launch_one<AL_TR, AL_TC, LCW_MAX, 2, 1>(ra, nwg, rh.lds_bytes, st);
    return rl_launch_one<AL_TR, AL_TC, LCW, 2, 1>(ra, nwg, rh.lds_bytes, st);
  }
  if (rh.NS == 3) {
    // three row slots per thread and ten operand pairs per column (velocity + joint limits at 7-DOF x 20: 1100 rows)
    if (rh.CW == LCW && rh.TR == 5 && rh.TC == 9) return rl_launch_one<5, 9, LCW, 3>(ra, nwg, rh.lds_bytes, st);
    if (rh.CW == LCW && rh.TR == 5 && rh.TC == 10) return rl_launch_one<5, 10, LCW, 3>(ra, nwg, rh.lds_bytes, st);
    if (rh.TR == 5 && rh.TC == 9) return rl_launch_one<5, 9, LCW_MAX3, 3>(ra, nwg, rh.lds_bytes, st);
    if (rh.TR == 5 && rh.TC == 10) return rl_launch_one<5, 10, LCW_MAX3, 3>(ra, nwg, rh.lds_bytes, st);
    sco_set_error("rl_launch: unsupported tile");
    return SCO_ERR_CAPACITY;
  }
  if (rh.CW > LCW) {
    if (rh.TR == 5 && rh.TC == 9) return rl_launch_one<5, 9, LCW_MAX>(ra, nwg, rh.lds_bytes, st);
    if (rh.TR == 5 && rh.TC == 10) return rl_launch_one<5, 10, LCW_MAX>(ra, nwg, rh.lds_bytes, st);
    sco_set_error("rl_launch: unsupported tile");
    return SCO_ERR_CAPACITY;
  }
  switch (rh.TR * 100 + rh.TC) {
    case 102: return rl_launch_one<1, 2>(ra, nwg, rh.lds_bytes, st);
    case 204: return rl_launch_one<2, 4>(ra, nwg, rh.lds_bytes, st);
    case 306: return rl_launch_one<3, 6>(ra, nwg, rh.lds_bytes, st);
    case 408: return rl_launch_one<4, 8>(ra, nwg, rh.lds_bytes, st);
    case 509: return rl_launch_one<5, 9>(ra, nwg, rh.lds_bytes, st);
    case 510: return rl_launch_one<5, 10>(ra, nwg, rh.lds_bytes, st);
  }
  sco_set_error("rl_launch: unsupported tile");
  return SCO_ERR_CAPACITY;
}
