// qp_plan.cpp -- see qp_plan.h.  Pure host C++, no HIP calls.
#include "qp_plan.h"

#include <algorithm>
#include <numeric>

static bool csc_ok(int ncol, int nrow, const int *p, const int *i, bool upper) {
  if (p[0] != 0) return false;
  for (int j = 0; j < ncol; j++) {
    if (p[j + 1] < p[j]) return false;
    for (int t = p[j]; t < p[j + 1]; t++) {
      if (i[t] < 0 || i[t] >= nrow) return false;
      if (t > p[j] && i[t] <= i[t - 1]) return false;
      if (upper && i[t] > j) return false;
    }
  }
  return true;
}

int qp_plan_build(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                  int allow_elim, QpPlan &pl) {
  if (n < 0 || m < 0 || !Pp || !Ap) return -1;
  if (!csc_ok(n, n, Pp, Pi, true) || !csc_ok(n, m, Ap, Ai, false)) return -1;
  pl = QpPlan();
  pl.n = n; pl.m = m; pl.nnzP = Pp[n]; pl.nnzA = Ap[n];
  pl.Pp.assign(Pp, Pp + n + 1); pl.Pi.assign(Pi, Pi + pl.nnzP);
  pl.Ap.assign(Ap, Ap + n + 1); pl.Ai.assign(Ai, Ai + pl.nnzA);

  // ---- CSR view of A -------------------------------------------------------
  pl.Rp.assign(m + 1, 0);
  for (int t = 0; t < pl.nnzA; t++) pl.Rp[Ai[t] + 1]++;
  for (int i = 0; i < m; i++) pl.Rp[i + 1] += pl.Rp[i];
  pl.Rj.resize(pl.nnzA); pl.Rpos.resize(pl.nnzA);
  {
    std::vector<int> cur(pl.Rp.begin(), pl.Rp.end() - 1);
    for (int j = 0; j < n; j++)
      for (int t = Ap[j]; t < Ap[j + 1]; t++) {
        int i = Ai[t];
        pl.Rj[cur[i]] = j; pl.Rpos[cur[i]] = t; cur[i]++;
      }
  }
  // ---- full symmetric P by column -------------------------------------------
  pl.Pdiag.assign(n, -1);
  pl.Fp.assign(n + 1, 0);
  for (int j = 0; j < n; j++)
    for (int t = Pp[j]; t < Pp[j + 1]; t++) {
      int i = Pi[t];
      pl.Fp[j + 1]++;
      if (i != j) pl.Fp[i + 1]++; else pl.Pdiag[j] = t;
    }
  for (int j = 0; j < n; j++) pl.Fp[j + 1] += pl.Fp[j];
  pl.Fi.resize(pl.Fp[n]); pl.Fpos.resize(pl.Fp[n]);
  {
    std::vector<int> cur(pl.Fp.begin(), pl.Fp.end() - 1);
    // rows i < j of column j come from the triu column j; rows i > j come from
    // triu columns i > j, visited in increasing column order => rows stay sorted.
    for (int j = 0; j < n; j++)
      for (int t = Pp[j]; t < Pp[j + 1]; t++) {
        int i = Pi[t];
        pl.Fi[cur[j]] = i; pl.Fpos[cur[j]] = t; cur[j]++;
        if (i != j) { pl.Fi[cur[i]] = j; pl.Fpos[cur[i]] = t; cur[i]++; }
      }
  }
  // ---- choose the eliminated set E -------------------------------------------
  std::vector<char> has_offdiag(n, 0);
  for (int j = 0; j < n; j++)
    for (int t = Pp[j]; t < Pp[j + 1]; t++)
      if (Pi[t] != j) { has_offdiag[j] = 1; has_offdiag[Pi[t]] = 1; }
  // degree of j in K's graph restricted to the A'A part (P part is empty for candidates)
  std::vector<int> mark(n, -1), deg(n, 0);
  for (int j = 0; j < n; j++) {
    int d = 0;
    for (int t = Ap[j]; t < Ap[j + 1]; t++) {
      int i = Ai[t];
      for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
        int k = pl.Rj[s];
        if (k != j && mark[k] != j) { mark[k] = j; d++; }
      }
    }
    deg[j] = d;
  }
  std::vector<int> cand;
  // allow_elim == 2: only variables that sit in at most two rows (what the row-local kernels can keep
  // in one thread); == 1: any variable without an off-diagonal P entry
  if (allow_elim)
    for (int j = 0; j < n; j++)
      if (!has_offdiag[j] && (allow_elim != 2 || Ap[j + 1] - Ap[j] <= 2)) cand.push_back(j);
  std::stable_sort(cand.begin(), cand.end(), [&](int a, int b) { return deg[a] < deg[b]; });
  std::vector<char> in_e(n, 0), blocked(n, 0);
  for (int j : cand) {
    if (blocked[j]) continue;
    // Eliminating an isolated variable (deg 0) is pointless but harmless; keep it
    // in the core only when the core would otherwise be empty.
    in_e[j] = 1;
    for (int t = Ap[j]; t < Ap[j + 1]; t++) {
      int i = Ai[t];
      for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) blocked[pl.Rj[s]] = 1;
    }
  }
  pl.elim_of.assign(n, -1); pl.core_of.assign(n, -1);
  for (int j = 0; j < n; j++) {
    if (in_e[j]) { pl.elim_of[j] = (int)pl.elim_var.size(); pl.elim_var.push_back(j); }
    else { pl.core_of[j] = (int)pl.core_var.size(); pl.core_var.push_back(j); }
  }
  pl.n_e = (int)pl.elim_var.size(); pl.n_c = (int)pl.core_var.size();

  // ---- coupling pairs (a in C, e in E) ---------------------------------------
  pl.e_ptr.assign(pl.n_e + 1, 0);
  std::vector<std::vector<int>> contrib;   // per pair: flattened (row, pa, pe)
  {
    std::vector<int> slot(pl.n_c, -1);
    for (int ei = 0; ei < pl.n_e; ei++) {
      int ve = pl.elim_var[ei];
      int first = (int)pl.pair_core.size();
      for (int t = Ap[ve]; t < Ap[ve + 1]; t++) {
        int i = Ai[t];
        for (int s = pl.Rp[i]; s < pl.Rp[i + 1]; s++) {
          int k = pl.Rj[s];
          if (k == ve) continue;
          int a = pl.core_of[k];
          if (a < 0) return -2;   // two eliminated variables share a row: E not independent
          if (slot[a] < first) {
            slot[a] = (int)pl.pair_core.size();
            pl.pair_core.push_back(a); pl.pair_elim.push_back(ei);
            contrib.emplace_back();
          }
          auto &c = contrib[slot[a]];
          c.push_back(i); c.push_back(pl.Rpos[s]); c.push_back(t);
        }
      }
      pl.e_ptr[ei + 1] = (int)pl.pair_core.size();
      for (int k = first; k < (int)pl.pair_core.size(); k++) slot[pl.pair_core[k]] = -1;
    }
  }
  pl.ncpl = (int)pl.pair_core.size();
  pl.cp_ptr.assign(pl.ncpl + 1, 0);
  for (int k = 0; k < pl.ncpl; k++) {
    pl.cp_ptr[k + 1] = pl.cp_ptr[k] + (int)contrib[k].size() / 3;
    for (size_t t = 0; t < contrib[k].size(); t += 3) {
      pl.cp_row.push_back(contrib[k][t]); pl.cp_pa.push_back(contrib[k][t + 1]); pl.cp_pe.push_back(contrib[k][t + 2]);
    }
  }
  // pairs grouped by core index
  pl.a_ptr.assign(pl.n_c + 1, 0);
  for (int k = 0; k < pl.ncpl; k++) pl.a_ptr[pl.pair_core[k] + 1]++;
  for (int a = 0; a < pl.n_c; a++) pl.a_ptr[a + 1] += pl.a_ptr[a];
  pl.a_pair.resize(pl.ncpl);
  {
    std::vector<int> cur(pl.a_ptr.begin(), pl.a_ptr.end() - 1);
    for (int k = 0; k < pl.ncpl; k++) pl.a_pair[cur[pl.pair_core[k]]++] = k;
  }

  // ---- structurally non-zero entries of S (a >= b) ---------------------------
  // entry ids through a hash on (a, b); n_c is at most a few thousand.
  std::vector<std::vector<int>> row_entries(pl.n_c);   // per a: list of (b, id) pairs flattened
  auto entry_id = [&](int a, int b) -> int {
    if (a < b) std::swap(a, b);
    auto &re = row_entries[a];
    for (size_t t = 0; t < re.size(); t += 2) if (re[t] == b) return re[t + 1];
    int id = (int)pl.s_a.size();
    re.push_back(b); re.push_back(id);
    pl.s_a.push_back(a); pl.s_b.push_back(b); pl.s_ppos.push_back(-1);
    return id;
  };
  for (int a = 0; a < pl.n_c; a++) entry_id(a, a);
  std::vector<std::vector<int>> sa, ss;   // per entry contributions, grown on demand
  auto grow = [&](int id) { if ((int)sa.size() <= id) { sa.resize(id + 1); ss.resize(id + 1); } };
  for (int j = 0; j < n; j++)
    for (int t = Pp[j]; t < Pp[j + 1]; t++) {
      int a = pl.core_of[Pi[t]], b = pl.core_of[j];
      if (a < 0 || b < 0) continue;   // diagonal P entry of an eliminated variable
      int id = entry_id(a, b); pl.s_ppos[id] = t;
    }
  for (int i = 0; i < m; i++)
    for (int s1 = pl.Rp[i]; s1 < pl.Rp[i + 1]; s1++) {
      int a = pl.core_of[pl.Rj[s1]];
      if (a < 0) continue;
      for (int s2 = pl.Rp[i]; s2 <= s1; s2++) {
        int b = pl.core_of[pl.Rj[s2]];
        if (b < 0) continue;
        int id = entry_id(a, b); grow(id);
        // (a, b) may come out swapped by entry_id; the product is symmetric
        sa[id].push_back(i); sa[id].push_back(pl.Rpos[s1]); sa[id].push_back(pl.Rpos[s2]);
      }
    }
  for (int ei = 0; ei < pl.n_e; ei++)
    for (int k1 = pl.e_ptr[ei]; k1 < pl.e_ptr[ei + 1]; k1++)
      for (int k2 = pl.e_ptr[ei]; k2 <= k1; k2++) {
        int id = entry_id(pl.pair_core[k1], pl.pair_core[k2]); grow(id);
        ss[id].push_back(k1); ss[id].push_back(k2); ss[id].push_back(ei);
      }
  pl.nS = (int)pl.s_a.size();
  grow(pl.nS - 1 < 0 ? 0 : pl.nS - 1);
  pl.sa_ptr.assign(pl.nS + 1, 0); pl.ss_ptr.assign(pl.nS + 1, 0);
  for (int id = 0; id < pl.nS; id++) {
    pl.sa_ptr[id + 1] = pl.sa_ptr[id] + (int)sa[id].size() / 3;
    pl.ss_ptr[id + 1] = pl.ss_ptr[id] + (int)ss[id].size() / 3;
    for (size_t t = 0; t < sa[id].size(); t += 3) {
      pl.sa_row.push_back(sa[id][t]); pl.sa_pa.push_back(sa[id][t + 1]); pl.sa_pb.push_back(sa[id][t + 2]);
    }
    for (size_t t = 0; t < ss[id].size(); t += 3) {
      pl.ss_k1.push_back(ss[id][t]); pl.ss_k2.push_back(ss[id][t + 1]); pl.ss_e.push_back(ss[id][t + 2]);
    }
  }
  return 0;
}
