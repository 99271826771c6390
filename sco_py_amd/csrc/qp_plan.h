// qp_plan.h -- host-side symbolic analysis of a QP sparsity pattern.
//
// The ADMM x-update of OSQP solves the quasi-definite KKT system with a sparse
// LDL' (third-party call behind /root/reference/sco_py/sco_osqp/osqp_utils.py:216).
// On the GPU we solve the mathematically identical reduced system
//     K x~ = sigma x - q + A'(R z - y),   K = P + sigma I + A' R A,  R = diag(w rho)
// in two levels, both chosen here, once per pattern:
//   * E  -- an independent set of K's graph among variables with no off-diagonal
//           P entry (in a penalty QP these are the slack variables t_i of the
//           hinge rows built at prob.py:251-278).  K_EE is diagonal, so E is
//           eliminated in closed form.
//   * C  -- the remaining "core" variables; the Schur complement
//           S = K_CC - K_CE K_EE^-1 K_EC is formed densely, factored and inverted
//           once per QP, and applied as a dense mat-vec every ADMM iteration.
// Everything below is index plans (shared by all problems of a batch); values are
// per problem and live on the device.
#pragma once
#include <vector>

struct QpPlan {
  int n = 0, m = 0, nnzP = 0, nnzA = 0;
  // input patterns (copied)
  std::vector<int> Pp, Pi, Ap, Ai;
  // CSR view of A: row i -> entries Rj[t] (column), Rpos[t] (position in the CSC value array)
  std::vector<int> Rp, Rj, Rpos;
  // full symmetric P by column: column j -> Fi[t] (row), Fpos[t] (position in the triu value array)
  std::vector<int> Fp, Fi, Fpos;
  std::vector<int> Pdiag;          // position of P_jj in the triu array or -1
  // partition
  int n_e = 0, n_c = 0;
  std::vector<int> elim_var, core_var;   // index -> variable
  std::vector<int> elim_of, core_of;     // variable -> index or -1
  // coupling block K_CE as (core a, elim e) pairs, sorted by e
  int ncpl = 0;
  std::vector<int> e_ptr;          // [n_e+1] pairs of elim e
  std::vector<int> pair_core;      // [ncpl]  core index of the pair
  std::vector<int> pair_elim;      // [ncpl]  elim index of the pair
  std::vector<int> cp_ptr;         // [ncpl+1] contributions of pair k
  std::vector<int> cp_row, cp_pa, cp_pe;   // row i, position of A_ia, position of A_ie
  std::vector<int> a_ptr;          // [n_c+1] pairs of core a
  std::vector<int> a_pair;         // pair indices grouped by core a
  // structurally non-zero entries of S (lower triangle incl. diagonal, a >= b)
  int nS = 0;
  std::vector<int> s_a, s_b, s_ppos;        // [nS]; s_ppos = position in P triu array or -1
  std::vector<int> sa_ptr, sa_row, sa_pa, sa_pb;   // A' R A contributions
  std::vector<int> ss_ptr, ss_k1, ss_k2, ss_e;     // Schur contributions (pair idx, pair idx, elim idx)
};

// Returns 0 on success, negative on a malformed pattern.  allow_elim = 0 forces
// E = {} (dense core of order n), used to cross-check the elimination path;
// allow_elim = 2 eliminates only variables that sit in at most two rows.
int qp_plan_build(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai,
                  int allow_elim, QpPlan &out);
