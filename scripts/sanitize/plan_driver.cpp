// Host driver over csrc/qp_plan.cpp for the CPU sanitizer job (scripts/cpu_sanitize.sh): the two debug entry points
// tests/test_qp_plan.py uses, without HIP.  Built with g++ -fsanitize=address,undefined; never part of the product.
#include <algorithm>
#include <cstring>
#include <vector>

#include "../../sco_py_amd/csrc/qp_plan.h"

static QpPlan g_plan;
extern "C" int sco_debug_plan_build(int n, int m, const int *Pp, const int *Pi, const int *Ap, const int *Ai, int allow_elim, int *sizes) {
  const int rc = qp_plan_build(n, m, Pp, Pi, Ap, Ai, allow_elim, g_plan);
  if (rc) return -1;
  const QpPlan &p = g_plan;
  sizes[0] = p.n_e; sizes[1] = p.n_c; sizes[2] = p.ncpl; sizes[3] = p.nS;
  sizes[4] = (int)p.cp_row.size(); sizes[5] = (int)p.sa_row.size(); sizes[6] = (int)p.ss_k1.size(); sizes[7] = (int)p.Fi.size();
  return 0;
}
extern "C" int sco_debug_plan_get(const char *name, int *out, int cap) {
  const QpPlan &p = g_plan;
  const std::vector<int> *v = nullptr;
#define F(x) if (!strcmp(name, #x)) v = &p.x;
  F(Rp) F(Rj) F(Rpos) F(Fp) F(Fi) F(Fpos) F(Pdiag) F(elim_var) F(core_var) F(elim_of) F(core_of)
  F(e_ptr) F(pair_core) F(pair_elim) F(cp_ptr) F(cp_row) F(cp_pa) F(cp_pe) F(a_ptr) F(a_pair)
  F(s_a) F(s_b) F(s_ppos) F(sa_ptr) F(sa_row) F(sa_pa) F(sa_pb) F(ss_ptr) F(ss_k1) F(ss_k2) F(ss_e)
#undef F
  if (!v) return -1;
  if ((int)v->size() > cap) return -4;
  std::copy(v->begin(), v->end(), out);
  return (int)v->size();
}
