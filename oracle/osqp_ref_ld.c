/*
 * oracle/osqp_ref_ld.c -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * The SAME restatement as oracle/osqp_ref.c (the file is included below, not repeated), compiled with every `double` of
 * it turned into x87 `long double` (64-bit mantissa, unit round-off 5.4e-20 against 1.1e-16): Ruiz scaling, the KKT
 * LDL' (or the reduced Cholesky), every ADMM iterate and every termination test run with 2048 x less rounding noise, on
 * the same constants (the literals stay double literals).  Behind the double ABI of osqp_ref_solve, so oracle/osqp_ref.py
 * calls either build with one set of ctypes signatures (solve(..., extended=True)).
 *
 * What it is for: OSQP's iterate sequence is "parity unpinned" here (library absent, osqp_ref.c
 * header), and on QPs that creep along their tolerance for thousands of iterations the double KKT route of the oracle
 * and the reduced-system route of the device can pass the termination test at different checks.  This build says where
 * the ALGORITHM passes it when rounding is taken out of the picture, i.e. which of the two double routes the
 * extended-precision trajectory sides with (tests/test_adjudicate.py, profiles/r04_adjudication.txt).
 * Reference call site: /root/reference/sco_py/sco_osqp/osqp_utils.py:195-216.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <tgmath.h>

typedef struct {
  double rho, sigma, alpha, eps_abs, eps_rel, eps_prim_inf, eps_dual_inf;
  int max_iter, check_termination, scaling, expand_dups, linsys, adaptive_rho, adaptive_rho_interval;
  double adaptive_rho_tolerance;
} settings_abi;
typedef struct { int status, iters; double obj, pri_res, dua_res, rho; int rho_updates; } info_abi;

#define double long double
#define osqp_ref_solve osqp_ref_ld_solve_impl
#define osqp_ref_default_settings osqp_ref_ld_default_settings_impl
#include "osqp_ref.c"
#undef osqp_ref_default_settings
#undef osqp_ref_solve
#undef double

typedef long double ld;

void osqp_ref_default_settings(settings_abi *s) {
  osqp_ref_settings t; osqp_ref_ld_default_settings_impl(&t);
  s->rho = (double)t.rho; s->sigma = (double)t.sigma; s->alpha = (double)t.alpha; s->eps_abs = (double)t.eps_abs; s->eps_rel = (double)t.eps_rel;
  s->eps_prim_inf = (double)t.eps_prim_inf; s->eps_dual_inf = (double)t.eps_dual_inf;
  s->max_iter = t.max_iter; s->check_termination = t.check_termination; s->scaling = t.scaling; s->expand_dups = t.expand_dups;
  s->linsys = t.linsys; s->adaptive_rho = t.adaptive_rho; s->adaptive_rho_interval = t.adaptive_rho_interval;
  s->adaptive_rho_tolerance = (double)t.adaptive_rho_tolerance;
}

static ld *widen(const double *a, int k) {
  ld *r = (ld *)malloc(sizeof(ld) * (k > 0 ? k : 1));
  for (int i = 0; i < k; i++) r[i] = (ld)a[i];
  return r;
}

int osqp_ref_solve(int n, int m, const int *Pp, const int *Pi, const double *Px, const double *q, const int *Ap, const int *Ai,
                   const double *Ax, const double *l, const double *u, const int *w, const settings_abi *s, double *x_out,
                   double *y_out, info_abi *info, double *trace, int trace_cap, int *trace_len) {
  osqp_ref_settings t;
  t.rho = s->rho; t.sigma = s->sigma; t.alpha = s->alpha; t.eps_abs = s->eps_abs; t.eps_rel = s->eps_rel;
  t.eps_prim_inf = s->eps_prim_inf; t.eps_dual_inf = s->eps_dual_inf;
  t.max_iter = s->max_iter; t.check_termination = s->check_termination; t.scaling = s->scaling; t.expand_dups = s->expand_dups;
  t.linsys = s->linsys; t.adaptive_rho = s->adaptive_rho; t.adaptive_rho_interval = s->adaptive_rho_interval;
  t.adaptive_rho_tolerance = s->adaptive_rho_tolerance;
  ld *Pxl = widen(Px, Pp[n]), *ql = widen(q, n), *Axl = widen(Ax, Ap[n]), *ll = widen(l, m), *ul = widen(u, m);
  ld *xl = (ld *)calloc(n > 0 ? n : 1, sizeof(ld)), *yl = (ld *)calloc(m > 0 ? m : 1, sizeof(ld));
  ld *tl = (ld *)calloc(4 * (size_t)(trace_cap > 0 ? trace_cap : 1), sizeof(ld));
  osqp_ref_info inf; memset(&inf, 0, sizeof inf);
  int len = 0;
  int rc = osqp_ref_ld_solve_impl(n, m, Pp, Pi, Pxl, ql, Ap, Ai, Axl, ll, ul, w, &t, xl, yl, &inf, trace ? tl : NULL, trace_cap, &len);
  for (int i = 0; i < n; i++) x_out[i] = (double)xl[i];
  for (int i = 0; i < m; i++) y_out[i] = (double)yl[i];
  info->status = inf.status; info->iters = inf.iters; info->obj = (double)inf.obj; info->pri_res = (double)inf.pri_res;
  info->dua_res = (double)inf.dua_res; info->rho = (double)inf.rho; info->rho_updates = inf.rho_updates;
  if (trace) for (int i = 0; i < 4 * len; i++) trace[i] = (double)tl[i];
  if (trace_len) *trace_len = len;
  free(Pxl); free(ql); free(Axl); free(ll); free(ul); free(xl); free(yl); free(tl);
  return rc;
}
