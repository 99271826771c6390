"""Penalty-SQP driver of the SCO front-end.

Mirror of ``sco_py.sco_osqp.solver`` (/root/reference/sco_py/sco_osqp/solver.py):
same attributes, same ``solve`` signature and return value, same accept / shrink /
expand / converge decisions in the same order.  Each QP is solved on the GPU via
``Prob.optimize``; the fully device-resident batch loop lives in
:mod:`sco_py_amd.batch` (``sco_sqp_*`` in include/sco_hip.h).

``Solver.trace`` (not in the reference) records one tuple per trust-region trial so
that runs can be compared stage by stage.
"""
import time

import numpy as np

from . import osqp_utils

# decision codes stored in the trace (shared with the C ABI, include/sco_hip.h)
STEP_PROJECT, STEP_ACCEPT, STEP_SHRINK, STEP_YCONV, STEP_XCONV, STEP_BAD, STEP_GROUP = range(7)


class Solver(object):
    """SCO solver; defaults from Pieter Abbeel's CS287 penalty_sqp.m (solver.py:13-28)."""

    def __init__(self):
        self.improve_ratio_threshold = 0.25
        self.min_trust_region_size = 1e-4
        self.min_approx_improve = 1e-8
        self.max_iter = 50                      # never read, as in the reference (Q5)
        self.trust_shrink_ratio = 0.1
        self.trust_expand_ratio = 1.5
        self.cnt_tolerance = 1e-4
        self.max_merit_coeff_increases = 1
        self.merit_coeff_increase_ratio = 1e1
        self.initial_trust_region_size = 1
        self.initial_penalty_coeff = 1e3
        self.trace = []

    def solve(self,
              prob,
              method=None,
              tol=None,
              verbose=False,
              osqp_eps_abs=osqp_utils.DEFAULT_EPS_ABS,
              osqp_eps_rel=osqp_utils.DEFAULT_EPS_REL,
              osqp_max_iter=osqp_utils.DEFAULT_MAX_ITER,
              rho: float = osqp_utils.DEFAULT_RHO,
              adaptive_rho: bool = osqp_utils.DEFAULT_ADAPTIVE_RHO,
              sigma: float = osqp_utils.DEFAULT_SIGMA,
              ):
        """Returns whether the solve succeeded (solver.py:30-59).  ``tol``
        permanently overwrites three thresholds of this instance (Q8)."""
        if tol is not None:
            self.min_trust_region_size = tol
            self.min_approx_improve = tol
            self.cnt_tolerance = tol
        if method != "penalty_sqp":
            raise Exception("This method is not supported.")
        qp_kw = dict(osqp_eps_abs=osqp_eps_abs, osqp_eps_rel=osqp_eps_rel,
                     osqp_max_iter=osqp_max_iter, rho=rho, adaptive_rho=adaptive_rho, sigma=sigma)
        return self._penalty_sqp(prob, verbose=verbose, **qp_kw)

    # @profile
    def _penalty_sqp(self, prob, verbose=False, **qp_kw):
        """Outer loop: project onto the linear constraints, then minimise the
        merit function, raising the penalty coefficient while the constraints stay
        violated (solver.py:62-105)."""
        t0 = time.time()
        self.trace = []
        trust = self.initial_trust_region_size
        penalty = self.initial_penalty_coeff

        # the projection QP runs with DEFAULT solver settings (Q7, solver.py:81)
        if not prob.find_closest_feasible_point():
            return False
        self.trace.append((STEP_PROJECT, 0.0, 0.0, 0.0, trust, penalty))

        for _ in range(self.max_merit_coeff_increases):
            success = self._min_merit_fn(prob, penalty, trust, verbose=verbose, **qp_kw)
            if verbose:
                print("\n")
            if prob.get_max_cnt_violation() > self.cnt_tolerance:
                penalty = penalty * self.merit_coeff_increase_ratio
                trust = self.initial_trust_region_size
            else:
                if verbose:
                    print("sqp time: ", time.time() - t0)
                return success
        if verbose:
            print("sqp time: ", time.time() - t0)
        return False

    # @profile
    def _min_merit_fn(self, prob, penalty_coeff, trust_region_size, verbose=False, **qp_kw):
        """Trust-region minimisation of the l1 merit function for a fixed penalty
        coefficient (solver.py:108-253)."""
        sqp_iter = 1
        while True:
            if verbose:
                print("  sqp_iter: {0}".format(sqp_iter))
            prob.convexify()
            prob.update_obj(penalty_coeff)
            merit = prob.get_value(penalty_coeff)
            merit_vec = prob.get_value(penalty_coeff, True)
            prob.save()

            while True:
                if verbose:
                    print("    trust region size: {0}".format(trust_region_size))
                prob.add_trust_region(trust_region_size)
                _ = prob.optimize(verbose=verbose, **qp_kw)     # result ignored (Q6)
                model_merit = prob.get_approx_value(penalty_coeff)
                model_merit_vec = prob.get_approx_value(penalty_coeff, True)
                new_merit = prob.get_value(penalty_coeff)

                approx_improve = merit - model_merit
                if not approx_improve:
                    approx_improve += 1e-12
                exact_improve = merit - new_merit
                ratio = exact_improve / approx_improve

                # per-group bookkeeping; with no groups the scalar takes their place
                approx_improve_vec = merit_vec - model_merit_vec
                violated = merit_vec > self.cnt_tolerance
                if approx_improve_vec.shape == (0,):
                    approx_improve_vec = np.array([approx_improve])
                    violated = approx_improve_vec > -np.inf

                if verbose:
                    print("      merit: {0}. model_merit: {1}. new_merit: {2}".format(
                        merit, model_merit, new_merit))
                    print("      approx_merit_improve: {0}. exact_merit_improve: {1}. "
                          "merit_improve_ratio: {2}".format(approx_improve, exact_improve, ratio))

                rec = (merit, model_merit, new_merit, trust_region_size, penalty_coeff)
                if self._bad_model(approx_improve):
                    if verbose:
                        print("Approximate merit function got worse ({0})".format(approx_improve))
                        print("Either convexification is wrong to zeroth order, or you're in "
                              "numerical trouble.")
                    prob.restore()
                    self.trace.append((STEP_BAD,) + rec)
                    return False

                if self._y_converged(approx_improve):
                    if verbose:
                        print("Converged: y tolerance")
                    prob.restore()
                    self.trace.append((STEP_YCONV,) + rec)
                    return True

                # a violated group whose model predicts no progress, and none of
                # whose overlapping groups is progressing, ends the run
                prob.nonconverged_groups = []
                for gid, idx in prob.gid2ind.items():
                    if violated[idx] and approx_improve_vec[idx] < self.min_approx_improve:
                        if any(approx_improve_vec[prob.gid2ind[g2]] > self.min_approx_improve
                               for g2 in prob._cnt_groups_overlap[gid]):
                            continue
                        prob.nonconverged_groups.append(gid)
                if len(prob.nonconverged_groups) > 0:
                    if verbose:
                        print("Converged: y tolerance")
                    prob.restore()
                    for i, g in enumerate(sorted(prob._cnt_groups.keys())):
                        if violated[i] and self._y_converged(approx_improve_vec[i]):
                            prob.nonconverged_groups.append(g)
                    self.trace.append((STEP_GROUP,) + rec)
                    return True

                if self._shrink_trust_region(exact_improve, ratio):
                    prob.restore()
                    if verbose:
                        print("Shrinking trust region")
                    self.trace.append((STEP_SHRINK,) + rec)
                    trust_region_size = trust_region_size * self.trust_shrink_ratio
                else:
                    if verbose:
                        print("Growing trust region")
                    self.trace.append((STEP_ACCEPT,) + rec)
                    trust_region_size = trust_region_size * self.trust_expand_ratio
                    break

                if self._x_converged(trust_region_size):
                    if verbose:
                        print("Converged: x tolerance")
                    self.trace[-1] = (STEP_XCONV,) + rec      # one trace row per QP solve
                    return True

            sqp_iter = sqp_iter + 1

    def _bad_model(self, approx_merit_improve):
        return approx_merit_improve < -1e-5                 # solver.py:261

    def _shrink_trust_region(self, exact_merit_improve, merit_improve_ratio):
        return (exact_merit_improve < 0) or (merit_improve_ratio < self.improve_ratio_threshold)

    def _x_converged(self, trust_region_size):
        return trust_region_size < self.min_trust_region_size

    def _y_converged(self, approx_merit_improve):
        return approx_merit_improve < self.min_approx_improve
