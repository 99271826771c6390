"""Penalty-SQP driver of the SCO front-end (host side).

Same public surface as ``sco_py.sco_osqp.solver.Solver`` (/root/reference/sco_py/sco_osqp/solver.py:8-59):
the eleven tuning attributes with the reference's defaults (:17-28), ``solve(prob, method, tol, verbose,
osqp_eps_abs, osqp_eps_rel, osqp_max_iter, rho, adaptive_rho, sigma) -> bool``, the exception for an unknown
method (:59), ``False`` when the projection QP fails (:81-82), and ``prob.nonconverged_groups`` filled on a
group stall (:209-235).

The control flow is organised the way the device loop is (``sqp_post_kernel`` in csrc/sco_sqp.hip), not the
way the reference writes it: every trust-region trial produces one ``Trial`` record of numbers, one pure
function ``classify_trial`` turns the record into a decision code, and a flat state machine
(CONVEXIFY -> TRIAL -> ... -> DONE) acts on the code.  The order in which the decisions are tested is part of
the behaviour (it decides which exit a borderline step takes) and is the reference's:
bad model (:181) -> y-converged (:196) -> stalled group (:205-235) -> shrink (:237) / accept (:242), and the
x-convergence test only after a shrink (:248).

``Solver.trace`` (not in the reference) keeps one row per trial, (code, merit, model_merit, new_merit, trust,
penalty): the artefact the parity tests compare with the device loop and with the golden reference runs.
"""
import time
from collections import namedtuple

import numpy as np

from . import osqp_utils

# decision codes (shared with the C ABI, include/sco_hip.h, and with oracle/sco_ref.py)
STEP_PROJECT, STEP_ACCEPT, STEP_SHRINK, STEP_YCONV, STEP_XCONV, STEP_BAD, STEP_GROUP = range(7)

BAD_MODEL_THRESHOLD = -1e-5          # solver.py:261
ZERO_IMPROVE_NUDGE = 1e-12           # solver.py:151-152: an exactly zero model improvement is nudged off zero

Thresholds = namedtuple("Thresholds", "improve_ratio min_approx_improve cnt_tolerance")
Trial = namedtuple("Trial", "merit model_merit new_merit merit_vec model_vec")
Verdict = namedtuple("Verdict", "code approx_improve exact_improve ratio stalled reported")


def classify_trial(trial, thr, group_index, group_overlap, group_order, predicates=None):
    """Decision for one trust-region trial; a pure function of the trial's numbers.

    ``group_index``: gid -> position in the merit vectors (``prob.gid2ind``); ``group_overlap``: gid -> gids
    sharing a constraint (``prob._cnt_groups_overlap``); ``group_order``: the gids in vector order (sorted,
    prob.py:559).  Returns a ``Verdict``: ``stalled`` = the violated groups whose model predicts no progress and
    none of whose overlapping groups progresses (they end the run); ``reported`` = what the reference leaves in
    ``prob.nonconverged_groups`` then (the stalled ones followed by every violated group under the y threshold,
    solver.py:232-234).  STEP_SHRINK here never means x-converged: that depends on the trust size and is the
    caller's test.  ``predicates`` = (bad_model, y_converged, shrink_trust_region) callables: ``Solver`` hands in its
    own ``_bad_model`` / ``_y_converged`` / ``_shrink_trust_region`` (the reference's overridable predicate methods,
    solver.py:255-283), so a subclass that overrides them decides here exactly as it does in the reference; without
    them the thresholds of ``thr`` are used (the same tests).
    """
    approx = trial.merit - trial.model_merit
    if not approx:
        approx += ZERO_IMPROVE_NUDGE
    exact = trial.merit - trial.new_merit
    ratio = exact / approx

    def verdict(code, stalled=(), reported=()):
        return Verdict(code, approx, exact, ratio, list(stalled), list(reported))

    bad_model, y_converged, shrink = predicates or (
        lambda a: a < BAD_MODEL_THRESHOLD, lambda a: a < thr.min_approx_improve,
        lambda e, r: e < 0 or r < thr.improve_ratio)
    if bad_model(approx):
        return verdict(STEP_BAD)
    if y_converged(approx):
        return verdict(STEP_YCONV)

    per_group = np.asarray(trial.merit_vec, dtype=np.float64) - np.asarray(trial.model_vec, dtype=np.float64)
    if per_group.shape == (0,):              # no groups: the scalar stands in, and it already passed the y test
        per_group = np.array([approx])
        violated = np.array([True])
    else:
        violated = np.asarray(trial.merit_vec) > thr.cnt_tolerance
    stalled = []
    for gid, k in group_index.items():
        if not (violated[k] and per_group[k] < thr.min_approx_improve):
            continue
        if any(per_group[group_index[other]] > thr.min_approx_improve for other in group_overlap[gid]):
            continue                         # a group sharing a constraint with it is still making progress
        stalled.append(gid)
    if stalled:
        # the report uses the overridable predicate (solver.py:233), the stall test above the raw threshold (:213-221)
        under = [g for k, g in enumerate(group_order) if violated[k] and y_converged(per_group[k])]
        return verdict(STEP_GROUP, stalled, stalled + under)

    if shrink(exact, ratio):
        return verdict(STEP_SHRINK)
    return verdict(STEP_ACCEPT)


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
        # not in the reference: False keeps every solve on the host loop; last_path says where the last solve ran
        # ("device": the resident loop, "host"); last_device has the device loop's round count and stage times
        self.device_loop = True
        self.last_path = None
        self.last_device = None

    # ------------------------------------------------------------------ public entry
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
        """Whether the solve succeeded (solver.py:30-59).  ``tol`` permanently overwrites three thresholds of
        this instance (Q8), also when the method is then rejected."""
        if tol is not None:
            self.min_trust_region_size = self.min_approx_improve = self.cnt_tolerance = tol
        if method != "penalty_sqp":
            raise Exception("This method is not supported.")
        qp_kw = dict(osqp_eps_abs=osqp_eps_abs, osqp_eps_rel=osqp_eps_rel, osqp_max_iter=osqp_max_iter, rho=rho,
                     adaptive_rho=adaptive_rho, sigma=sigma)
        if self.device_loop and self._runs_reference_control_flow():
            # a Prob whose non-linear expressions are device expressions runs the WHOLE solve in the device-resident
            # loop (sco_sqp_*); anything else keeps the host loop below with one device QP per optimize
            from . import compile as sco_compile
            cp = sco_compile.compile_prob(prob)
            if cp is not None:
                return self._solve_compiled([prob], [cp], verbose=verbose, **qp_kw)[0]
        self.last_path = "host"
        return self._penalty_sqp(prob, verbose=verbose, osqp_eps_abs=osqp_eps_abs, osqp_eps_rel=osqp_eps_rel,
                                 osqp_max_iter=osqp_max_iter, rho=rho, adaptive_rho=adaptive_rho, sigma=sigma)

    # ------------------------------------------------------------------ device-resident loop
    def _runs_reference_control_flow(self):
        """The device loop implements the reference's decisions (csrc/sco_sqp.hip: sqp_post_kernel); a subclass that
        overrides a predicate or a stage keeps the host loop, where its override is called."""
        return all(getattr(type(self), name) is getattr(Solver, name) for name in (
            "_penalty_sqp", "_min_merit_fn", "_bad_model", "_shrink_trust_region", "_x_converged", "_y_converged"))

    def device_params(self):
        """``sco_sqp_params`` from this instance's attributes (one-to-one, include/sco_hip.h)."""
        from .. import _lib
        return _lib.default_sqp_params(
            improve_ratio_threshold=self.improve_ratio_threshold, min_trust_region_size=self.min_trust_region_size,
            min_approx_improve=self.min_approx_improve, trust_shrink_ratio=self.trust_shrink_ratio,
            trust_expand_ratio=self.trust_expand_ratio, cnt_tolerance=self.cnt_tolerance,
            merit_coeff_increase_ratio=self.merit_coeff_increase_ratio,
            initial_trust_region_size=self.initial_trust_region_size, initial_penalty_coeff=self.initial_penalty_coeff,
            max_merit_coeff_increases=int(self.max_merit_coeff_increases))

    def _solve_compiled(self, probs, cps, verbose=False, osqp_eps_abs=osqp_utils.DEFAULT_EPS_ABS,
                        osqp_eps_rel=osqp_utils.DEFAULT_EPS_REL, osqp_max_iter=osqp_utils.DEFAULT_MAX_ITER,
                        rho=osqp_utils.DEFAULT_RHO, adaptive_rho=osqp_utils.DEFAULT_ADAPTIVE_RHO,
                        sigma=osqp_utils.DEFAULT_SIGMA, device=0):
        """Compiled problems of ONE structure as one device batch (solver.py:30-253 per problem); results go back into
        the Prob / Variable objects.  Returns the list of ``solve`` return values."""
        from .. import _lib
        from . import compile as sco_compile
        qs = _lib.default_qp_settings(eps_abs=osqp_eps_abs, eps_rel=osqp_eps_rel, max_iter=int(osqp_max_iter), rho=rho,
                                      adaptive_rho=1 if adaptive_rho else 0, sigma=sigma)
        res = sco_compile.run_compiled(cps, self.device_params(), qs, device=device)
        out = [sco_compile.write_back(p, c, res, b) for b, (p, c) in enumerate(zip(probs, cps))]
        # same rows as the host loop's trace: (code, merit, model_merit, new_merit, trust, penalty)
        self.trace = [tuple([int(r[0])] + [float(v) for v in r[1:6]]) for r in res.trace[-1]]
        self.last_path = "device"
        self.last_device = dict(res.timing, batch=len(probs), sqp_iters=res.sqp_iters.copy(), qp_solves=res.qp_solves.copy(),
                                admm_iters=res.admm_iters.copy(), flags=res.flags.copy(), traces=res.trace)
        if verbose:
            print("device-resident penalty sqp: %d problem(s), %d rounds, %.3f ms" % (len(probs), res.timing["rounds"], res.timing["total_ms"]))
        return out

    # ------------------------------------------------------------------ outer loop
    def _penalty_sqp(self, prob, verbose=False, **qp_kw):
        """Project onto the linear constraints, then minimise the merit function once per penalty level,
        raising the penalty while the constraints stay violated (solver.py:62-105)."""
        say = _Narrator(verbose)
        clock = time.time()
        self.trace = []
        if not prob.find_closest_feasible_point():        # default QP settings on purpose (Q7, solver.py:81)
            return False
        penalty = self.initial_penalty_coeff
        self.trace.append((STEP_PROJECT, 0.0, 0.0, 0.0, self.initial_trust_region_size, penalty))
        outcome = False
        for _level in range(self.max_merit_coeff_increases):
            minimised = self._min_merit_fn(prob, penalty, self.initial_trust_region_size, verbose=verbose, **qp_kw)
            if prob.get_max_cnt_violation() <= self.cnt_tolerance:
                outcome = minimised
                break
            penalty *= self.merit_coeff_increase_ratio    # still infeasible: next level starts from the full trust box
        say("penalty sqp finished in %.3f s" % (time.time() - clock))
        return outcome

    # ------------------------------------------------------------------ inner loop
    def _min_merit_fn(self, prob, penalty_coeff, trust_region_size, verbose=False, **qp_kw):
        """Trust-region minimisation of the l1 merit function at a fixed penalty coefficient
        (solver.py:108-253) as a two-state machine: CONVEXIFY builds the model at the current point and saves
        it; TRIAL solves the boxed QP and lets ``classify_trial`` decide."""
        say = _Narrator(verbose)
        thr = Thresholds(self.improve_ratio_threshold, self.min_approx_improve, self.cnt_tolerance)
        trust = trust_region_size
        state, sqp_iter = "CONVEXIFY", 0
        base_merit = base_vec = None
        while True:
            if state == "CONVEXIFY":
                sqp_iter += 1
                prob.convexify()
                prob.update_obj(penalty_coeff)
                base_merit = prob.get_value(penalty_coeff)
                base_vec = prob.get_value(penalty_coeff, True)
                prob.save()
                say("sqp iteration %d: merit %r" % (sqp_iter, base_merit))
                state = "TRIAL"
                continue

            prob.add_trust_region(trust)
            prob.optimize(verbose=verbose, **qp_kw)        # a failed QP leaves the point where it was (Q6)
            trial = Trial(base_merit, prob.get_approx_value(penalty_coeff), prob.get_value(penalty_coeff),
                          base_vec, prob.get_approx_value(penalty_coeff, True))
            v = classify_trial(trial, thr, prob.gid2ind, prob._cnt_groups_overlap, sorted(prob._cnt_groups.keys()),
                               predicates=(self._bad_model, self._y_converged, self._shrink_trust_region))
            row = (trial.merit, trial.model_merit, trial.new_merit, trust, penalty_coeff)
            say("  trust %g: model %r, new %r, improvement model %.3e / exact %.3e (ratio %.3g) -> %s"
                % (trust, trial.model_merit, trial.new_merit, v.approx_improve, v.exact_improve, v.ratio,
                   _CODE_NAMES[v.code]))

            if v.code in (STEP_BAD, STEP_YCONV):
                prob.restore()
                self.trace.append((v.code,) + row)
                return v.code == STEP_YCONV
            # from here on the group report is fresh for this trial (solver.py:209)
            prob.nonconverged_groups = list(v.reported)
            if v.code == STEP_GROUP:
                prob.restore()
                self.trace.append((STEP_GROUP,) + row)
                return True
            if v.code == STEP_ACCEPT:
                self.trace.append((STEP_ACCEPT,) + row)
                trust *= self.trust_expand_ratio
                state = "CONVEXIFY"
                continue
            prob.restore()                                  # STEP_SHRINK
            trust *= self.trust_shrink_ratio
            if self._x_converged(trust):
                self.trace.append((STEP_XCONV,) + row)      # one trace row per QP solve
                return True
            self.trace.append((STEP_SHRINK,) + row)

    # ------------------------------------------------------------------ the reference's predicate names
    def _bad_model(self, approx_merit_improve):
        return approx_merit_improve < BAD_MODEL_THRESHOLD

    def _shrink_trust_region(self, exact_merit_improve, merit_improve_ratio):
        return (exact_merit_improve < 0) or (merit_improve_ratio < self.improve_ratio_threshold)

    def _x_converged(self, trust_region_size):
        return trust_region_size < self.min_trust_region_size

    def _y_converged(self, approx_merit_improve):
        return approx_merit_improve < self.min_approx_improve


_CODE_NAMES = {STEP_PROJECT: "project", STEP_ACCEPT: "accept, expand", STEP_SHRINK: "reject, shrink",
               STEP_YCONV: "converged (model improvement)", STEP_XCONV: "converged (trust region)",
               STEP_BAD: "model got worse: give up", STEP_GROUP: "converged (stalled constraint group)"}


class _Narrator(object):
    """verbose=True progress lines (wording is ours; the reference prints different text)."""

    def __init__(self, on):
        self.on = bool(on)

    def __call__(self, text):
        if self.on:
            print(text)
