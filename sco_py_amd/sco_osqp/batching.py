"""Run many ``Solver.solve`` calls concurrently and batch their QP solves on the GPU.

The reference solves candidate plans one ``Prob`` after another
(/root/reference/sco_py/sco_osqp/solver.py:30); every QP is a separate OSQP call
(osqp_utils.py:195-216).  ``solve_many`` keeps that per-problem algorithm and the whole
object API -- including arbitrary Python callables for ``f / grad / hess``, which are
evaluated on the host exactly as before -- but runs the solves in threads and parks each
one at its QP seam until every running solve has arrived there; QPs that share a
sparsity pattern and settings then go to the device as ONE batched ``sco_qp_solve``.

This is the "host-callback fallback" of SURVEY.md 8(f): it keeps the API for constraint
functions the GPU cannot evaluate, and removes the QP-solve cost, not the Python cost.
Problems whose constraints belong to a device family should use ``batch.TrajOptBatch``.
"""
import threading

import numpy as np

_local = threading.local()


def current_server():
    return getattr(_local, "server", None)


def _pattern_key(req):
    P, A = req["P"], req["A"]
    return (A.shape, P.indptr.tobytes(), P.indices.tobytes(), A.indptr.tobytes(), A.indices.tobytes(),
            tuple(float(v) for v in req["settings"]))


class QPBatchServer(object):
    """Rendezvous point: `submit` blocks until all active workers are parked, then the
    last arrival solves every pending QP (grouped by pattern) and wakes the others."""

    def __init__(self, n_workers):
        self._cv = threading.Condition()
        self._active = n_workers
        self._pending = []          # (request, slot)
        self.launches = 0
        self.qps = 0

    def _flush_locked(self):
        from . import osqp_utils
        groups = {}
        for req, slot in self._pending:
            groups.setdefault(_pattern_key(req), []).append((req, slot))
        self._pending = []
        for items in groups.values():
            try:
                results = osqp_utils._solve_qp_batch([r for r, _ in items])
                for (_, slot), res in zip(items, results):
                    slot["result"] = res
            except Exception as exc:      # hand the failure to every waiting solve
                for _, slot in items:
                    slot["error"] = exc
            self.launches += 1
            self.qps += len(items)
        self._cv.notify_all()

    def submit(self, req):
        slot = {}
        with self._cv:
            self._pending.append((req, slot))
            if len(self._pending) >= self._active:
                self._flush_locked()
            while "result" not in slot and "error" not in slot:
                self._cv.wait()
        if "error" in slot:
            raise slot["error"]
        return slot["result"]

    def worker_done(self):
        with self._cv:
            self._active -= 1
            if self._pending and len(self._pending) >= self._active:
                self._flush_locked()


def solve_many(probs, solver_factory=None, method="penalty_sqp", **solve_kwargs):
    """``[Solver().solve(p, method=..., **solve_kwargs) for p in probs]`` for many problems at once.  Returns
    (list of return values, stats).

    Problems the device can evaluate itself (``compile.compile_prob``: every non-linear expression a
    ``devexpr.DeviceExpr``) are bucketed by structure and each bucket runs as ONE batch of the device-resident loop
    (``sco_sqp_*``): no Python inside the solve.  The rest keep the per-problem host loop in threads with their QP solves
    batched on the GPU (below).  stats: ``device_problems`` / ``device_batches`` / ``compile_s`` / ``device_solve_s`` /
    ``write_back_s`` for the first kind, ``device_launches`` / ``qps`` for the second."""
    import time
    from .solver import Solver
    from . import compile as sco_compile
    factory = solver_factory or Solver
    if method != "penalty_sqp":
        raise Exception("This method is not supported.")
    out = [None] * len(probs)
    stats = dict(device_problems=0, device_batches=0, compile_s=0.0, device_solve_s=0.0, device_launches=0, qps=0)
    lead = factory()
    if solve_kwargs.get("tol") is not None:                 # Q8, as Solver.solve does it
        lead.min_trust_region_size = lead.min_approx_improve = lead.cnt_tolerance = solve_kwargs["tol"]
    host = list(range(len(probs)))
    if lead.device_loop and lead._runs_reference_control_flow():
        t0 = time.perf_counter()
        buckets, host = {}, []
        for k, p in enumerate(probs):
            cp = sco_compile.compile_prob(p)
            if cp is None:
                host.append(k)
            else:
                buckets.setdefault(cp.key, []).append((k, cp))
        stats["compile_s"] = time.perf_counter() - t0
        kw = {k: v for k, v in solve_kwargs.items() if k not in ("tol", "verbose")}
        for items in buckets.values():
            t0 = time.perf_counter()
            res = lead._solve_compiled([probs[k] for k, _ in items], [c for _, c in items], **kw)
            stats["device_solve_s"] += time.perf_counter() - t0
            for (k, _), r in zip(items, res):
                out[k] = r
            stats["device_problems"] += len(items); stats["device_batches"] += 1
        stats["last_device"] = lead.last_device
    if not host:
        return out, stats
    server = QPBatchServer(len(host))
    errors = [None] * len(probs)

    def run(k):
        _local.server = server
        try:
            out[k] = factory().solve(probs[k], method=method, **solve_kwargs)
        except Exception as exc:
            errors[k] = exc
        finally:
            _local.server = None
            server.worker_done()

    threads = [threading.Thread(target=run, args=(k,), daemon=True) for k in host]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for e in errors:
        if e is not None:
            raise e
    stats["device_launches"] = server.launches; stats["qps"] = server.qps
    return out, stats
