class TrajOptBatchGroups(object):
    """The same batch split into `groups` independent TrajOptBatch handles, each with its own
    HIP stream, solved from concurrent host threads.  The device loop of one handle advances
    in lock-step rounds (one QP per active problem per round), so a round lasts as long as its
    slowest QP; with several handles in flight the rounds of different groups overlap and the
    CUs a thinning group leaves idle are used by the others.  Results are concatenated in
    problem order; the per-problem algorithm is unchanged."""

    def __init__(self, batch, dof, horizon, n_points, n_obstacles, groups=4, device=0, **kw):
        pass
        self.B, self.groups = int(batch), max(1, min(int(groups), int(batch)))
        self.spans = [_dist.shard_range(self.B, g, self.groups) for g in range(self.groups)]
        self.parts = [TrajOptBatch(hi - lo, dof, horizon, n_points, n_obstacles, device=device, **kw)
                      for lo, hi in self.spans]

    def close(self):
        for p in self.parts:
            p.close()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def load(self, x0, start, goal, link_len, point_link, point_frac, obstacles):
        for (lo, hi), p in zip(self.spans, self.parts):
            p.load(x0[lo:hi], start[lo:hi], goal[lo:hi], link_len[lo:hi], point_link, point_frac, obstacles[lo:hi])

    def solve(self, params=None, qp_settings=None):
        import threading
        errs = []

        def run(p):
            try:
                p.solve(params, qp_settings)      # ctypes releases the GIL for the duration of the call
            except Exception as exc:              # noqa: BLE001
                errs.append(exc)

        ts = [threading.Thread(target=run, args=(p,)) for p in self.parts]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        if errs:
            raise errs[0]

    def fetch(self, with_merit=True):
        rs = [p.fetch(with_merit) for p in self.parts]
        cat = lambda name: None if getattr(rs[0], name) is None else np.concatenate([getattr(r, name) for r in rs])
        return SimpleNamespace(**{k: cat(k) for k in ("x", "success", "sqp_iters", "qp_solves", "admm_iters",
                                                      "merit", "max_violation")})

    def trace(self, cap=64):
        out = []
        for p in self.parts:
            out.extend(p.trace(cap))
        return out

    def last_timing(self):
        ts = [p.last_timing() for p in self.parts]
        return {k: float(sum(t[k] for t in ts)) for k in ts[0]}
