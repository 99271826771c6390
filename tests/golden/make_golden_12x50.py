"""BASELINE configs[4] (12-DOF x 50 timesteps, 5000 non-linear rows) run by the REFERENCE's own modules, with the
same stand-ins for the two absent third-party packages as make_golden.py (our ADMM at the `osqp` seam, our finite
differences for `numdifftools`).  Takes several minutes and a few GB (the reference assembles a dense 10 624 x 5600
A, osqp_utils.py:146-193), so the result is committed:

    python tests/golden/make_golden_12x50.py      ->  tests/golden/trajopt_12x50.npz

Recorded: the result (success flag, trajectory, max violation, merit-call log, status and iteration count of every
QP) and the assembled (P, q, A, l, u) of the projection QP and of the first penalty QP in sparse form."""
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from oracle import arm_family as af           # noqa: E402

if __name__ == "__main__":
    mg.install_standins()
    mods = mg.import_reference()
    t0 = time.time()
    res = mg.run_trajopt(mods, af.make_problem(0, d=12, T=50, K=10, O=10))
    print("reference run: success", res["success"], "QPs", [(q["status"], q["iters"]) for q in res["qps"]],
          "%.0f s" % (time.time() - t0), flush=True)
    full = res["qps"]
    out = {}
    res["qps"] = full[:2]                      # (P, q, A, l, u, x) of the projection QP and the first penalty QP
    mg.pack("p0_", res, out, sparse=True)
    out["p0_n_qp_total"] = np.array(len(full))
    out["p0_qp_status"] = np.array([q["status"] for q in full]); out["p0_qp_iters"] = np.array([q["iters"] for q in full])
    np.savez_compressed(os.path.join(HERE, "trajopt_12x50.npz"), **out)
    print("trajopt_12x50.npz", os.path.getsize(os.path.join(HERE, "trajopt_12x50.npz")), "bytes")
