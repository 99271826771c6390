"""Golden vectors for NON-QUADRATIC objective terms (Prob.add_obj_expr on a plain Expr: numeric gradient and Hessian,
degree-2 convexification with the eigenvalue shift, expr.py:102-156, prob.py:88-104, 532-534) recorded from the
REFERENCE's own modules, with the same stand-ins as make_golden.py (see its header):
    python tests/golden/make_golden_obj.py   ->  tests/golden/trajopt_obj.npz

Problems: the planar-arm trajectory problem plus weight * ||ee(theta_t) - ee(goal)||^2 for every timestep
(sco_py_amd/workloads.py: ee_cost).  obj_cases.py lists them; two of them run with the reference's quirks, the rest
with a Solver whose penalty loop is allowed to escalate."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg                      # noqa: E402
from obj_cases import CASES                   # noqa: E402
from oracle import arm_family as af           # noqa: E402


def main():
    mg.install_standins()
    mods = mg.import_reference()
    out = {}
    for name, kw, solver_attrs in CASES:
        res = mg.run_trajopt(mods, af.make_problem(**kw), solver_attrs=solver_attrs)
        print(name, "success", res["success"], [(q["status"], q["iters"]) for q in res["qps"]])
        mg.pack(name + "_", res, out, sparse=kw.get("d", 7) > 3)
    np.savez_compressed(os.path.join(HERE, "trajopt_obj.npz"), **out)
    print("trajopt_obj.npz", os.path.getsize(os.path.join(HERE, "trajopt_obj.npz")), "bytes")


if __name__ == "__main__":
    main()
