"""Parity sweep: GPU device loop vs oracle on problems [first, first+N) of the 7x20 workload (parity mode)."""
import sys, os, time
import numpy as np
from concurrent.futures import ProcessPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af, sco_ref as sr

MODE = sys.argv[3] if len(sys.argv) > 3 else "parity"          # parity | intended (quirks off, at most 20 QPs)
KW = {}
for tok in sys.argv[4:]:                                         # reach, vel, groups, jl
    if tok == "reach": KW["reach"] = True
    if tok == "vel": KW["vel_limit"] = 0.4
    if tok == "groups": KW["groups"] = "split"
    if tok == "jl": KW["joint_limit"] = 0.2
    if tok == "point": KW.update(point=True, d=2, T=20, K=1, O=3)
    if tok == "quad": KW.update(quadratic=True, d=3, T=12, K=1, O=4)
    if tok == "prog": KW.update(program=True, d=2, T=12, K=1)                     # corridor rows (r02 form)
    if tok.startswith("prog:"):                                                    # prog:sweep | prog:dynamics | prog:curve | prog:attract
        v = tok.split(":")[1]
        KW.update(program=True, variant=v, d=3 if v in ("dynamics", "curve") else 2, T=10, K=1)
    if tok == "objw": KW["obj_weights"] = True                                     # r04: weighted smoothing objective
    if tok == "steps": KW["per_step"] = True                                       # r04: program parameters per timestep
    if tok == "circles": KW["circles"] = 2                                         # r04: two kinds of non-linear rows (with prog)
    if tok == "acc": KW["acc_weights"] = True                                      # r04: acceleration term in the quadratic objective
    if tok == "rows": KW["lin_rows"] = True                                        # r04: general affine rows
    if tok == "ajac": AJ = True
AJ = "ajac" in sys.argv[4:]


def ref_one(i):
    params = sr.SolverParams(compound_penalty=False, duplicate_rows=False, max_qp_solves=20) if MODE == "intended" else None
    out = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(i, **KW), analytic_jac=AJ), params, emulate_memo=True)
    return i, out.trace, out.x, out.success

if __name__ == "__main__":
    first, N = int(sys.argv[1]), int(sys.argv[2])
    with ProcessPoolExecutor(int(os.environ.get("SWEEP_PROCS", "16"))) as ex:                 # oracle on the host cores first (no GPU touched yet)
        refs = list(ex.map(ref_one, range(first, first + N), chunksize=4))
    from sco_py_amd import batch as sb, _lib
    arrays, _ = af.make_batch(N, first=first, **KW)
    p = _lib.default_sqp_params(compound_penalty=0, duplicate_rows=0, max_sqp_iters=20) if MODE == "intended" else None
    res = sb.solve_batch(arrays, params=p, analytic_jac=AJ)
    bad = 0; worst = 0.0
    for k, (i, tr, x, ok) in enumerate(refs):
        g = res.trace[k]
        same = g.shape == tr[:64].shape and np.array_equal(g[:, 0], tr[:64, 0]) and np.array_equal(g[:, 6:8], tr[:64, 6:8])
        dx = float(np.abs(res.x[k] - x).max())
        worst = max(worst, dx)
        if not same or dx > 1e-6 or bool(res.success[k]) != ok:
            bad += 1
            print("MISMATCH problem", i, "gpu", g[:, [0, 6, 7]].astype(int).tolist(), "oracle", tr[:, [0, 6, 7]].astype(int).tolist(), "dx %.2e" % dx)
    print("%s %s problems %d..%d: %d mismatches, worst |dx| %.2e" % (MODE, sorted(KW), first, first + N - 1, bad, worst))
