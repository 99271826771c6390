"""Fixture of the two QPs behind r03's 2 / 512 parity misses -- run on the CPU, commit the .npz.

    python tests/golden/make_adjudicate.py        ->  tests/golden/adjudicate_r03.npz   (deterministic: seeded problems)

prog:dynamics with forward-mode Jacobians, problem 43 (its 13th QP) and problem 57 (its 14th): the oracle (oracle/sco_ref.py
over oracle/osqp_ref.c) ends them after 7325 and 66 225 ADMM iterations, the r03 device after 6250 and 64 700; every other
number of both traces agrees.  Stored per QP: P (upper), q, A, l, u, row multiplicities w, and the oracle's iterate (x, y) with
its residuals at the checks on either side of the disagreement, from FOUR CPU routes that share no linear algebra: float64
KKT LDL', float64 reduced Cholesky, and both again in x87 extended precision (oracle/osqp_ref_ld.c).
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import arm_family as af, sco_ref as sr, osqp_ref as o

KW = dict(program=True, variant="dynamics", d=3, T=10, K=1)
CASES = ((43, 12, 6250, 7325), (57, 13, 64700, 66225))          # problem, QP index, device's count in r03, oracle's count
ROUTES = (("f64_kkt", {}), ("f64_reduced", dict(linsys=1)), ("x87_kkt", dict(extended=True)), ("x87_reduced", dict(extended=True, linsys=1)))

if __name__ == "__main__":
    out = {}
    for i, k, it_dev, it_orc in CASES:
        res = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(i, **KW), analytic_jac=True), None, emulate_memo=True, record_qps=True)
        q = res.qps[k]
        assert q["iters"] == it_orc and q["status"] == 1, (q["iters"], q["status"])
        tag = "p%d_" % i
        for key in ("P", "q", "A", "l", "u", "w"):
            out[tag + key] = np.triu(q[key]) if key == "P" else np.asarray(q[key])
        out[tag + "counts"] = np.array([it_dev, it_orc])
        for name, kw in ROUTES:
            full = o.solve(q["P"], q["q"], q["A"], q["l"], q["u"], w=q["w"], **kw)
            rows = []
            for it in (it_dev - 25, it_dev, it_orc - 25, it_orc):
                r = o.solve(q["P"], q["q"], q["A"], q["l"], q["u"], w=q["w"], max_iter=it, **kw)
                rows.append([it, r.info.status_val, r.info.pri_res, r.info.dua_res])
                if it == it_dev:
                    out[tag + name + "_x"] = r.x; out[tag + name + "_y"] = r.y
            out[tag + name + "_checks"] = np.array(rows)
            out[tag + name + "_final"] = np.array([full.info.status_val, full.info.iter])
            print(tag, name, "ends at", full.info.iter, "checks", rows)
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "adjudicate_r03.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")
