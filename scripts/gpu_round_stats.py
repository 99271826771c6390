"""How many problems are still active in every lock-step round (parity mode, 7x20, B = 1024)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
arrays, _ = af.make_batch(B)
res = sb.solve_batch(arrays)
q = res.qp_solves
print("timing", res.timing)
print("qp_solves histogram (incl. projection):", {int(k): int(v) for k, v in zip(*np.unique(q, return_counts=True))})
for r in range(1, int(q.max())):
    its = np.array([t[r, 7] for t in res.trace if len(t) > r])
    print("round %d: %d problems, ADMM iterations min %d mean %.0f max %d" % (r, len(its), its.min(), its.mean(), its.max()))
