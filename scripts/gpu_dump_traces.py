"""Per-problem ADMM iteration counts of every QP (parity mode, 7x20) -> gpurun_out/qp_iters_B.npy"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import arm_family as af
from sco_py_amd import batch as sb
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
arrays, _ = af.make_batch(B)
res = sb.solve_batch(arrays)
out = np.zeros((B, 6), dtype=np.int64)
for b, t in enumerate(res.trace):
    out[b, :len(t)] = t[:, 7]
os.makedirs("gpurun_out", exist_ok=True)
np.save("gpurun_out/qp_iters_%d.npy" % B, out)
print(out[:4], res.timing)
