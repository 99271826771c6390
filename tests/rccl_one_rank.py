"""Child process of tests/test_dist_gpu.py: the RCCL calls of the N > 1 path in a process group of ONE rank.

A one-GPU box cannot hold two RCCL ranks (RCCL refuses two ranks on one device), but a communicator of one rank is a real
RCCL communicator: init_process_group(backend="nccl", device_id=...) exactly as bench.py does it, the all-gather of the
24-byte result records on device tensors (sco_py_amd/dist.py: gather_results, forced past its one-rank shortcut), the MAX
reduction of the clock (max_over_ranks) and the barriers.  The records come from a real solve of a small shard through
libsco_hip.so.  Prints one JSON line.  Run as a fresh process (it initialises the GPU)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    port = int(sys.argv[1]) if len(sys.argv) > 1 else 29533
    import torch
    import torch.distributed as dist
    from sco_py_amd import batch as sb, dist as sd
    from sco_py_amd import workloads as af
    torch.cuda.set_device(0)
    dist.init_process_group(backend="nccl", init_method="tcp://127.0.0.1:%d" % port, world_size=1, rank=0,
                            device_id=torch.device("cuda", 0))
    dims = dict(d=7, T=20, K=5, O=2)
    B = 24
    arrays, _ = af.make_batch(B, first=0, **dims)
    with sb.TrajOptBatch(B, dims["d"], dims["T"], dims["K"], dims["O"], device=0) as tb:
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])
        dist.barrier()
        tb.solve()
        res = tb.fetch(with_merit=True)
    rec = sd.pack_results(res.merit, res.max_violation, res.success, res.sqp_iters)
    got = sd.gather_results(rec, B, force_collective=True)
    clock = sd.max_over_ranks(1.25)
    dist.barrier()
    out = {"backend": dist.get_backend(), "device": sd.collective_device(), "world": dist.get_world_size(),
           "identical": bool(got.tobytes() == rec.tobytes()), "clock": clock, "sqp_iters": int(got["sqp_iters"].sum()),
           "success": int(np.count_nonzero(got["success"]))}
    dist.destroy_process_group()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
