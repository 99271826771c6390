"""Multi-GPU sharding of a batch of independent SCO problems.

Each ``Prob`` of the reference is self-contained (all state hangs off ``self``,
/root/reference/sco_py/sco_osqp/prob.py:48-86) and a caller solves them one after
another, so a batch shards embarrassingly: one process per GPU, a contiguous
slice of the batch per rank, NO data-path collective.  The only exchange is the
final gather of per-problem results -- (merit f64, max_violation f64, success i32,
sqp_iters i32) = 24 bytes per problem -- done with one all-gather (RCCL over xGMI
when the backend is "nccl", gloo on CPU for tests).
"""
import numpy as np

RESULT_DTYPE = np.dtype([("merit", "<f8"), ("max_violation", "<f8"), ("success", "<i4"), ("sqp_iters", "<i4")])


def shard_range(total, rank, world):
    """Contiguous slice [lo, hi) of `total` problems owned by `rank`; the first
    `total % world` ranks take one extra problem."""
    if world <= 0 or not (0 <= rank < world) or total < 0:
        raise ValueError("bad shard request")
    base, extra = divmod(total, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def pack_results(merit, max_violation, success, sqp_iters):
    rec = np.zeros(len(merit), dtype=RESULT_DTYPE)
    rec["merit"] = merit; rec["max_violation"] = max_violation
    rec["success"] = np.asarray(success, dtype=np.int32); rec["sqp_iters"] = sqp_iters
    return rec


def collective_device():
    """Where the tensors of this module's collectives live: the current HIP device under RCCL ("nccl"), host memory under gloo."""
    import torch.distributed as dist
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def max_over_ranks(value):
    """MAX of a float over the ranks (the clock of a timed region, bench.py); `value` itself without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return float(value)
    t = torch.tensor([value], dtype=torch.float64, device=collective_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_rank_stats(values):
    """All-gather a short vector of floats per rank (bench.py: a rank's own clock, SCO iterations, ADMM time): returns a
    (world, len(values)) array on every rank -- what tells a slow shard from a slow collective when scaling falls short of
    linear (data-dependent iteration counts are the one risk).  Outside the timed region."""
    import torch
    import torch.distributed as dist
    v = np.asarray(values, dtype=np.float64)
    if not (dist.is_available() and dist.is_initialized()):
        return v[None, :]
    t = torch.tensor(v, dtype=torch.float64, device=collective_device())
    out = torch.empty(dist.get_world_size() * v.shape[0], dtype=torch.float64, device=t.device)      # flat: gloo wants it so
    dist.all_gather_into_tensor(out, t)
    return out.cpu().numpy().reshape(dist.get_world_size(), v.shape[0])


def gather_results(local, total, device=None, force_collective=False):
    """All-gather the per-problem result records of every rank.

    local: structured array (RESULT_DTYPE) of this rank's shard; total: batch size
    over all ranks.  Returns the (total,) structured array on every rank.  With no
    initialised process group (single process) it returns `local` unchanged; so it does in a
    group of one rank unless `force_collective` asks for the collective all the same (a one-GPU
    box runs the RCCL calls of the N > 1 path that way, tests/test_dist_gpu.py)."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size() == 1 and not force_collective):
        assert len(local) == total
        return local
    world, rank = dist.get_world_size(), dist.get_rank()
    lo, hi = shard_range(total, rank, world)
    assert len(local) == hi - lo, (len(local), lo, hi)
    cap = -(-total // world)                       # shards padded to equal length
    item = RESULT_DTYPE.itemsize
    buf = np.zeros(cap * item, dtype=np.uint8)
    buf[: len(local) * item] = np.frombuffer(local.tobytes(), dtype=np.uint8)
    dev = device if device is not None else collective_device()
    send = torch.from_numpy(buf).to(dev)
    recv = torch.empty(world * cap * item, dtype=torch.uint8, device=dev)
    dist.all_gather_into_tensor(recv, send)
    raw = recv.cpu().numpy()
    out = np.zeros(total, dtype=RESULT_DTYPE)
    for r in range(world):
        rlo, rhi = shard_range(total, r, world)
        chunk = raw[r * cap * item: r * cap * item + (rhi - rlo) * item]
        out[rlo:rhi] = np.frombuffer(chunk.tobytes(), dtype=RESULT_DTYPE)
    return out


def solve_sharded(shard, total, params=None, qp_settings=None):
    """One step of the sharded job on this rank: solve the rank's own shard (`shard` is a
    ``batch.TrajOptBatch`` holding problems [lo, hi) of the `total`), then all-gather the
    24-byte result records.  Returns (this rank's full fetch result, the (total,) records).
    No other inter-rank traffic exists on the path."""
    shard.solve(params, qp_settings)
    res = shard.fetch(with_merit=True)
    rec = pack_results(res.merit, res.max_violation, res.success, res.sqp_iters)
    return res, gather_results(rec, total)
