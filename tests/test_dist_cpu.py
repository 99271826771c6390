"""Multi-process sharding path on CPU: world_size 2, gloo backend."""
import os
import socket
import sys

import numpy as np
import pytest

from sco_py_amd import dist as sd


def test_shard_ranges_partition_the_batch():
    for total in (0, 1, 7, 8, 1024, 8192, 8195):
        for world in (1, 2, 3, 8):
            spans = [sd.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        sd.shard_range(8, 2, 2)


def test_single_process_gather_is_identity():
    rec = sd.pack_results([1.0, 2.0], [0.0, 0.5], [True, False], [3, 4])
    assert sd.gather_results(rec, 2) is rec and rec.dtype.itemsize == 24


def _worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sd.shard_range(total, rank, world)
    idx = np.arange(lo, hi)
    rec = sd.pack_results(idx * 1.5, idx * 0.25, idx % 2 == 0, idx + 10)
    out = sd.gather_results(rec, total)
    q.put((rank, out.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("total", [10, 7])
def test_gloo_all_gather_of_result_records(total):
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    idx = np.arange(total)
    want = sd.pack_results(idx * 1.5, idx * 0.25, idx % 2 == 0, idx + 10)
    for r in range(2):
        assert np.array_equal(np.frombuffer(got[r], dtype=sd.RESULT_DTYPE), want)
