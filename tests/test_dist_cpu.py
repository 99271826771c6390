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


def test_rank_stats_without_a_process_group():
    st = sd.gather_rank_stats([1.5, 7.0, 0.5])
    assert st.shape == (1, 3) and st[0].tolist() == [1.5, 7.0, 0.5]


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
    assert sd.collective_device() == "cpu" and sd.max_over_ranks(1.0 + rank) == float(world)      # the clock of bench.py
    # bench.py's per-rank record (own clock, SCO iterations, ADMM time): every rank sees every rank's three numbers
    stats = sd.gather_rank_stats([0.5 + rank, 100.0 * (rank + 1), 0.25 * (rank + 1)])
    assert stats.shape == (world, 3) and stats[:, 0].tolist() == [0.5 + r for r in range(world)]
    assert stats[:, 1].tolist() == [100.0 * (r + 1) for r in range(world)]
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


def _one_rank_worker(port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    rec = sd.pack_results([1.0, 2.0, 3.0], [0.0, 0.5, 0.25], [True, False, True], [3, 4, 5])
    short = sd.gather_results(rec, 3)
    forced = sd.gather_results(rec, 3, force_collective=True)
    q.put((short is rec, forced is not rec and forced.tobytes() == rec.tobytes(), sd.max_over_ranks(2.5)))
    dist.destroy_process_group()


def test_group_of_one_rank_takes_the_collective_only_when_forced():
    """tests/test_dist_gpu.py runs the RCCL calls this way on a one-GPU box; here the same switch under gloo."""
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_one_rank_worker, args=(port, q))
    p.start()
    assert q.get(timeout=120) == (True, True, 2.5)
    p.join(timeout=60)
    assert p.exitcode == 0


class _StubShard(object):
    """Stands in for batch.TrajOptBatch on a box without a GPU: problem g's result is a function of g."""

    def __init__(self, lo, hi):
        self.idx = np.arange(lo, hi)
        self.solved_with = None

    def solve(self, params=None, qp_settings=None):
        self.solved_with = (params, qp_settings)

    def fetch(self, with_merit=True):
        from types import SimpleNamespace
        assert self.solved_with is not None and with_merit
        i = self.idx
        return SimpleNamespace(merit=0.5 * i, max_violation=1e-3 * i, success=(i % 3 != 0), sqp_iters=(i % 5 + 2).astype(np.int32))


def _step_worker(rank, world, port, total, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sd.shard_range(total, rank, world)
    res, allrec = sd.solve_sharded(_StubShard(lo, hi), total, "params", "qs")
    q.put((rank, len(res.merit), allrec.tobytes()))
    dist.barrier()
    dist.destroy_process_group()


def test_whole_step_function_at_world_size_two():
    """shard -> solve -> fetch -> pack -> all-gather, the function bench.py times, on two gloo ranks."""
    import torch.multiprocessing as mp
    total = 11
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_step_worker, args=(r, 2, port, total, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    i = np.arange(total)
    want = sd.pack_results(0.5 * i, 1e-3 * i, i % 3 != 0, i % 5 + 2)
    assert sorted(g[1] for g in got) == [5, 6]
    for _, _, raw in got:
        assert np.array_equal(np.frombuffer(raw, dtype=sd.RESULT_DTYPE), want)
    assert int(want["sqp_iters"].sum()) == int((i % 5 + 2).sum())


_RANK_SCRIPT = """
import os, sys, json
import torch.distributed as dist
dist.init_process_group("gloo")
import torch
t = torch.tensor([float(os.environ["RANK"]) + 1.0])
dist.all_reduce(t)
if dist.get_rank() == 0:
    print(json.dumps({"n_gpus": dist.get_world_size(), "sum": float(t.item()), "argv": sys.argv[1:]}))
dist.barrier()
dist.destroy_process_group()
"""


def test_bench_starts_its_own_rank_processes(tmp_path):
    """`python bench.py --gpus N` without torchrun: bench.self_launch starts N fresh rank processes as a child
    job (checked here with a CPU rank script in place of bench.py itself) and hands back its exit code."""
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "rank_script.py"
    script.write_text(_RANK_SCRIPT)
    code = ("import sys; sys.path.insert(0, %r); import bench; "
            "sys.exit(bench.self_launch(2, ['--gpus', '2'], script=%r))" % (root, str(script)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [l for l in out.stdout.splitlines() if l.startswith("{")][-1]
    rec = json.loads(line)
    assert rec == {"n_gpus": 2, "sum": 3.0, "argv": ["--gpus", "2"]}


def test_bench_main_relaunches_before_touching_the_gpu():
    """The --gpus N > 1 branch sits in front of every torch / library import of main()."""
    import ast
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tree = ast.parse(open(os.path.join(root, "bench.py")).read())
    main = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main"][0]
    first_launch = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "self_launch")
    imports = [n.lineno for n in ast.walk(main) if isinstance(n, (ast.Import, ast.ImportFrom))]
    calls = [n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "cpu_baseline_all_cores"]
    assert first_launch < min(imports) and first_launch < min(calls)
