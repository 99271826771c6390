"""The N > 1 path of bench.py on real hardware, in the driver-run record (r02 verdict item 2).

A one-GPU box cannot give every rank its own device, so this is the REHEARSAL form: `bench.py --gpus 2` starts its own
two rank processes (torch.distributed.run as a fresh CHILD process -- never an exec of this pytest process, which has
touched the GPU), both on device 0: self_launch -> torchrun -> init_process_group -> shard -> solve_sharded (the real
libsco_hip.so) -> all-gather of the 24-byte records -> max-over-ranks clock -> rank 0's JSON line.  Problems are
independent (/root/reference/sco_py/sco_osqp/prob.py:48-86: all state hangs off one Prob), so what the gather returns
must equal two single-rank solves of the two shards."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from sco_py_amd import _lib, batch as sb, dist as sd
from sco_py_amd import workloads as af

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DIMS = dict(d=7, T=20, K=5, O=2)


def _single_rank_iters(total, world):
    """sqp_iters of the shards [lo, hi) solved one after the other in this process."""
    out = 0
    for r in range(world):
        lo, hi = sd.shard_range(total, r, world)
        arrays, _ = af.make_batch(hi - lo, first=lo, **DIMS)
        with sb.TrajOptBatch(hi - lo, DIMS["d"], DIMS["T"], DIMS["K"], DIMS["O"]) as tb:
            tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                    arrays["point_frac"], arrays["obstacles"])
            tb.solve()
            out += int(tb.fetch().sqp_iters.sum())
    return out


def _run_group(cmd, env, timeout=600):
    """Run a child that starts rank processes of its own (bench.py -> torch.distributed.run -> ranks) in a NEW SESSION: on a
    timeout the whole process group is killed -- the grandchildren have initialised the GPU and would otherwise keep
    device 0 for the rest of the test session -- and the test fails."""
    import signal
    proc = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        out, err = proc.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        proc.communicate()
        pytest.fail("child process group timed out and was killed: %r" % (cmd,))
    return subprocess.CompletedProcess(cmd, proc.returncode, out, err)


def _bench_two_ranks(mode, batch, extra=()):
    env = dict(os.environ)
    env["SCO_BENCH_REHEARSE"] = mode
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", str(batch),
           "--cpu-problems", "0", "--aux-12x50", "0", "--aux-b4096", "0", "--aux-object-api", "0"] + list(extra)
    return _run_group(cmd, env)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout[-2000:]
    return json.loads(lines[0])


def test_bench_with_two_ranks_on_one_device_gloo(gpu):
    p = _bench_two_ranks("1", 1024)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 2048 and out["scaling"] == "weak"
    assert out["aux"]["backend"] == "gloo" and out["value"] > 0 and out["roofline"]["frac"] > 0
    assert out["aux"]["sco_iters_per_step"] == _single_rank_iters(2048, 2)
    # what makes a first real multi-GPU run explain itself: every rank's own clock, SCO iterations and ADMM time
    pr = out["aux"]["per_rank"]
    assert len(pr["elapsed_s"]) == len(pr["sco_iters"]) == len(pr["admm_s"]) == 2
    assert sum(pr["sco_iters"]) == out["aux"]["sco_iters_per_step"] and all(0 < a <= e for a, e in zip(pr["admm_s"], pr["elapsed_s"]))
    assert 1.0 <= pr["imbalance_max_over_mean"] < 2.0 and max(pr["elapsed_s"]) <= out["ms_per_step"] * 1e-3 + 1e-3


def test_bench_strong_scaling_mode_splits_one_global_batch(gpu):
    """--global-batch N: N problems in total over the ranks (BASELINE configs[3] is 8192 over 8 GPUs); here 600 over two ranks
    on one device: the gathered SCO iterations are those of the two shards [0, 300), [300, 600)."""
    p = _bench_two_ranks("1", 1024, extra=["--global-batch", "600"])
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 600 and out["scaling"] == "strong"
    assert out["aux"]["sco_iters_per_step"] == _single_rank_iters(600, 2) and len(out["aux"]["per_rank"]["sco_iters"]) == 2


def test_bench_with_two_ranks_on_one_device_rccl_branch(gpu):
    """The nccl (= RCCL) branch of bench.py / dist.py with both ranks on device 0: init with device_id, all-gather and
    max-reduce on device tensors.  RCCL may refuse two ranks on one device ("duplicate GPU"); the branch has then at least
    been imported, argument-checked and run up to the communicator, and the test says which."""
    p = _bench_two_ranks("nccl", 64)
    if p.returncode != 0:
        err = (p.stderr or "")[-6000:]
        low = err.lower()
        assert "init_process_group" in err or "nccl" in low or "rccl" in low, err      # it failed in the communicator, nowhere else
        pytest.skip("RCCL refuses two ranks on one device here: " + " | ".join(l for l in err.splitlines() if "rror" in l)[-400:])
    out = _json_line(p.stdout)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 128 and out["aux"]["backend"] == "nccl"
    assert out["aux"]["sco_iters_per_step"] == _single_rank_iters(128, 2)


def test_rccl_calls_of_the_sharded_path_in_a_group_of_one_rank(gpu):
    """What a one-GPU box CAN run of the RCCL branch: a communicator of one rank is a real RCCL communicator.  A fresh child
    process (tests/rccl_one_rank.py) initialises the group exactly as bench.py does, solves a small shard through
    libsco_hip.so and sends its records through dist.gather_results (forced past the one-rank shortcut: all_gather_into_tensor
    on uint8 DEVICE tensors), dist.max_over_ranks (all_reduce MAX on a float64 device tensor) and two barriers."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 29500 + (os.getpid() % 400)
    p = _run_group([sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank.py"), str(port)], env)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    out = _json_line(p.stdout)
    assert out["backend"] == "nccl" and out["device"] == "cuda" and out["world"] == 1
    assert out["identical"] and out["clock"] == 1.25
    assert out["sqp_iters"] == _single_rank_iters(24, 1)
