"""Headline benchmark: SCO iterations / second over a batch of trajectory problems.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Metric (BASELINE.json): SCO iters/sec, one SCO iteration = one pass of the outer
loop body of the reference's _min_merit_fn (sco_py/sco_osqp/solver.py:126-253) for
one problem.  Workload: BASELINE.json configs[2], "batch=1024 independent 7-DOF x
20-timestep problems, 1 MI355X (ADMM throughput run)" per GPU (weak scaling:
N GPUs solve N x 1024 distinct seeded problems; 8 GPUs = configs[3], batch 8192).
One "step" = one complete penalty-SQP solve of the rank's whole shard starting from
the state resident in HBM, followed (N > 1) by the RCCL all-gather of the 24-byte
per-problem result records.  Reference solver defaults (solver.py:17-28,
osqp_utils.py:10-15), reference quirks reproduced (parity mode).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12            # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md, chip-level parameters)
N_CU = 256
CLOCK_HZ = 2.4e9             # spec maximum; the chip holds less under load, so the on-chip fractions below read low
F64_PEAK = N_CU * 64 * 2 * CLOCK_HZ      # 78.6 TFLOP/s: 64 f64 FMA / clk / CU; MI355X's dense f64 MFMA rate is the same
LDS_PEAK = N_CU * 256 * CLOCK_HZ         # 157 TB/s: 256 B / clk / CU for 8- and 16-byte LDS reads (MI355X_MICROARCH.md, LDS)


def algorithmic_bytes_per_iter(n, m):
    # reads x~,x,q (3n) + nu,z,y,l,u,rho (6m); writes x,rhs_top (2n) + z,y,rhs_bottom (3m)
    return (5 * n + 9 * m) * 8


def onchip_model(d, T, K, O, projection=False):
    """Work of ONE ADMM iteration of one QP of the planar-arm family as the row-local kernel does it:
    flops: the dense core solve W r (n_c^2 FMA), A x~ and A' t' (one FMA per non-zero of A each), 12 per constraint
    row (relaxation, projection, dual step, next right-hand side) and 4 per variable; LDS bytes: the right-hand-side
    broadcast of the W tile (512 threads x 9 doubles), 32 B per operand pair of the row and column gather-dots
    (16 B of A values + 16 B of operands), the three vector writes and the x~_C read-back."""
    n_x, R = d * T, K * O
    if projection:
        n, m, nnzA, n_c = n_x, 2 * d + n_x, 2 * d + n_x, n_x
        col_pairs, row_pairs = n_x * 1, (2 * d + n_x) * 1
    else:
        n = n_x + T * R
        m = 2 * d + T * R + n
        nnzA = 2 * d + T * R * (d + 1) + n
        n_c = n_x
        col_pairs = n_x * (-(-R // 2) + 1)                 # the block's R rows in aligned pairs + (trust row, pin)
        row_pairs = T * R * (d // 2 + 1) + (n_x + 2 * d)   # d consecutive core entries -> d/2 + 1 pairs; single-entry rows
    flops = 2 * (n_c * n_c + 2 * nnzA) + 12 * m + 4 * n
    lds = 512 * 9 * 8 + 32 * (col_pairs + row_pairs) + 8 * (2 * n_c + m) + 8 * n_c
    return flops, lds


def wavefront_model(d, T, K, O):
    """Counted flops of ONE ADMM iteration of one penalty QP on the wavefront tier (csrc/sco_admm_wv.hip): the
    core solve is the twisted block-tridiagonal sweep -- every block but the middle one is visited twice (forward, backward),
    a visit = a d x d mat-vec + d coupling multiply-adds -- instead of the dense n_c x n_c mat-vec of the row-local kernel;
    A x~ and A' t' (one FMA per non-zero each), 12 per row and 4 per variable as in onchip_model."""
    n_x, R = d * T, K * O
    n = n_x + T * R
    m = 2 * d + T * R + n
    nnzA = 2 * d + T * R * (d + 1) + n
    sweep = (2 * (T - 1) + 1) * (d * d + d)
    return 2 * (sweep + 2 * nnzA) + 12 * m + 4 * n


def kernel_src_sha():
    """Hash of every kernel source: a PMC measurement is only quoted for the kernels it was taken on."""
    import hashlib
    h = hashlib.sha256()
    src = os.path.join(ROOT, "sco_py_amd", "csrc")
    for f in sorted(os.listdir(src)):
        if f.endswith((".hip", ".h", ".cpp")):
            with open(os.path.join(src, f), "rb") as fh:
                h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(name):
    """HBM bytes of the dominant kernel from the committed PMC passes (scripts/gpu_pmc.sh -> profiles/<name>), or
    None when the kernels have changed since they were taken."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, "no PMC record"
    with open(path) as fh:
        tj = json.load(fh)
    if tj.get("kernel_src_sha") != kernel_src_sha():
        return None, "stale PMC record (taken on kernel sources %s)" % tj.get("kernel_src_sha")
    return tj, path


def cpu_baseline(n_problems, first, dims):
    """The oracle (CPU restatement of the same path) timed on this host, ONE thread (BLAS pools limited to 1)."""
    from oracle import arm_family as af
    from oracle import sco_ref as sr
    try:
        from threadpoolctl import threadpool_limits
    except Exception:                      # pragma: no cover - threadpoolctl ships with the image
        import contextlib
        threadpool_limits = lambda limits: contextlib.nullcontext()     # noqa: E731
    with threadpool_limits(limits=1):
        sr.penalty_sqp(sr.trajopt_flat(af.make_problem(first, d=3, T=6, K=2, O=2)), emulate_memo=True)   # loads the C library
        iters = 0
        t0 = time.perf_counter()
        for i in range(n_problems):
            out = sr.penalty_sqp(sr.trajopt_flat(af.make_problem(first + i, **dims)), emulate_memo=True)
            iters += out.sqp_iters
        dt = time.perf_counter() - t0
    return iters / dt, dt, iters


# One worker per host core.  It imports everything and solves a tiny problem first (start-up and the C library load stay
# outside the timer), reports "ready", waits for "go", then solves its share and prints its SCO iteration count.
_CPU_WORKER = """
import sys
sys.path.insert(0, sys.argv[1])
from oracle import arm_family as af
from oracle import sco_ref as sr
dims = dict(d=int(sys.argv[2]), T=int(sys.argv[3]), K=int(sys.argv[4]), O=int(sys.argv[5]))
sr.penalty_sqp(sr.trajopt_flat(af.make_problem(0, d=3, T=6, K=2, O=2)), emulate_memo=True)
print("ready", flush=True)
sys.stdin.readline()
it = 0
for i in sys.argv[6:]:
    it += sr.penalty_sqp(sr.trajopt_flat(af.make_problem(int(i), **dims)), emulate_memo=True).sqp_iters
print(it, flush=True)
"""


def cpu_baseline_all_cores(dims, per_core=8, timeout=240):
    """One single-threaded oracle process per host core of this box, each solving `per_core` problems of the same
    batch (the generous CPU baseline of SURVEY 8(d); the reference itself is single-threaded).  The timer runs from
    the moment every worker has reported ready to the last result.  Plain child processes, started before anything
    touches the GPU; returns None if a worker fails or overruns."""
    import subprocess
    nproc = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = _cpu_share()                      # a container may own fewer CPUs than it can see
    cores = max(1, min(nproc, share if share else nproc, 64))
    # children never touch the GPU: drop any profiler preload / tool hooks from their environment
    env = {k: v for k, v in os.environ.items()
           if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF", "HSA_TOOLS", "ROCTRACER"))}
    for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        env[k] = "1"
    t_start = time.perf_counter()
    procs = []
    for c in range(cores):
        ids = [str(c * per_core + k) for k in range(per_core)]
        procs.append(subprocess.Popen([sys.executable, "-c", _CPU_WORKER, ROOT] +
                                      [str(dims[k]) for k in ("d", "T", "K", "O")] + ids,
                                      stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=env))
    iters, ok = 0, True
    try:
        for pr in procs:                                  # ready handshake
            if pr.stdout.readline().decode().strip() != "ready":
                ok = False
        if ok:
            t0 = time.perf_counter()
            for pr in procs:
                pr.stdin.write(b"go\n"); pr.stdin.flush()
            for pr in procs:
                out, _ = pr.communicate(timeout=max(1.0, timeout - (time.perf_counter() - t_start)))
                iters += int(out.decode().strip())
            dt = time.perf_counter() - t0
    except Exception:
        ok = False
    if not ok:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        return None
    return {"value": iters / dt, "unit": "sco_iters/s", "cores": cores, "nproc": nproc, "cpu_share": share, "kind": "port",
            "threads_per_process": 1,
            "sample": "problems 0..%d of the same batch, %d per single-threaded oracle process, one process per core, "
                      "%.1f s after the ready handshake (start-up excluded)" % (cores * per_core - 1, per_core, dt)}


def _cpu_share():
    """CPUs this container may actually use (cgroup quota), or None if unlimited / unknown."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:                     # cgroup v2: "<quota> <period>" or "max <period>"
            quota, period = fh.read().split()[:2]
        if quota != "max":
            return max(1, int(int(quota) / int(period)))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:        # cgroup v1
            quota = int(fh.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
            period = int(fh.read())
        if quota > 0:
            return max(1, quota // period)
    except Exception:
        pass
    return None


def build_arm_prob(pr):
    """One problem record of the 7x20 workload as a ``Prob`` of the reference's object API (the construction a sco_py
    caller writes: OSQPVar atoms, a trajectory Variable, QuadExpr objective, EqExpr pins, one LEqExpr block per timestep
    -- tests/trajopt_build.py is the same code), with device expressions as the constraint bodies."""
    from sco_py_amd import devexpr as dx, expr as ex
    from sco_py_amd.sco_osqp import osqp_utils as ou, prob as pb, variable as vr
    d, T = pr["d"], pr["T"]
    n_x = d * T
    prob = pb.Prob()
    atoms = np.empty((n_x, 1), dtype=object)
    for t in range(T):
        for j in range(d):
            atoms[t * d + j, 0] = ou.OSQPVar("q%03d_%02d" % (t, j))
            prob.add_osqp_var(atoms[t * d + j, 0])
    traj = vr.Variable(atoms, pr["x0"].reshape(n_x, 1))
    prob.add_var(traj)
    Q = np.zeros((n_x, n_x))
    i = np.arange(n_x - d)
    Q[i, i] += 2.0; Q[i + d, i + d] += 2.0; Q[i, i + d] -= 2.0; Q[i + d, i] -= 2.0
    prob.add_obj_expr(ex.BoundExpr(ex.QuadExpr(Q, np.zeros((1, n_x)), np.zeros((1, 1))), traj))
    pins = np.zeros((2 * d, n_x))
    pins[np.arange(d), np.arange(d)] = 1.0; pins[d + np.arange(d), (T - 1) * d + np.arange(d)] = 1.0
    prob.add_cnt_expr(ex.BoundExpr(ex.EqExpr(ex.AffExpr(pins, np.zeros((2 * d, 1))),
                                             np.concatenate([pr["start"], pr["goal"]]).reshape(-1, 1)), traj))
    R = pr["K"] * pr["O"]
    for t in range(T):
        sv = vr.Variable(atoms[t * d:(t + 1) * d, :], pr["x0"][t * d:(t + 1) * d].reshape(d, 1))
        e = dx.ArmCirclesExpr(pr["link_len"], pr["point_link"], pr["point_frac"], pr["obstacles"])
        prob.add_cnt_expr(ex.BoundExpr(ex.LEqExpr(e, np.zeros((R, 1))), sv))
    return prob, traj


def object_api_line(B, array_step_s):
    """aux.object_api_1024: B Probs -> solve_many.  Reports the Python construction, the compile step, and the solve
    (device solve + fetch: the part bench.py times as a step of the array API) separately."""
    from sco_py_amd import workloads as af
    from sco_py_amd.sco_osqp import batching
    t0 = time.perf_counter()
    built = [build_arm_prob(af.make_problem(i)) for i in range(B)]
    build_s = time.perf_counter() - t0
    probs = [b[0] for b in built]
    t0 = time.perf_counter()
    oks, stats = batching.solve_many(probs)
    total_s = time.perf_counter() - t0
    tm = stats["last_device"]
    host_evals = sum(be.expr.expr.host_evals for p in probs for be in p._nonlin_cnt_exprs)
    return {"workload": "the %d problems of the headline batch as %d Prob objects (Variable / BoundExpr / LEqExpr on device "
                        "expressions), solve_many -> compile_prob -> one device batch; parity mode" % (B, B),
            "build_probs_s": build_s, "compile_s": stats["compile_s"], "upload_s": tm["load_s"],
            "solve_fetch_s": tm["solve_fetch_s"], "trace_s": tm["trace_s"], "solve_many_total_s": total_s,
            "array_api_step_s": array_step_s, "solve_vs_array_api": tm["solve_fetch_s"] / array_step_s,
            "sco_iters_per_s_solve": float(tm["sqp_iters"].sum()) / tm["solve_fetch_s"],
            "sco_iters_per_s_incl_compile_and_write_back": float(tm["sqp_iters"].sum()) / total_s,
            "device_problems": stats["device_problems"], "device_batches": stats["device_batches"],
            "python_f_calls": int(host_evals), "success_fraction": float(np.mean(oks))}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch(n, argv, script=None):
    """`python bench.py --gpus N` without torchrun: start N fresh rank processes (torch.distributed.run as a child
    process, never an exec) BEFORE this process makes any GPU call, and return its exit code.  Rank 0 of the child
    job prints the one JSON line."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), script or os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", "1")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=None, help="problems per GPU (default 1024; 256 for 12x50)")
    ap.add_argument("--global-batch", type=int, default=None,
                    help="STRONG scaling: this many problems in total, split over the GPUs (e.g. 8192 = BASELINE configs[3]); "
                         "the default is weak scaling, --batch problems per GPU")
    ap.add_argument("--workload", choices=["7x20", "12x50"], default="7x20",
                    help="7x20 = BASELINE configs[2] (headline); 12x50 = configs[4] shape (structured global-memory tier)")
    ap.add_argument("--cpu-problems", type=int, default=16, help="size of the CPU-baseline sample (0 = skip)")
    ap.add_argument("--aux-12x50", type=int, default=2048,
                    help="batch of the 12-DOF x 50 line reported under aux in a default 7x20 run (0 = skip)")
    ap.add_argument("--aux-b4096", type=int, default=1, help="also report a 4096-problem 7x20 step under aux (0 = skip)")
    ap.add_argument("--aux-object-api", type=int, default=1,
                    help="also solve the 1024 problems as 1024 Prob objects through solve_many, reported under aux (0 = skip)")
    ap.add_argument("--intended", action="store_true",
                    help="disable reference quirks Q1/Q2 (NOT the headline number)")
    ap.add_argument("--beyond", action="store_true",
                    help="--intended plus the opt-in extensions: adaptive rho, warm-started QPs, at most 20 QPs "
                         "per problem (NOT the headline number, NOT parity mode)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))          # nothing has touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    cpu_all = None
    if args.intended or args.beyond:
        args.cpu_problems = 0          # the CPU baseline below times the parity-mode oracle
    if world == 1 and args.cpu_problems > 0 and args.workload == "7x20":
        cpu_all = cpu_baseline_all_cores(dict(d=7, T=20, K=5, O=2))
    import torch
    import torch.distributed as dist
    # rehearsal on a 1-GPU box: SCO_BENCH_REHEARSE=1 (or "gloo") puts every rank on device 0 and uses gloo;
    # SCO_BENCH_REHEARSE=nccl keeps the RCCL branch (init with device_id, all-gather and max-reduce on device tensors) with
    # every rank on device 0 -- RCCL may refuse two ranks on one device, then the run fails at init and says so.
    # The driver's real multi-GPU run never sets the variable.
    rehearse_mode = os.environ.get("SCO_BENCH_REHEARSE", "0")
    rehearse = rehearse_mode in ("1", "gloo", "nccl")
    use_gloo = rehearse_mode in ("1", "gloo")
    if rehearse:
        local_rank = 0
    if world > 1:
        torch.cuda.set_device(local_rank)
        if use_gloo:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world:
        if rank == 0:
            print("bench.py: --gpus %d but WORLD_SIZE %d: the launcher and the flag disagree" % (args.gpus, world),
                  file=sys.stderr)
        sys.exit(2)

    from sco_py_amd import workloads as af       # seeded inputs (shared with tests and oracle)
    from sco_py_amd import _lib, batch as sb
    from sco_py_amd import dist as sd

    def sync():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    def run(workload, B, steps, warmup, intended=False, beyond=False):
        """Load `B` problems per rank (resident in HBM before the clock starts), time `steps` complete sharded solves."""
        big = workload == "12x50"
        dims = dict(d=12, T=50, K=10, O=10) if big else dict(d=7, T=20, K=5, O=2)
        total = args.global_batch if (args.global_batch and not big) else B * world
        lo, hi = sd.shard_range(total, rank, world)
        arrays, _ = af.make_batch(hi - lo, first=lo, **dims)
        params = _lib.default_sqp_params()
        if intended or beyond:
            params.compound_penalty = 0; params.duplicate_rows = 0
        qs = _lib.default_qp_settings()
        if beyond:
            params.max_sqp_iters = 20; params.warm_start_qps = 1; qs.adaptive_rho = 1
        tb = sb.TrajOptBatch(hi - lo, dims["d"], dims["T"], dims["K"], dims["O"], device=local_rank)
        tb.load(arrays["x0"], arrays["start"], arrays["goal"], arrays["link_len"], arrays["point_link"],
                arrays["point_frac"], arrays["obstacles"])          # inputs resident in HBM from here on
        for _ in range(warmup):
            sd.solve_sharded(tb, total, params, qs)
        # the interpreter's cyclic collector walks every object torch has imported (a full pass took 40-75 ms in the
        # middle of a step): park what exists now in the permanent generation
        import gc
        gc.collect(); gc.freeze()
        sync()
        t0 = time.perf_counter()
        sco_iters = 0; admm_ms = 0.0; qp_launches = 0; groups = 1
        wv = dict(ms=0.0, iters=0, launches=0, other_ms=0.0, other_launches=0)
        my_iters = 0
        stage_ms = np.zeros(5)
        for _ in range(steps):
            res, allrec = sd.solve_sharded(tb, total, params, qs)   # solve the shard + RCCL all-gather of 24 B/problem (no-op at N = 1)
            sco_iters += int(allrec["sqp_iters"].sum())
            my_iters += int(res.sqp_iters.sum())
            tm = tb.last_timing()                                  # HIP-event sums of the step just finished, taken on the library's stream
            stage_ms += [tm["convexify_ms"], tm["qp_setup_ms"], tm["admm_ms"], tm["decide_ms"], tm["total_ms"]]
            admm_ms += tm["admm_ms"]
            qp_launches += tm["launches"]                         # penalty-QP launches over all stream groups (the projection launch is tiny)
            groups = tm["groups"]
            wv["ms"] += tm["wv_ms"]; wv["iters"] += tm["wv_iters"]; wv["launches"] += tm["wv_launches"]
            wv["other_ms"] += tm["other_admm_ms"]; wv["other_launches"] += tm["other_launches"]
        my_elapsed = time.perf_counter() - t0               # this rank's own clock, before the closing barrier
        sync()
        elapsed = time.perf_counter() - t0
        # bookkeeping outside the timed region: every step solves the same loaded problems and the solve is
        # deterministic (tests/test_sqp_gpu.py), so the per-QP iteration counts of the last step are those of every step
        traces = tb.trace()
        it_proj = steps * sum(int(t[0, 7]) for t in traces)
        it_pen = steps * sum(int(t[1:, 7].sum()) for t in traces)
        qp_solves = steps * int(res.qp_solves.sum())
        tb.close()
        # per-rank record of the step (north_star risk: data-dependent iteration counts -> unequal shards): own clock, own
        # SCO iterations, own ADMM time; gathered like the result records (three doubles per rank), imbalance = max / mean
        per_rank = sd.gather_rank_stats([my_elapsed, float(my_iters), admm_ms * 1e-3]) if world > 1 else np.array([[my_elapsed, float(my_iters), admm_ms * 1e-3]])
        if world > 1:
            elapsed = sd.max_over_ranks(elapsed)                  # device tensor under RCCL, host tensor under gloo
        return dict(dims=dims, B=B, total=total, steps=steps, elapsed=elapsed, sco_iters=sco_iters, admm_ms=admm_ms, wv=wv, per_rank=per_rank,
                    qp_launches=qp_launches, groups=groups, stage_ms=stage_ms / steps, it_proj=it_proj, it_pen=it_pen, qp_solves=qp_solves,
                    success=float(np.mean(allrec["success"] != 0)))

    def roofline(r, big):
        """Roofline block of the dominant kernel: algorithmic work of this rank's launches / their HIP-event time."""
        d = r["dims"]; n_x = d["d"] * d["T"]; R = d["K"] * d["O"]
        n = n_x + d["T"] * R; m = 2 * d["d"] + d["T"] * R + n
        n0, m0 = n_x, 2 * d["d"] + n_x
        secs = r["admm_ms"] * 1e-3
        launches = max(r["qp_launches"], 1)
        alg_bytes = r["it_proj"] * algorithmic_bytes_per_iter(n0, m0) + r["it_pen"] * algorithmic_bytes_per_iter(n, m)
        hbm_eq = {"label": "HBM-equivalent: the bytes an ADMM that streams its iterates through HBM would move, SURVEY 8(d) "
                           "(5n + 9m) * 8 per problem-iteration; NOT traffic of this kernel",
                  "achieved_GBps": alg_bytes / secs / 1e9 if secs > 0 else 0.0,
                  "ratio_to_8TBps": alg_bytes / secs / HBM_PEAK if secs > 0 else 0.0,
                  "algorithmic_bytes_per_launch": alg_bytes / launches}
        if big:
            tj, src = measured_traffic("r04_traffic_12x50.json")
            # a PMC record is quoted for the batch it was taken at only (the working set, and with it the share of the
            # Infinity Cache, follows the batch)
            if tj and tj.get("batch") != r["B"]:
                tj, src = None, "PMC record taken at batch %s, this run has %d" % (tj.get("batch"), r["B"])
            traffic = tj["hbm_bytes_per_problem_iteration"] * (r["it_proj"] + r["it_pen"]) / launches if tj else None
            return {"bound": "hbm", "kernel": "qp_admm_bt_kernel", "achieved": hbm_eq["achieved_GBps"], "peak": HBM_PEAK / 1e9,
                    "unit": "GB/s", "frac": hbm_eq["ratio_to_8TBps"], "traffic": traffic, "traffic_source": src,
                    "algorithmic_bytes_per_launch": hbm_eq["algorithmic_bytes_per_launch"],
                    "note": "one workgroup per problem streams A (565 KB) and the row state from L2 / MALL / HBM every "
                            "iteration: achieved = algorithmic (5n+9m)*8 B per problem-iteration x iterations / kernel time "
                            "(HIP events on the library stream)"}
        f_pen, l_pen = onchip_model(d["d"], d["T"], d["K"], d["O"])
        f_pro, l_pro = onchip_model(d["d"], d["T"], d["K"], d["O"], projection=True)
        f_wv = wavefront_model(d["d"], d["T"], d["K"], d["O"])
        its_all = r["it_pen"] + r["it_proj"]
        w = r["wv"]
        # the step's ADMM launches by kernel: wavefront tier (rounds with >= ~3 live problems per CU) and the row-local
        # kernel (the tail of the step, the projection QPs); each priced against its OWN HIP-event time and its own count
        # of problem-iterations (the library counts the wavefront kernel's on the device)
        its_wv, secs_wv = w["iters"], w["ms"] * 1e-3
        its_rl, secs_rl = its_all - its_wv, w["other_ms"] * 1e-3
        flops_rl = max(its_rl - r["it_proj"], 0) * f_pen + r["it_proj"] * f_pro
        tiers = {
            "wavefront": {"kernel": "qp_admm_wv_kernel<7,4,3,10,3>", "ms_per_step": w["ms"] / r["steps"], "launches_per_step": w["launches"] / r["steps"],
                          "avg_launch_ms": w["ms"] / max(w["launches"], 1), "problem_iterations_per_step": its_wv / r["steps"],
                          "problem_iterations_per_us_per_cu": its_wv / secs_wv / 1e6 / N_CU if secs_wv > 0 else None,
                          "flops_per_problem_iteration": f_wv, "valu_frac": its_wv * f_wv / secs_wv / F64_PEAK if secs_wv > 0 else None},
            "row_local": {"kernel": "qp_admm_rl_kernel<5,9,12,false,2,0>", "ms_per_step": w["other_ms"] / r["steps"],
                          "launches_per_step": w["other_launches"] / r["steps"], "avg_launch_ms": w["other_ms"] / max(w["other_launches"], 1),
                          "problem_iterations_per_step": its_rl / r["steps"],
                          "problem_iterations_per_us_per_cu": its_rl / secs_rl / 1e6 / N_CU if secs_rl > 0 else None,
                          "flops_per_problem_iteration": f_pen, "valu_frac": flops_rl / secs_rl / F64_PEAK if secs_rl > 0 else None},
        }
        use_wv = secs_wv >= secs_rl
        dom = tiers["wavefront" if use_wv else "row_local"]
        flops = its_wv * f_wv if use_wv else flops_rl
        dsecs = secs_wv if use_wv else secs_rl
        dlaunches = max(w["launches"] if use_wv else w["other_launches"], 1)
        tj, src = measured_traffic("r04_traffic.json")
        key = "wavefront" if use_wv else "row_local"
        traffic = None
        if tj and r["B"] == tj.get("batch") and key in tj.get("hbm_bytes_per_launch", {}):
            traffic = tj["hbm_bytes_per_launch"][key]
        return {"bound": "valu_f64", "kernel": dom["kernel"], "achieved": flops / dsecs / 1e12 if dsecs > 0 else 0.0,
                "peak": F64_PEAK / 1e12, "unit": "TFLOP/s", "frac": flops / dsecs / F64_PEAK if dsecs > 0 else 0.0,
                "traffic": traffic, "traffic_source": "PMC passes stored under " + src if tj else src,
                "problem_iterations_per_us_per_cu": {"this_run_all_admm_launches": its_all / secs / 1e6 / N_CU if secs > 0 else None,
                                                     "r03_row_local_only": 1.05},
                "frac_at_the_r03_flop_count": its_all * f_pen / secs / F64_PEAK if secs > 0 else None,
                "tiers": tiers,
                "onchip": {"valu_frac": flops / dsecs / F64_PEAK if dsecs > 0 else 0.0,
                           "flops_per_problem_iteration": dom["flops_per_problem_iteration"],
                           "clock_ghz_assumed": CLOCK_HZ / 1e9, "clock_ghz_measured": (tj or {}).get("clock_ghz_measured"), "cus": N_CU},
                "hbm_measured": {"bytes_per_launch": traffic,
                                 "frac_of_8TBps": traffic * dlaunches / dsecs / HBM_PEAK if traffic and dsecs > 0 else None},
                "hbm_equivalent": hbm_eq,
                "note": "bound = the f64 VECTOR ALU (no MFMA in either ADMM kernel).  r04 changed the algebra of the kernel that runs "
                        "most of the step: a twisted block-tridiagonal sweep (20.2 k counted flops per problem-iteration) instead "
                        "of the dense 140 x 140 inverse (55.0 k), one wavefront per problem and four problems per CU -- so frac "
                        "(counted flops / peak) FELL while problem-iterations per second rose; compare problem_iterations_per_us_per_cu "
                        "and frac_at_the_r03_flop_count (what the same launches would score at the old flop count).  Iterates live in "
                        "registers / LDS for a whole launch: HBM sees the problem once per launch (hbm_measured, PMC)"}

    big = args.workload == "12x50"
    B = args.batch if args.batch is not None else (256 if big else 1024)
    if big and args.cpu_problems == 16:
        args.cpu_problems = 0          # one 12x50 oracle solve takes minutes (tests/golden/make_big_oracle.py)
    r = run(args.workload, B, args.steps, args.warmup, args.intended, args.beyond)
    aux12 = None
    if world == 1 and not big and not (args.intended or args.beyond) and args.aux_12x50 > 0:
        # BASELINE configs[4] shape on the structured global-memory tier: one step, reported under aux (a failure here
        # must not take the headline line with it)
        try:
            r12 = run("12x50", args.aux_12x50, 1, 0)
            rf12 = roofline(r12, True)
            aux12 = {"workload": "batch=%d x 12-DOF x 50-timestep (n=5600, m=10624, 5000 nonlinear rows) per GPU, parity mode, 1 step; "
                                 "structured global-memory ADMM tier on the f64 vector ALU; the one GEMM-shaped stage of the path, the block "
                                 "normal matrices J' R J of the QP setup (100 x 12 per block), runs on v_mfma_f64_16x16x4 since r04, "
                                 "bit-identically (qp_setup stage -21 %%, 0.3 %% of this step: profiles/r04_n1_mfma.txt)" % args.aux_12x50,
                     "sco_iters_per_s": r12["sco_iters"] / r12["elapsed"], "ms_per_step": 1e3 * r12["elapsed"],
                     "admm_problem_iterations_per_s": (r12["it_proj"] + r12["it_pen"]) / (r12["admm_ms"] * 1e-3),
                     "roofline": {k: rf12[k] for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic")}}
        except Exception as e:                                  # pragma: no cover
            aux12 = {"error": "%s: %s" % (type(e).__name__, e)}

    aux4096 = None
    if world == 1 and not big and B == 1024 and not (args.intended or args.beyond) and args.aux_b4096:
        # the north_star's batch (4096 problems on one GPU): one warm-up, one step, reported under aux
        try:
            r4 = run("7x20", 4096, 1, 1)
            aux4096 = {"workload": "batch=4096 x 7-DOF x 20-timestep on one GPU, parity mode, 1 step",
                       "sco_iters_per_s": r4["sco_iters"] / r4["elapsed"], "ms_per_step": 1e3 * r4["elapsed"],
                       "admm_ms": r4["admm_ms"], "onchip_frac": roofline(r4, False)["frac"]}
        except Exception as e:                                  # pragma: no cover
            aux4096 = {"error": "%s: %s" % (type(e).__name__, e)}

    aux_obj = None
    if world == 1 and not big and B == 1024 and not (args.intended or args.beyond) and args.aux_object_api:
        # the reference's OBJECT API on the same 1024 problems: one Prob per problem built from Variable / BoundExpr
        # objects (device expressions), all of them through solve_many -> compile_prob -> ONE device batch
        try:
            aux_obj = object_api_line(1024, r["elapsed"] / args.steps)
        except Exception as e:                                  # pragma: no cover
            aux_obj = {"error": "%s: %s" % (type(e).__name__, e)}

    if rank == 0:
        dims = r["dims"]
        out = {
            "metric": "SCO iters/sec (batch of N trajopt QPs)",
            "value": r["sco_iters"] / r["elapsed"],
            "unit": "sco_iters/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * r["elapsed"] / args.steps,
            "higher_is_better": True, "scaling": "strong" if (args.global_batch and not big) else "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("batch=%d independent 12-DOF x 50-timestep planar-arm trajopt problems per GPU "
                                    "(n=5600, m=10624 + duplicated penalty rows, 5000 nonlinear rows), penalty SQP with "
                                    "reference defaults; structured global-memory ADMM tier, f64 vector ALU; block normal "
                                    "matrices of the QP setup on v_mfma_f64_16x16x4 (profiles/r04_n1_mfma.txt, DESIGN.md 5)" % B) if big else
                                   ("batch=%d independent 7-DOF x 20-timestep planar-arm trajopt problems per GPU "
                                    "(n=340, m=554 + duplicated penalty rows, 200 nonlinear rows), penalty SQP with "
                                    "reference defaults" % B),
                       "global_batch": r["total"],
                       "mode": "beyond-parity (quirks off, adaptive rho, warm QPs, <= 20 QPs)" if args.beyond else
                               ("intended" if args.intended else "parity"),
                       "parallelism": "batch-shard x%d, no data-path collective" % world},
            "aux": {"qp_solves_per_s": r["qp_solves"] * world / r["elapsed"],
                    "admm_iters_per_s": (r["it_proj"] + r["it_pen"]) * world / r["elapsed"],
                    "stage_ms_per_step": dict(zip(["convexify", "qp_setup", "admm", "decide", "total"],
                                                  r["stage_ms"].round(3).tolist())),
                    "success_fraction": r["success"],
                    "sco_iters_per_step": r["sco_iters"] / args.steps,          # over ALL ranks (from the gathered records)
                    "backend": (dist.get_backend() if world > 1 else None),
                    # per rank: own clock of the timed region (s), own SCO iterations, own ADMM device time (s)
                    "per_rank": {"elapsed_s": r["per_rank"][:, 0].round(6).tolist(), "sco_iters": r["per_rank"][:, 1].astype(int).tolist(),
                                 "admm_s": r["per_rank"][:, 2].round(6).tolist(),
                                 "imbalance_max_over_mean": float(r["per_rank"][:, 0].max() / max(r["per_rank"][:, 0].mean(), 1e-12))},
                    "admm_launches_per_step": r["qp_launches"] / args.steps,
                    "stream_groups": r["groups"],
                    "kernel_src_sha": kernel_src_sha()},
            "roofline": roofline(r, big),
        }
        if aux12 is not None:
            out["aux"]["config4_12x50"] = aux12
        if aux4096 is not None:
            out["aux"]["batch_4096"] = aux4096
        if aux_obj is not None:
            out["aux"]["object_api_1024"] = aux_obj
        if world == 1 and args.cpu_problems > 0:
            v, dt, it = cpu_baseline(args.cpu_problems, 0, dims)
            out["cpu_baseline"] = {"value": v, "unit": "sco_iters/s", "cores": 1, "kind": "port",
                                   "sample": "problems 0..%d of the same batch, oracle/sco_ref.py + "
                                             "oracle/osqp_ref.c, %.1f s" % (args.cpu_problems - 1, dt)}
            if aux4096 is not None and "sco_iters_per_s" in aux4096:
                aux4096["vs_cpu_1core"] = aux4096["sco_iters_per_s"] / v
        if cpu_all is not None:
            out["aux"]["cpu_baseline_all_cores"] = cpu_all
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
