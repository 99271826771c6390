"""Build libsco_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

Every source is compiled to its own object file (in parallel, only when it or a header is newer
than the object), then linked: a one-kernel edit rebuilds in seconds instead of a minute."""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "build")
LIB = os.path.join(CSRC, "libsco_hip.so")
SOURCES = ["sco_qp.hip", "sco_admm_fast.hip", "sco_admm_reg.hip", "sco_admm_rl.hip", "sco_admm_wv.hip", "sco_qp_big.hip", "sco_sqp.hip", "qp_plan.cpp"]
HEADERS = ["sco_internal.h", "qp_plan.h", os.path.join("..", "..", "include", "sco_hip.h")]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]


def hipcc_path():
    p = shutil.which("hipcc")
    if p:
        return p
    p = "/opt/rocm/bin/hipcc"
    if os.path.exists(p):
        return p
    raise RuntimeError("hipcc not found: libsco_hip.so cannot be built")


def _newest_header():
    return max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS if os.path.exists(os.path.join(CSRC, h)))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False, defines=(), lib=None, obj_dir=None):
    """Compile every HIP source into csrc/libsco_hip.so (gfx950 code object).

    `defines` / `lib` / `obj_dir` build a diagnostic variant (e.g. -DSCO_STAMP) next to the product."""
    lib = lib or LIB
    if obj_dir is None:
        # a variant with its own defines gets its own objects: it must neither reuse the product's (compiled without
        # the defines) nor overwrite them
        obj_dir = OBJ if not defines else os.path.join(CSRC, "build_" + "_".join(sorted(d.replace("=", "-") for d in defines)))
    if not force and lib == LIB and not needs_build():
        return LIB
    os.makedirs(obj_dir, exist_ok=True)
    hipcc = hipcc_path()
    hdr_t = _newest_header()
    flags = FLAGS + ["-D" + d for d in defines]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        if not os.path.exists(src):
            continue
        o = os.path.join(obj_dir, os.path.splitext(s)[0] + ".o")
        objs.append(o)
        if force or not os.path.exists(o) or os.path.getmtime(o) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + flags + ["-c", src, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        subprocess.check_call(cmd, cwd=CSRC)

    workers = max(1, min(len(jobs), (os.cpu_count() or 2)))
    if jobs:
        with ThreadPoolExecutor(workers) as ex:
            list(ex.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force=True, verbose=True))
