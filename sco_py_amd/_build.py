"""Build libsco_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU)."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libsco_hip.so")
SOURCES = ["sco_qp.hip", "sco_admm_fast.hip", "sco_admm_reg.hip", "sco_admm_rl.hip", "sco_qp_big.hip", "sco_sqp.hip", "qp_plan.cpp"]
HEADERS = ["sco_internal.h", "qp_plan.h", os.path.join("..", "..", "include", "sco_hip.h")]


def hipcc_path():
    p = shutil.which("hipcc")
    if p:
        return p
    p = "/opt/rocm/bin/hipcc"
    if os.path.exists(p):
        return p
    raise RuntimeError("hipcc not found: libsco_hip.so cannot be built")


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    for f in SOURCES + HEADERS:
        p = os.path.join(CSRC, f)
        if os.path.exists(p) and os.path.getmtime(p) > t:
            return True
    return False


def build(force=False, verbose=False):
    """Compile every HIP source into csrc/libsco_hip.so (gfx950 code object)."""
    if not force and not needs_build():
        return LIB
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
           "-o", LIB] + srcs
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
