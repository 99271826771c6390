"""Diagnostic variants of the row-local ADMM kernel (-DRL_ABLATE=mask, see csrc/sco_admm_rl.hip): built here in
parallel into gpurun-visible files sco_py_amd/csrc/variants/libsco_ablate_<mask>.so; scripts/gpu_ablate.sh times them."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from sco_py_amd import _build

# arguments: ablation masks as integers; "vN" = -DRL_VARIANT=N; "name=path.hip" = an alternative kernel source;
# "wv:NAME=VAL[,NAME=VAL]" = the wavefront tier's source (sco_admm_wv.hip) with those defines (WV_PD, WV_VARIANT);
# "wvfile:TAG=path.hip" = an alternative source file in its place
masks = sys.argv[1:] or ["0", "1", "2", "4", "8", "12", "16", "31"]
_build.build()                                   # product objects are current
out = os.path.join(_build.CSRC, "variants")
os.makedirs(out, exist_ok=True)


def one(mask):
    src, defs, tag = os.path.join(_build.CSRC, "sco_admm_rl.hip"), [], mask
    base = "sco_admm_rl.hip"
    if mask.startswith("wvfile:"):               # "wvfile:TAG=path.hip": an alternative source of the wavefront tier
        base = "sco_admm_wv.hip"
        tag, src = mask[7:].split("=", 1)
        tag = "wvfile_" + tag
    elif mask.startswith("wv:"):
        base = "sco_admm_wv.hip"
        src = os.path.join(_build.CSRC, base)
        defs = ["-D" + d for d in mask[3:].split(",")]
        tag = "wv_" + mask[3:].replace("=", "").replace(",", "_")
    elif "=" in mask:
        tag, src = mask.split("=", 1)
    elif mask.startswith("v"):
        defs = ["-DRL_VARIANT=%d" % int(mask[1:])]
    else:
        defs = ["-DRL_ABLATE=%d" % int(mask)]
    mask = tag
    obj = os.path.join(out, "rl_%s.o" % mask)
    subprocess.check_call([_build.hipcc_path()] + _build.FLAGS + defs + ["-I", _build.CSRC, "-x", "hip", "-c", src, "-o", obj])
    objs = [os.path.join(_build.OBJ, os.path.splitext(s)[0] + ".o") for s in _build.SOURCES if s != base]
    lib = os.path.join(out, "libsco_ablate_%s.so" % mask)
    subprocess.check_call([_build.hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", "-o", lib, obj] + objs)
    os.remove(obj)
    return lib


with ThreadPoolExecutor(min(len(masks), os.cpu_count() or 2)) as ex:
    for lib in ex.map(one, masks):
        print(lib)
