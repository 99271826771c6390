#!/bin/bash
# per-iteration time of every diagnostic variant built by scripts/build_ablate.py (run on the GPU box)
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
: > $R/gpurun_out/ablate.txt
for f in $(ls $R/sco_py_amd/csrc/variants/libsco_ablate_*.so | sort); do
  echo "== $f" >> $R/gpurun_out/ablate.txt
  SCO_LIB_OVERRIDE=$f CHECK=${CHECK:-25} timeout -k 10 120 python3 $R/scripts/gpu_iter_time.py >> $R/gpurun_out/ablate.txt 2>&1 || exit 1
done
cat $R/gpurun_out/ablate.txt
