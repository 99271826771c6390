#!/bin/bash
# HBM traffic of the ADMM kernel from PMC counters: FETCH_SIZE and WRITE_SIZE in
# separate passes (they do not fit one TCC pass), kernel-trace only (no sys-trace).
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$C -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-problems 0 > $R/gpurun_out/pmc_$C.json 2> $R/gpurun_out/pmc_$C.err
  ls $R/gpurun_out/pmc_$C | head
done
python3 $R/scripts/pmc_summary.py $R/gpurun_out | tee $R/gpurun_out/pmc_summary.txt
