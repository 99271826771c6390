#!/bin/bash
# tests -> bench -> rocprofv3 kernel stats of the same bench command (run on the GPU box)
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tee $R/gpurun_out/pytest_gpu.log
python bench.py ${BENCH_ARGS:-} 2>$R/gpurun_out/bench.err | tee $R/gpurun_out/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof -o bench -- python3 $R/bench.py ${BENCH_ARGS:-} --cpu-problems 0 --aux-b4096 0 > $R/gpurun_out/bench_prof.json 2>$R/gpurun_out/prof.err
ls -R $R/gpurun_out/prof | head -30
