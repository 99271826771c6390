#!/bin/bash
# r04 parity sweeps: device loop (wavefront tier + row-local tail by default) vs oracle, fresh problem ranges (run on the GPU box)
set -e
cd ${GRAFT_REPO_ROOT:-$(pwd)}
O=gpurun_out/r04_parity_sweep.txt; : > $O
run() { python scripts/gpu_parity_sweep.py "$@" 2>&1 | tail -1 >> $O; }
runwv() { SCO_WV_MIN_PER_CU=0 python scripts/gpu_parity_sweep.py "$@" 2>&1 | tail -1 | sed 's/$/   [wavefront tier forced]/' >> $O; }
run 0 1024 parity
runwv 6000 256 intended
runwv 6000 256 parity point
run 6000 256 parity objw
run 6000 256 parity reach objw vel
run 6000 256 parity prog steps
run 6000 256 parity prog:sweep steps ajac
run 6000 256 parity prog:attract steps objw
run 6000 256 parity prog:accel
run 6000 256 parity prog:jerk ajac
run 6000 256 parity prog:dynamics ajac
run 6000 256 parity jl vel
run 6000 256 parity rows
run 6000 256 parity rows reach objw
run 6000 256 parity prog circles
run 6000 256 parity prog:attract circles steps rows
cat $O
