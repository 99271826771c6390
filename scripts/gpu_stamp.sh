#!/bin/bash
# Diagnostic build (-DSCO_STAMP): where does an ADMM iteration spend its cycles?
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd $R/sco_py_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSCO_STAMP -o /tmp/libsco_stamp.so sco_qp.hip sco_admm_fast.hip sco_admm_reg.hip sco_admm_rl.hip sco_qp_big.hip sco_sqp.hip qp_plan.cpp
cd $R
SCO_LIB_OVERRIDE=/tmp/libsco_stamp.so python3 scripts/stamp_run.py | tee $R/gpurun_out/stamps.txt
