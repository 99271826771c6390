#!/bin/bash
# Diagnostic builds: what does one termination test of the row-local kernel cost, and which loads dominate it?
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd $R/sco_py_amd/csrc
for v in 1 2 3; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSCO_CHK_EXP=$v -o /tmp/libsco_exp$v.so sco_qp.hip sco_admm_fast.hip sco_admm_reg.hip sco_admm_rl.hip sco_qp_big.hip sco_sqp.hip qp_plan.cpp
done
cd $R
python3 scripts/gpu_iter_time.py
for v in 1 2 3; do SCO_LIB_OVERRIDE=/tmp/libsco_exp$v.so python3 scripts/gpu_iter_time.py; done
