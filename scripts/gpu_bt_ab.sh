#!/bin/bash
# Same-device A/B of the structured global-memory kernel (12-DOF x 50): variant libraries (LIBS: names under
# sco_py_amd/csrc, e.g. libsco_old.so built by hand from `git archive HEAD`) against the working tree's library
# (with SCO_QP_BT_TRIPLE=0 and as it is), alternating, two passes.  Diagnostic only.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/bt_ab.txt
mkdir -p $R/gpurun_out; : > $O
cd $R
export BS=${BS:-1,128,256}
for rep in 1 2; do
  for lib in ${LIBS:-libsco_old.so}; do
    echo "== $lib (pass $rep)" >> $O
    SCO_LIB_OVERRIDE=$R/sco_py_amd/csrc/$lib timeout -k 10 200 python3 scripts/gpu_bt_iter_time.py >> $O 2>&1 || exit 1
  done
  echo "== libsco_hip.so, SCO_QP_BT_TRIPLE=0 (pass $rep)" >> $O
  SCO_QP_BT_TRIPLE=0 timeout -k 10 200 python3 scripts/gpu_bt_iter_time.py >> $O 2>&1 || exit 1
  echo "== libsco_hip.so (pass $rep)" >> $O
  timeout -k 10 200 python3 scripts/gpu_bt_iter_time.py >> $O 2>&1 || exit 1
done
grep -v "^library" $O
