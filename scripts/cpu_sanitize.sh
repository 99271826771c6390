#!/bin/bash
# CPU sanitizer job (SURVEY 5 "race detection / sanitizers"; r02 verdict item 8): the two pieces of hand-rolled host
# linear algebra and index planning -- csrc/qp_plan.cpp (symbolic plans: elimination set, Schur pattern, CSR / CSC index
# maps) and oracle/osqp_ref.c (sparse LDL' with minimum-degree ordering, Ruiz scaling, ADMM) -- built with
# AddressSanitizer + UndefinedBehaviorSanitizer and driven by the existing tests:
#   tests/test_qp_plan.py   (plan tests that need no HIP: the plan entry points come from scripts/sanitize/plan_driver.cpp)
#   tests/test_oracle_osqp.py, tests/test_golden.py   (the oracle under the reference's golden vectors)
#   tests/test_adjudicate.py   (r04: the x87 extended-precision build of the same file, oracle/osqp_ref_ld.c)
# CPU only -- never on the GPU box (GPU sanitizers are not available on this pool).  Usage: bash scripts/cpu_sanitize.sh
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-$R/gpurun_out}
mkdir -p $OUT
ASAN=$(gcc -print-file-name=libasan.so)
UBSAN=$(gcc -print-file-name=libubsan.so)
SAN="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=undefined"
g++ $SAN -std=c++17 -shared -fPIC -o $R/scripts/sanitize/libplan_asan.so $R/sco_py_amd/csrc/qp_plan.cpp $R/scripts/sanitize/plan_driver.cpp
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
export SCO_ORACLE_SANITIZE=1
cd $R
# the plan tests: sco_py_amd._lib.load is pointed at the sanitized host library for this process only
LD_PRELOAD="$ASAN $UBSAN" python3 - <<'PY' 2>&1 | tee $OUT/sanitize_plan.log
import ctypes, os, sys
import pytest
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from sco_py_amd import _lib
lib = ctypes.CDLL(os.path.join(os.getcwd(), "scripts", "sanitize", "libplan_asan.so"))
_lib.load = lambda: lib
sys.exit(pytest.main(["tests/test_qp_plan.py", "-x", "-q", "-p", "no:cacheprovider",
                      "-k", "plans_reproduce or eliminated_set or malformed"]))
PY
LD_PRELOAD="$ASAN $UBSAN" python3 -m pytest tests/test_oracle_osqp.py tests/test_golden.py tests/test_adjudicate.py -x -q -p no:cacheprovider -m "not gpu" 2>&1 | tee $OUT/sanitize_oracle.log
echo "sanitizer job finished: no AddressSanitizer / UndefinedBehaviorSanitizer report"
