#!/bin/bash
# PMC passes on one bench step (separate passes; kernel-trace only, no sys-trace).
# PMC_BENCH_ARGS adds bench.py options, e.g. "--workload 12x50 --batch 64".
set -e
R=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  name=$1; shift
  rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$name -o pmc -- python3 $R/bench.py --steps 1 --warmup 0 --cpu-problems 0 --aux-12x50 0 --aux-b4096 0 --aux-object-api 0 ${PMC_BENCH_ARGS:-} > $R/gpurun_out/pmc_$name.json 2> $R/gpurun_out/pmc_$name.err || { tail -5 $R/gpurun_out/pmc_$name.err; return 1; }
}
run FETCH_SIZE FETCH_SIZE
run WRITE_SIZE WRITE_SIZE
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS
run sq2 SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD GRBM_GUI_ACTIVE
python3 $R/scripts/pmc_summary.py $R/gpurun_out | tee $R/gpurun_out/pmc_summary.txt
