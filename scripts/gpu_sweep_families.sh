set -e
cd $GRAFT_REPO_ROOT
O=gpurun_out/sweep_final.txt; : > $O
run() { python scripts/gpu_parity_sweep.py "$@" 2>&1 | tail -1 >> $O; }
run 5000 256 parity point
run 5000 256 parity quad
run 5000 256 parity reach
run 5000 256 parity vel
run 5000 256 parity jl
run 5000 256 parity prog
run 5000 256 parity prog:sweep
run 5000 256 parity prog:dynamics
run 5000 256 parity prog:curve
run 5000 256 parity prog:attract
run 5000 256 parity prog:sweep ajac
run 5000 128 intended
cat $O
