#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
timeout -k 10 600 python -m pytest tests/test_gpu_syevd.py -q -x -k "two_stage or syevd_invariants or structured or stationary or back_transformation" > gpurun_out/r4p/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 5 gpurun_out/r4p/t.log
[ $rc -eq 0 ] || exit 1
for v in 0 1; do PG_BC_V2=$v PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 check 2>&1 | grep -E "band->tridiag|syevd n|orth|eig err" | tr "\n" " "; echo " <- BC_V2=$v"; done
PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 20000 2>&1 | grep -E "band->tridiag|syevd n" | tr "\n" " "; echo " <- n=20000"
