#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4b
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_syevd.py -q -x -k "dgemm" > gpurun_out/r4b/t_dgemm.log 2>&1; rc=$?; echo "dgemm tests rc=$rc"; tail -n 15 gpurun_out/r4b/t_dgemm.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 600 python tools/bench_dgemm_ex.py 20 > gpurun_out/r4b/dgemm_ex.log 2>&1; echo "dgemm_ex rc=$?"; cat gpurun_out/r4b/dgemm_ex.log
PG_DGEMM_RING=0 timeout -k 10 600 python tools/bench_dgemm_ex.py 20 > gpurun_out/r4b/dgemm_ex_old.log 2>&1; echo "old rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_syevd.py -q -x > gpurun_out/r4b/t_syevd.log 2>&1; echo "syevd tests rc=$?"; tail -n 5 gpurun_out/r4b/t_syevd.log
PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 check > gpurun_out/r4b/syevd10000.log 2>&1; echo "syevd10000 rc=$?"; grep -v stedc gpurun_out/r4b/syevd10000.log
