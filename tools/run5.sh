#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4e
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_syevd.py -q -x -k "dgemm" > gpurun_out/r4e/t_dgemm.log 2>&1; rc=$?; echo "dgemm tests rc=$rc"; tail -n 5 gpurun_out/r4e/t_dgemm.log
[ $rc -eq 0 ] || exit 1
for t in 0 4; do
  echo "== PG_DGEMM_TUNE=$t"
  PG_DGEMM_TUNE=$t timeout -k 10 300 python tools/bench_dgemm_ex.py 20 2>&1 | head -14 | tee gpurun_out/r4e/tune$t.log
done
PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 > gpurun_out/r4e/syevd10000.log 2>&1; echo "syevd10000 rc=$?"; grep -v stedc gpurun_out/r4e/syevd10000.log | tail -n 10
