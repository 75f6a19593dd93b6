#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4c
cd $GRAFT_REPO_ROOT
for t in 0 1 2 3; do
  echo "== PG_DGEMM_TUNE=$t"
  PG_DGEMM_TUNE=$t timeout -k 10 300 python tools/bench_dgemm_ex.py 20 2>&1 | head -14 | tee gpurun_out/r4c/tune$t.log
done
echo "== KSPLIT_FILL=1 tune=0"
PG_DGEMM_KSPLIT_FILL=1 timeout -k 10 300 python tools/bench_dgemm_ex.py 20 2>&1 | grep symX | tee gpurun_out/r4c/ksfill.log
