#!/bin/bash
cd $GRAFT_REPO_ROOT
for us in 0 6 10 13 16 20 26; do
  t=$(( (us > 0 ? 2 : 0) + us * 256 ))
  echo "== hold $us us (PG_DGEMM_TUNE=$t)"
  PG_DGEMM_TUNE=$t timeout -k 10 100 python tools/bench_dgemm_ex.py 20 "update" 2>&1 | head -2
  PG_DGEMM_TUNE=$t timeout -k 10 100 python tools/bench_dgemm_ex.py 20 "K=256" 2>&1 | head -1
done 2>&1 | tee gpurun_out/stagger_cu.log
