#!/bin/bash
cd $GRAFT_REPO_ROOT
for pair in 0 16; do for us in 5 10 15 20 30; do
  t=$(( 2 + pair + us * 256 ))
  echo "== pairing bit $pair hold $us us (PG_DGEMM_TUNE=$t)"
  PG_DGEMM_TUNE=$t timeout -k 10 100 python tools/bench_dgemm_ex.py 20 "9984" 2>&1 | grep -v symX
  PG_DGEMM_TUNE=$t timeout -k 10 100 python tools/bench_dgemm_ex.py 20 "K=256" 2>&1 | head -1
done; done 2>&1 | tee gpurun_out/stagger_sweep.log
