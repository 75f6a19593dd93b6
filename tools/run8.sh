#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4h
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_syevd.py -q -x -k "dgemm or two_stage or syevd_invariants" > gpurun_out/r4h/t_syevd.log 2>&1; rc=$?; echo "syevd tests rc=$rc"; tail -n 4 gpurun_out/r4h/t_syevd.log
[ $rc -eq 0 ] || exit 1
for la in off -1 4 8 16; do
  if [ $la = off ]; then unset PG_SB2_LOOKAHEAD; else export PG_SB2_LOOKAHEAD=$la; fi
  PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 check > gpurun_out/r4h/la_$la.log 2>&1; echo "lookahead=$la rc=$?"; grep -E "dense|syevd n|orth|eig err" gpurun_out/r4h/la_$la.log
done
unset PG_SB2_LOOKAHEAD
PG_SB2_LOOKAHEAD=-1 timeout -k 10 900 python -m pytest tests/test_gpu_syevd.py -q -x -k "two_stage or syevd_invariants or structured" > gpurun_out/r4h/t_syevd_la.log 2>&1; echo "tests with lookahead rc=$?"; tail -n 4 gpurun_out/r4h/t_syevd_la.log
