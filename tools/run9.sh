#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4i
cd $GRAFT_REPO_ROOT
export PG_SB2_LOOKAHEAD=-1
timeout -k 10 900 python -m pytest tests/test_gpu_syevd.py -q -x -k "dgemm or two_stage or syevd_invariants or structured" > gpurun_out/r4i/t_syevd.log 2>&1; rc=$?; echo "syevd tests (lookahead) rc=$rc"; tail -n 4 gpurun_out/r4i/t_syevd.log
[ $rc -eq 0 ] || exit 1
for la in off -1; do
  if [ $la = off ]; then unset PG_SB2_LOOKAHEAD; else export PG_SB2_LOOKAHEAD=$la; fi
  PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 check > gpurun_out/r4i/la_$la.log 2>&1; echo "lookahead=$la rc=$?"; grep -E "dense|syevd n|orth|eig err" gpurun_out/r4i/la_$la.log
done
export PG_SB2_LOOKAHEAD=-1
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4i/prof_la -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_syevd.py 10000 > $GRAFT_REPO_ROOT/gpurun_out/r4i/prof_la.log 2>&1; echo prof rc=$?
