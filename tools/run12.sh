#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
for la in off -1 -2; do
  if [ $la = off ]; then unset PG_SB2_LOOKAHEAD; else export PG_SB2_LOOKAHEAD=$la; fi
  PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 check > gpurun_out/r4n/la_$la.log 2>&1; echo "lookahead=$la rc=$?"; grep -E "dense|syevd n|orth" gpurun_out/r4n/la_$la.log
done
export PG_SB2_LOOKAHEAD=-2
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4n/prof_la -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_syevd.py 10000 > $GRAFT_REPO_ROOT/gpurun_out/r4n/prof_la.log 2>&1; echo prof rc=$?
