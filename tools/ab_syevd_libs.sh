#!/bin/bash
# one two-stage solve at n = 10 000 (phase timers) under alternative builds of the library: usage ab_syevd_libs.sh name=path ...   ("base" = shipped)
cd $GRAFT_REPO_ROOT
for kv in "$@"; do
  name=${kv%%=*}; path=${kv#*=}
  if [ "$path" = base ]; then unset PYGEMMA_HIP_LIB; else export PYGEMMA_HIP_LIB=$GRAFT_REPO_ROOT/$path; fi
  echo "== $name"; PG_SYEVD_TIMING=1 timeout -k 10 200 python3 tools/bench_syevd.py 10000 check 2>&1 | grep -E "dense->band|band->tridiag|back-transform|syevd n=|orth" | tail -n 7
done
