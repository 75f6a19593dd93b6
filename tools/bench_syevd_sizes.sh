#!/bin/bash
# pg_syevd_dev over a list of sizes, both reductions: wall time of the second call + phase timers of the two-stage path.
# usage (through gpurun, repo root): bash tools/bench_syevd_sizes.sh 2000 4096 10000 20000
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for n in "$@"; do
  for st in 1 2; do
    if [ $st = 1 ] && [ $n -gt 12000 ]; then continue; fi
    PG_SYEVD_STAGES=$st PG_SYEVD_TIMING=1 timeout -k 10 400 python3 $ROOT/tools/bench_syevd.py $n 2>&1 | grep -v "\[stedc" | awk -v n=$n -v st=$st '
      /^syevd.*run 0/ { t0 = $3; ph = ""; al = "" }
      /allocate/ { al = $(NF-1) }
      /dense->band|band->tridiag|divide&conquer|back-transform|tridiagonalise/ { ph = ph " " $(NF-2) "=" $(NF-1) }
      /^syevd.*run 1/ { t = $3 }
      END { printf "n=%d stages=%d: %s s (first call %s s; hipMalloc of the work space in the second call %s ms)  [%s ]\n", n, st, t, t0, al, ph }'
  done
done
