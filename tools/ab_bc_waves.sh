#!/bin/bash
# stationary bulge chasing with 16 / 8 / 4 wavefronts per workgroup (PG_BC_STAT_WAVES, one workgroup per CU): stage-2 time of one solve at n = 10 000
cd $GRAFT_REPO_ROOT
for w in 16 8 4; do echo "== waves $w"; PG_BC_STAT_WAVES=$w PG_SYEVD_TIMING=1 timeout -k 10 200 python3 tools/bench_syevd.py 10000 check 2>&1 | grep -E "band->tridiag|syevd n=|orth" | tail -n 4; done
