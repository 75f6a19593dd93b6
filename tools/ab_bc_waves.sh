cd $GRAFT_REPO_ROOT
for w in 16 8 4; do echo "== waves $w"; PG_BC_STAT_WAVES=$w PG_SYEVD_TIMING=1 timeout -k 10 200 python3 tools/bench_syevd.py 10000 check 2>&1 | grep -E "band->tridiag|syevd n=|orth" | tail -n 4; done
