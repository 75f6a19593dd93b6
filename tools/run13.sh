#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4o
timeout -k 10 600 python -m pytest tests/test_gpu_syevd.py -q -x -k "two_stage or syevd_invariants or structured or rank_deficient" > gpurun_out/r4o/t.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -n 5 gpurun_out/r4o/t.log
[ $rc -eq 0 ] || exit 1
for c in 0 1; do PG_SB2_CHOL16=$c PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 check 2>&1 | grep -E "dense|syevd n|orth|eig err" | tr "\n" " "; echo " <- CHOL16=$c"; done
cd /tmp && export TMPDIR=/tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4o/prof -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_syevd.py 10000 > $GRAFT_REPO_ROOT/gpurun_out/r4o/prof.log 2>&1; grep -E "chol_inv|recon" $GRAFT_REPO_ROOT/gpurun_out/r4o/prof/p_kernel_stats.csv | cut -c1-150
