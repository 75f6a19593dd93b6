#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4g
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_syevd.py -q -x > gpurun_out/r4g/t_syevd.log 2>&1; rc=$?; echo "syevd tests rc=$rc"; tail -n 8 gpurun_out/r4g/t_syevd.log
[ $rc -eq 0 ] || exit 1
for f in 0 1; do
PG_SB2_FUSED=$f PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 check > gpurun_out/r4g/syevd10000_f$f.log 2>&1; echo "syevd10000 fused=$f rc=$?"; grep -v stedc gpurun_out/r4g/syevd10000_f$f.log | tail -n 12
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r4g/prof -o p -- python3 $GRAFT_REPO_ROOT/tools/bench_syevd.py 10000 > $GRAFT_REPO_ROOT/gpurun_out/r4g/prof.log 2>&1; echo "prof rc=$?"
