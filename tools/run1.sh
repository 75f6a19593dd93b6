#!/bin/bash
# round-4 GPU call 1: new tests, n = 1940 probe, baselines
set -o pipefail
mkdir -p gpurun_out/r4a
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests/test_gpu_lmm.py -q -x -k "design_matrix" > gpurun_out/r4a/t_z.log 2>&1; echo "Z test rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -q -x -k "real_reference" > gpurun_out/r4a/t_tiera.log 2>&1; echo "tierA rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_stream.py -q -x -s -k "config4_one_gpu" > gpurun_out/r4a/t_cfg3.log 2>&1; echo "cfg3 rc=$?"
timeout -k 10 900 python -m pytest tests/test_gpu_syevd.py -q -x -k "two_stage or dgemm or last_device" > gpurun_out/r4a/t_syevd.log 2>&1; echo "syevd rc=$?"
PG_SYEVD_TIMING=1 PG_SYEVD_STAGES=2 timeout -k 10 300 python tools/probe_sizes.py 1940 1940 1940 1920 1920 1984 1984 1940 2000 2000 1940 > gpurun_out/r4a/probe_sizes.log 2>&1; echo "probe rc=$?"
PG_SYEVD_TIMING=1 timeout -k 10 300 python tools/bench_syevd.py 10000 > gpurun_out/r4a/syevd10000.log 2>&1; echo "syevd10000 rc=$?"
timeout -k 10 600 python tools/bench_dgemm_ex.py 20 > gpurun_out/r4a/dgemm_ex.log 2>&1; echo "dgemm_ex rc=$?"
timeout -k 10 120 tools/probe_event_pingpong.bin > gpurun_out/r4a/pingpong.log 2>&1; echo "pingpong rc=$?"
cd /tmp && export TMPDIR=/tmp
PG_SYEVD_STAGES=2 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4a/prof1940 -o p1940 -- python3 $GRAFT_REPO_ROOT/tools/probe_sizes.py 1940 1940 1940 > $GRAFT_REPO_ROOT/gpurun_out/r4a/prof1940.log 2>&1; echo "prof rc=$?"
