#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r4k
cd $GRAFT_REPO_ROOT
PG_DGEMM_WS=2 timeout -k 10 300 python -m pytest tests/test_gpu_syevd.py -q -x -k "dgemm" > gpurun_out/r4k/t_dgemm_ws.log 2>&1; rc=$?; echo "dgemm tests WS=2 rc=$rc"; tail -n 25 gpurun_out/r4k/t_dgemm_ws.log
python - <<'PY'
import ctypes as C, sys
sys.path.insert(0,'.')
from pygemma_amd import _lib
L=_lib.load(); v=C.c_int(-1); L.pgx_ws_aborts.argtypes=[C.c_void_p]; print("rc",L.pgx_ws_aborts(C.byref(v)),"aborts",v.value)
PY
[ $rc -eq 0 ] || exit 1
for wsm in 1 0; do echo "== PG_DGEMM_WS=$wsm"; PG_DGEMM_WS=$wsm timeout -k 10 200 python tools/bench_dgemm_ex.py 20 2>&1 | head -14; done | tee gpurun_out/r4k/bench.log
