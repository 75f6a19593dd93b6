#!/bin/bash
# Second tier of the round's profiles: kernels beside the bench step — the eigensolver's GEMM shapes, the kinship syrk, the LRT
# instantiation of the association kernel, the eigensolver's own kernel table.  Outputs under gpurun_out/prof_${TAG}_extra/.
TAG=${1:-r02}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_${TAG}_extra
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
pass() {  # name counters... -- script args
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $OUT/$name -o c -- python3 "$@" > $OUT/$name.log 2>&1
  local rc=$?; echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
}
pass dgemm_panel SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_dgemm.py 8192 8192 128 1
pass dgemm_big SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_dgemm.py 8192 8192 8192
pass kinship SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_kinship.py 10000 20000
pass lrt SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_lrt.py 10000 16384 5
pass rot32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_rotate.py 10000 16384
pass rot32_fetch FETCH_SIZE -- $ROOT/tools/bench_rotate.py 10000 16384
pass rot32_write WRITE_SIZE -- $ROOT/tools/bench_rotate.py 10000 16384
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/syevd_stats -o k -- python3 $ROOT/tools/bench_syevd.py 10000 > $OUT/syevd_stats.log 2>&1; echo "syevd stats rc=$?"
find $OUT -name "*.db" -delete
find $OUT -name "*kernel_trace.csv" -size +3M -delete
ls $OUT
