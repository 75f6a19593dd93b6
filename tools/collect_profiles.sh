#!/bin/bash
# Collect the round's judged measurements on the GPU box (run through gpurun from the repo root):
#   bench line, rocprofv3 kernel stats of the same bench command, PMC passes (one counter group per run) on the
#   per-kernel drivers at the bench's shapes.  Outputs under gpurun_out/prof_${TAG}_final/ ; tools/summarize_prof.py condenses them.
set -e -o pipefail
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_${TAG}_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err
echo "bench done"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $ROOT/bench.py --steps 10 --warmup 2 --cpu-sample 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
echo "stats done"
pass() {  # name counters... -- driver args
  local name=$1; shift
  local ctr=(); while [ "$1" != "--" ]; do ctr+=("$1"); shift; done; shift
  rocprofv3 --kernel-trace --pmc "${ctr[@]}" --output-format csv -d $OUT/$name -o c -- python3 "$@" > $OUT/$name.log 2>&1
  echo "$name done"
}
pass fetch_geno FETCH_SIZE -- $ROOT/tools/bench_rotate_geno.py 10000 16384
pass write_geno WRITE_SIZE -- $ROOT/tools/bench_rotate_geno.py 10000 16384
pass sq_geno SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY -- $ROOT/tools/bench_rotate_geno.py 10000 16384
pass fetch_assoc FETCH_SIZE -- $ROOT/tools/bench_assoc.py 10000 16384 5
pass write_assoc WRITE_SIZE -- $ROOT/tools/bench_assoc.py 10000 16384 5
pass sq_assoc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_assoc.py 10000 16384 5
pass fetch_rot FETCH_SIZE -- $ROOT/tools/bench_rotate.py 10000 16384
pass write_rot WRITE_SIZE -- $ROOT/tools/bench_rotate.py 10000 16384
pass sq_rot SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_rotate.py 10000 16384
pass sq_dgemm_panel SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_dgemm.py 8192 8192 128 1
pass sq_dgemm_big SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- $ROOT/tools/bench_dgemm.py 8192 8192 8192
find $OUT -name "*.db" -delete
ls -R $OUT | head -60
