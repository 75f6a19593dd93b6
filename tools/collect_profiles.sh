#!/bin/bash
# Collect the round's judged measurements on the GPU box (run through gpurun from the repo root):
#   the bench line, rocprofv3 --kernel-trace --stats of the SAME bench command, and PMC passes (one counter group per run) ON bench.py
#   ITSELF (the eigensolver is skipped in those through --eigh-cache: counter mode + its dispatch depth, profiles/r02_pmc_abort_diagnosis.txt;
#   one pass also runs WITH the eigensolver (the library drains the stream once per panel) to show that bench.py is profilable end to end).
# Outputs under gpurun_out/prof_${TAG}_final/ ; tools/summarize_prof.py condenses them into profiles/.
set -o pipefail
TAG=${1:-r04}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_${TAG}_final
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 python3 $ROOT/bench.py > $OUT/bench.json 2> $OUT/bench.err || { echo "bench failed"; tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $ROOT/bench.py --e2e 0 --cpu-sample 0 > $OUT/bench_under_rocprof.json 2> $OUT/stats.err || { echo "stats failed"; tail -5 $OUT/stats.err; exit 1; }
echo "stats done"
CACHE=$OUT/eigh_cache.npz
timeout -k 10 300 python3 $ROOT/bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 --eigh-cache $CACHE > /dev/null 2> $OUT/cache.err || { echo "cache run failed"; exit 1; }
pass() {  # name counters...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -o c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --e2e 0 --cpu-sample 0 --eigh-cache $CACHE > $OUT/$name.json 2> $OUT/$name.log
  local rc=$?
  echo "$name rc=$rc"
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq_valu SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq_mfma SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES
# bench.py WITH the eigensolver under the counter mode (the library bounds the queue depth itself since r3)
timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv -d $OUT/pmc_full -o c -- python3 $ROOT/bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 > $OUT/pmc_full.json 2> $OUT/pmc_full.log
echo "pmc_full rc=$?"
rm -f $CACHE
find $OUT -name "*.db" -delete
ls -la $OUT | head -40
