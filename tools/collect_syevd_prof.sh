#!/bin/bash
# Records of the two-stage eigensolver at n = 10 000 (run through gpurun from the repo root):
#   phase timers (PG_SYEVD_TIMING), rocprofv3 --kernel-trace of one solve, one --pmc pass (MFMA-busy cycles) on the same command.
# tools/summarize_syevd.py condenses them into profiles/<tag>_syevd_summary.json
TAG=${1:-r04}
N=${2:-10000}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_${TAG}_syevd
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
PG_SYEVD_TIMING=1 timeout -k 10 300 python3 $ROOT/tools/bench_syevd.py $N check 2>&1 | grep -v "\[stedc" > $OUT/timing.txt || { echo "timing run failed"; exit 1; }
tail -4 $OUT/timing.txt
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o t -- python3 $ROOT/tools/bench_syevd.py $N > $OUT/trace.log 2>&1 || { echo "trace failed"; exit 1; }
echo "trace done"
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc -o c -- python3 $ROOT/tools/bench_syevd.py $N > $OUT/pmc.log 2>&1
echo "pmc rc=$?"
python3 $ROOT/tools/summarize_syevd.py $TAG $N > $OUT/summary_head.txt && cp $ROOT/profiles/${TAG}_syevd_summary.json $OUT/   # profiles/ does not travel back, gpurun_out/ does
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +3M -delete
