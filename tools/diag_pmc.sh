#!/bin/bash
# Diagnosis of the round-1 "rocprofv3 --pmc aborts on bench.py" report.  Finding (profiles/r02_pmc_abort_diagnosis.txt):
# the SIGSEGV is inside librocprofiler-sdk's hsa intercept, reached from a hipLaunchKernel of the tridiagonalisation loop, once
# ~10^4 dispatches are in flight without a synchronisation; draining the stream once per panel avoids it — since round 3 the
# library does that for every caller.  By default this script runs the fixed configuration only (counter mode on the eigensolver
# at n = 10 000).  `diag_pmc.sh reproduce` first re-runs the known-crashing deep-queue configuration (PG_SYEVD_PANEL_SYNC=0): that
# deliberately SIGSEGVs a GPU process on a shared box — the committed diagnosis file is the evidence, do not run it in routine
# collection or soak scripts.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmcdiag
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
step() {  # name seconds cmd...
  local name=$1 secs=$2; shift 2
  echo "== $name" | tee -a $OUT/summary.txt
  timeout -k 10 $secs "$@" > $OUT/$name.log 2>&1
  local rc=$?
  echo "$name rc=$rc" | tee -a $OUT/summary.txt
  grep -v "^[0-9a-f]*-[0-9a-f]* r-xp" $OUT/$name.log | tail -n 12 >> $OUT/summary.txt
  if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping" | tee -a $OUT/summary.txt; exit 1; fi
  return 0
}
PMC="rocprofv3 --kernel-trace --pmc SQ_WAVES --output-format csv"
if [ "$1" = "reproduce" ]; then
  PG_SYEVD_PANEL_SYNC=0 step pmc_syevd10k_deep_queue 500 $PMC -d $OUT/pmc_syevd10k -o c -- python3 $ROOT/tools/diag_pmc2.py syevd 10000
fi
unset PG_SYEVD_PANEL_SYNC
step pmc_syevd10k_panel_sync 500 $PMC -d $OUT/pmc_syevd10k_sync -o c -- python3 $ROOT/tools/diag_pmc2.py syevd 10000
find $OUT -name "*.db" -delete
find $OUT -name "*.csv" -size +2M -delete
ls -laR $OUT | head -40
