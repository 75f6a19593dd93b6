#!/bin/bash
# LDS bank conflicts per kernel of one two-stage solve at n = 10 000: rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE (cycles summed over the CUs' SQs)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_lds; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS GRBM_GUI_ACTIVE --output-format csv -d $OUT/p -o c -- python3 $ROOT/tools/bench_syevd.py 10000 > $OUT/log.txt 2>&1
python3 - <<PY
import csv, glob, collections
f = glob.glob("$OUT/p/**/c_counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); dur = collections.defaultdict(float); calls = collections.defaultdict(set)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    calls[k].add(r["Dispatch_Id"])
for r in csv.DictReader(open(glob.glob("$OUT/p/**/c_kernel_trace.csv", recursive=True)[0])):
    dur[r["Kernel_Name"][:60]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
print("%-60s %8s %6s %12s %12s %8s" % ("kernel (both solves of the run)", "ms", "calls", "LDS active", "bank confl", "confl/act"))
for k, ms in sorted(dur.items(), key=lambda kv: -kv[1])[:16]:
    a = acc[k]
    print("%-60s %8.2f %6d %12.3g %12.3g %8.2f" % (k, ms, len(calls[k]), a["SQ_LDS_IDX_ACTIVE"], a["SQ_LDS_BANK_CONFLICT"], a["SQ_LDS_BANK_CONFLICT"] / max(a["SQ_LDS_IDX_ACTIVE"], 1)))
PY
find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +3M -delete
