#!/bin/bash
# What the association kernel's wavefronts do with their cycles at the bench shape: instruction mix and stall reasons (two rocprofv3 --pmc passes on tools/bench_assoc.py)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_issue; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/a -o c -- python3 $ROOT/tools/bench_assoc.py 10000 16384 5 > $OUT/a.log 2>&1; echo a rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/b -o c -- python3 $ROOT/tools/bench_assoc.py 10000 16384 5 > $OUT/b.log 2>&1; echo b rc=$?
python3 - <<PY
import csv, collections
for d in "ab":
    rows = list(csv.DictReader(open("$OUT/%s/c_counter_collection.csv" % d)))
    acc = collections.defaultdict(list)
    for r in rows:
        if "assoc_kernel" in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items(): print(d, "%-28s %.4g" % (k, sum(v) / len(v)))
PY
find $OUT -name "*.db" -delete
