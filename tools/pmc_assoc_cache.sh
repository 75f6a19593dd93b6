#!/bin/bash
# L1 (TCP) / L2 (TCC) behaviour of the association kernel at the bench shape: two rocprofv3 --pmc passes on tools/bench_assoc.py
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/pmc_cache; mkdir -p $OUT; cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum --output-format csv -d $OUT/a -o c -- python3 $ROOT/tools/bench_assoc.py 10000 16384 5 > $OUT/a.log 2>&1; echo a rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_READ_sum --output-format csv -d $OUT/b -o c -- python3 $ROOT/tools/bench_assoc.py 10000 16384 5 > $OUT/b.log 2>&1; echo b rc=$?
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT/c -o c -- python3 $ROOT/tools/bench_assoc.py 10000 16384 5 > $OUT/c.log 2>&1; echo c rc=$?
python3 - <<PY
import csv, collections
for d in "abc":
    try:
        rows = list(csv.DictReader(open("$OUT/%s/c_counter_collection.csv" % d)))
    except Exception as ex:
        print(d, "no csv", ex); continue
    acc = collections.defaultdict(list)
    for r in rows:
        if "assoc_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(d, k, sum(v) / len(v), "n", len(v))
PY
