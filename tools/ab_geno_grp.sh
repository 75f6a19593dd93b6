#!/bin/bash
# A/B of rotate_geno_kernel's m-tile group size (PG_GENO_GRP): kernel time (--kernel-trace --stats) and FETCH_SIZE per variant.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_grp; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base g2 g8 g16; do
  if [ $v = base ]; then unset PYGEMMA_HIP_LIB; else export PYGEMMA_HIP_LIB=$ROOT/pygemma_amd/lib_dev/$v/libpygemma_hip.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$v -o k -- python3 $ROOT/tools/bench_rotate_geno.py 10000 16384 > $OUT/$v.log 2>&1 || { echo "fail $v"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${v}_f -o c -- python3 $ROOT/tools/bench_rotate_geno.py 10000 16384 > $OUT/${v}_f.log 2>&1 || { echo "fail pmc $v"; exit 1; }
  echo "== $v"; grep rotate_geno_kernel $OUT/$v/k_kernel_stats.csv | cut -d, -f1-6 | cut -c1-40,200-
  python3 - <<PY
import csv
rows=[r for r in csv.DictReader(open("$OUT/${v}_f/c_counter_collection.csv")) if "rotate_geno_kernel" in r["Kernel_Name"]]
v=[float(r["Counter_Value"]) for r in rows]; print("FETCH_SIZE KB avg", sum(v)/len(v), "n", len(v))
PY
done
find $OUT -name "*.db" -delete
