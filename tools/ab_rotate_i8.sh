#!/bin/bash
# int8 3-digit rotation (rotate_geno_i8_kernel, PG_GENO_I8=1) against the fp16 x 2 kernel (PG_GENO_I8=0) inside the bench step, under PMC passes:
# matrix-pipe busy and issue stalls; LDS activity; L2 hits / misses; fabric fetches.  Only the launches of the bench's 100 000-SNP step are counted (duration >= half of the kernel's longest).
# Output: gpurun_out/ab_i8/summary.txt (copied to profiles/r04_rotate_i8_ab.txt)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/ab_i8; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C=$OUT/ec.npz
python3 $ROOT/bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 --eigh-cache $C > /dev/null 2>&1
: > $OUT/summary.txt
for v in 0 1; do
  PG_GENO_I8=$v timeout -k 10 300 python3 $ROOT/bench.py --e2e 0 --cpu-sample 0 --eigh-cache $C > $OUT/bench_i8_$v.json 2> $OUT/bench_i8_$v.err || { echo "bench failed"; exit 1; }
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    PG_GENO_I8=$v timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/v${v}_$i -o c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --e2e 0 --cpu-sample 0 --eigh-cache $C > $OUT/v${v}_$i.json 2> $OUT/v${v}_$i.log
    rc=$?; [ $rc -eq 124 ] && { echo "timeout"; exit 1; }
  done
  python3 - >> $OUT/summary.txt <<PY
import csv, collections, glob, json
name = "rotate_geno_i8_kernel" if $v else "rotate_geno_kernel"
acc = collections.defaultdict(float); dur = []
for i in (1, 2, 3, 4):
    fs = glob.glob("$OUT/v${v}_%d/**/c_counter_collection.csv" % i, recursive=True)
    if not fs: continue
    rows = [r for r in csv.DictReader(open(fs[0])) if name in r["Kernel_Name"]]
    dmax = max(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    rows = [r for r in rows if int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) >= 0.5 * dmax]      # the 100 000-SNP launches only
    nl = len({r["Dispatch_Id"] for r in rows})
    for r in rows:
        key = r["Counter_Name"] if r["Counter_Name"] not in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES") else r["Counter_Name"] + "_%d" % i
        acc[key] += float(r["Counter_Value"]) / max(nl, 1)
        if i == 1 and r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
ms = sum(dur) / len(dur)
cyc = acc["GRBM_GUI_ACTIVE_1"] / 8.0
b = json.loads(open("$OUT/bench_i8_$v.json").read().strip().splitlines()[-1])
print("PG_GENO_I8=$v (%s): bench step %.2f ms = %.0f SNPs/s (no profiler); under --pmc, per launch of 100 000 SNPs at n = 10 000:" % (name, b["ms_per_step"], b["value"]))
print("  duration %.2f ms, clock %.2f GHz; matrix pipe busy %.3f; of the wave-cycles: waiting on memory/barriers (SQ_WAIT_ANY) %.2f, issue-stalled (SQ_WAIT_INST_ANY) %.2f, issuing %.2f" % (
    ms, cyc / (ms * 1e6), acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), acc["SQ_WAIT_ANY"] / acc["SQ_WAVE_CYCLES_1"], acc["SQ_WAIT_INST_ANY"] / acc["SQ_WAVE_CYCLES_1"], acc["SQ_ACTIVE_INST_ANY"] / acc["SQ_WAVE_CYCLES_1"]))
print("  LDS: %.3g instructions, active %.3f and issue-stalled %.3f of the wave-cycles, bank-conflict cycles %.3g" % (
    acc["SQ_INSTS_LDS"], acc["SQ_ACTIVE_INST_LDS"] / max(acc["SQ_WAVE_CYCLES_2"], 1), acc["SQ_WAIT_INST_LDS"] / max(acc["SQ_WAVE_CYCLES_2"], 1), acc["SQ_LDS_BANK_CONFLICT"]))
hit, miss = acc["TCC_HIT_sum"], acc["TCC_MISS_sum"]
print("  L2: %.3g requests, hit rate %.3f; fabric fetch 2 x FETCH_SIZE = %.1f GB = %.2f TB/s" % (
    acc["TCC_REQ_sum"], hit / max(hit + miss, 1), 2 * acc["FETCH_SIZE"] * 1024 / 1e9, 2 * acc["FETCH_SIZE"] * 1024 / (ms * 1e-3) / 1e12))
PY
done
cat $OUT/summary.txt
rm -f $C; find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
