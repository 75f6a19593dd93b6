#!/bin/bash
# Why the 32 x 32 x 16 instantiation of the genotype rotation loses ("a measured negative with the counters that show why"):
# both MFMA shapes of rotate_geno_kernel (PG_GENO_MFMA=16|32) under three PMC passes each — matrix-pipe busy and issue stalls; waits on memory
# and LDS activity; L2 hits / misses and fabric fetches.  Output: gpurun_out/ab_rot2/summary.txt (copied to profiles/r04_rotate_mfma_ab.txt)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/ab_rot2; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C=$OUT/ec.npz
python3 $ROOT/bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 --eigh-cache $C > /dev/null 2>&1
: > $OUT/summary.txt
for mf in 16 32; do
  i=0
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum GRBM_GUI_ACTIVE" "FETCH_SIZE GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    PG_GENO_MFMA=$mf timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/mf${mf}_$i -o c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --e2e 0 --cpu-sample 0 --eigh-cache $C > $OUT/mf${mf}_$i.json 2> $OUT/mf${mf}_$i.log
    rc=$?; [ $rc -eq 124 ] && { echo "timeout"; exit 1; }
  done
  python3 - >> $OUT/summary.txt <<PY
import csv, collections, glob
acc = collections.defaultdict(float); dur = []
for i in (1, 2, 3, 4):
    fs = glob.glob("$OUT/mf${mf}_%d/**/c_counter_collection.csv" % i, recursive=True)
    if not fs: continue
    rows = [r for r in csv.DictReader(open(fs[0])) if "rotate_geno_kernel" in r["Kernel_Name"]]
    nl = len({r["Dispatch_Id"] for r in rows})
    for r in rows:
        key = r["Counter_Name"] if r["Counter_Name"] not in ("GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES") else r["Counter_Name"] + "_%d" % i
        acc[key] += float(r["Counter_Value"]) / max(nl, 1)
        if i == 1 and r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
ms = sum(dur) / len(dur)
cyc = acc["GRBM_GUI_ACTIVE_1"] / 8.0
print("PG_GENO_MFMA=$mf (v_mfma_f32_%s_f16), per launch of 100 000 SNPs at n = 10 000:" % ("16x16x32" if $mf == 16 else "32x32x16"))
print("  duration %.2f ms, clock %.2f GHz; matrix pipe busy %.3f; of the wave-cycles: waiting on memory/barriers (SQ_WAIT_ANY) %.2f, issue-stalled (SQ_WAIT_INST_ANY) %.2f, issuing %.2f" % (
    ms, cyc / (ms * 1e6), acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), acc["SQ_WAIT_ANY"] / acc["SQ_WAVE_CYCLES_1"], acc["SQ_WAIT_INST_ANY"] / acc["SQ_WAVE_CYCLES_1"], acc["SQ_ACTIVE_INST_ANY"] / acc["SQ_WAVE_CYCLES_1"]))
print("  LDS: %.3g instructions, active %.2f and issue-stalled %.2f of the wave-cycles, bank-conflict cycles %.3g" % (
    acc["SQ_INSTS_LDS"], acc["SQ_ACTIVE_INST_LDS"] / max(acc["SQ_WAVE_CYCLES_2"], 1), acc["SQ_WAIT_INST_LDS"] / max(acc["SQ_WAVE_CYCLES_2"], 1), acc["SQ_LDS_BANK_CONFLICT"]))
hit, miss = acc["TCC_HIT_sum"], acc["TCC_MISS_sum"]
print("  L2: %.3g requests, hit rate %.3f; fabric fetch 2 x FETCH_SIZE = %.1f GB = %.2f TB/s; LDS-DMA volume 236 GB = %.2f TB/s L2 -> LDS" % (
    acc["TCC_REQ_sum"], hit / max(hit + miss, 1), 2 * acc["FETCH_SIZE"] * 1024 / 1e9, 2 * acc["FETCH_SIZE"] * 1024 / (ms * 1e-3) / 1e12, 236e9 / (ms * 1e-3) / 1e12))
PY
done
cat $OUT/summary.txt
rm -f $C; find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
