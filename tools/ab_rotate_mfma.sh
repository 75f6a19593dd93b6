#!/bin/bash
# A/B of the genotype rotation's two MFMA shapes (PG_GENO_MFMA=16|32) under rocprofv3 --pmc: duration, MFMA-busy fraction and clock of rotate_geno_kernel.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/ab_rot; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
C=$OUT/ec.npz
python3 $ROOT/bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 --eigh-cache $C > /dev/null 2>&1
for mf in 16 32; do
  PG_GENO_MFMA=$mf timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAVE_CYCLES --output-format csv -d $OUT/mf$mf -o c -- python3 $ROOT/bench.py --steps 2 --warmup 1 --e2e 0 --cpu-sample 0 --eigh-cache $C > $OUT/mf$mf.json 2> $OUT/mf$mf.log
  python3 - <<PY
import csv, collections, glob
f = glob.glob("$OUT/mf$mf/**/c_counter_collection.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "rotate_geno_kernel" in r["Kernel_Name"]]
acc = collections.defaultdict(float); dur = []
for r in rows:
    acc[r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE": dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
cyc = acc["GRBM_GUI_ACTIVE"] / 8.0
print("PG_GENO_MFMA=$mf: %d launches, avg %.2f ms, MFMA busy %.3f, clock %.2f GHz, waiting %.2f of wave-cycles" % (
    len(dur), sum(dur) / len(dur), acc["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024.0 * cyc), cyc / (sum(dur) * 1e6), acc["SQ_WAIT_INST_ANY"] / max(acc["SQ_WAVE_CYCLES"], 1)))
PY
done
rm -f $C; find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -size +2M -delete
