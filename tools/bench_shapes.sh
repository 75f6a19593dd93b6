#!/bin/bash
# The other BASELINE shapes through the same bench (one GPU, resident inputs), plus the per-kernel rates at the configs[4] size.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$ROOT/gpurun_out/shapes; mkdir -p $OUT
run() { name=$1; shift; timeout -k 10 300 python3 $ROOT/bench.py --e2e 0 --cpu-sample 0 --steps 4 --warmup 1 "$@" > $OUT/$name.json 2> $OUT/$name.err; echo "$name rc=$?"; }
run n1940_c1 --n 1940 --c 1 --snps 65536
run n2000_c5 --n 2000 --c 5 --snps 100000
run n2000_c5_grid --n 2000 --c 5 --snps 100000 --grid 1
run n10000_c1 --n 10000 --c 1 --snps 65536
run n10000_c5_grid --n 10000 --c 5 --grid 1
run n10000_c10 --n 10000 --c 10 --snps 65536
timeout -k 10 300 python3 $ROOT/tools/bench_assoc.py 50000 4096 5 1 > $OUT/assoc_n50000_grid.log 2>&1; echo "assoc50k rc=$?"
timeout -k 10 400 python3 $ROOT/tools/bench_rotate_geno.py 50000 4096 > $OUT/rotate_n50000.log 2>&1; echo "rot50k rc=$?"
python3 - <<PY
import json, glob, os
for f in sorted(glob.glob("$OUT/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f)[:-5], "| %.2f M SNPs/s | %.2f ms/step | rotate %.2f M/s | assoc %.2f M/s | eigh %.3f s | evals %.1f+%.1f" % (
            d["value"] / 1e6, d["ms_per_step"], d["stage_snps_per_s_per_gpu"]["rotate"] / 1e6, d["stage_snps_per_s_per_gpu"]["assoc"] / 1e6,
            d["eigh_seconds"], d["evals_per_snp"]["fast"], d["evals_per_snp"]["newton"]))
    except Exception as ex:
        print(f, "ERR", ex)
PY
tail -2 $OUT/assoc_n50000_grid.log; tail -2 $OUT/rotate_n50000.log
