#!/bin/bash
# bench.py under alternative builds of the library (PYGEMMA_HIP_LIB): usage ab_bench_lib.sh name=path ...  ("base" = shipped lib)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_bench; mkdir -p $OUT
cd $ROOT
python3 bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 --eigh-cache $OUT/eig.npz > $OUT/prime.json 2> $OUT/prime.err || exit 1
for round in 1 2; do
for kv in "$@"; do
  name=${kv%%=*}; path=${kv#*=}
  if [ "$path" = base ]; then unset PYGEMMA_HIP_LIB; else export PYGEMMA_HIP_LIB=$ROOT/$path; fi
  timeout -k 10 300 python3 bench.py --steps 10 --warmup 1 --e2e 0 --cpu-sample 0 --eigh-cache $OUT/eig.npz > $OUT/${name}_$round.json 2> $OUT/${name}_$round.err || { echo "fail $name"; exit 1; }
  python3 - <<PY
import json
j=json.loads(open("$OUT/${name}_$round.json").read().strip().splitlines()[-1])
print("$name $round value %.0f ms/step %.2f rot %.3f assoc %.3f" % (j["value"], j["ms_per_step"], j["roofline_rotate"]["avg_launch_ms"], j["roofline_assoc"]["avg_launch_ms"]))
PY
done; done
