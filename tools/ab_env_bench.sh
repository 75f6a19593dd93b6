#!/bin/bash
# bench.py under settings of ONE environment knob of the library: usage ab_env_bench.sh NAME v1 v2 ...   (two rounds, eigenpairs cached)
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/ab_env; mkdir -p $OUT
cd $ROOT
NAME=$1; shift
python3 bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 --eigh-cache $OUT/eig.npz > $OUT/prime.json 2> $OUT/prime.err || exit 1
for round in 1 2; do
for v in "$@"; do
  env $NAME=$v timeout -k 10 300 python3 bench.py --steps 10 --warmup 1 --e2e 0 --cpu-sample 0 --eigh-cache $OUT/eig.npz > $OUT/${v}_$round.json 2> $OUT/${v}_$round.err || { echo "fail $v"; exit 1; }
  python3 - <<PY
import json
j=json.loads(open("$OUT/${v}_$round.json").read().strip().splitlines()[-1])
print("$NAME=$v round $round: value %.0f ms/step %.2f rot %.3f assoc %.3f" % (j["value"], j["ms_per_step"], j["roofline_rotate"]["avg_launch_ms"], j["roofline_assoc"]["avg_launch_ms"]))
PY
done; done
rm -f $OUT/eig.npz
