#!/bin/bash
# Rehearsal + randomised sweeps on one GPU box: the launcher path the driver uses for N > 1 (here with one rank and the collectives
# forced on), the default bench line with its wall clock, then new seeds of the stress sweeps.  Stops at the first failure.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
OUT=$ROOT/gpurun_out/soak; mkdir -p $OUT
set -e
export PYGEMMA_BENCH_FORCE_COMM=1
echo "== torch.distributed.run, 1 rank, collectives forced"
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 2 --warmup 1 --e2e 0 --cpu-sample 0 | tee $OUT/torchrun1.json
unset PYGEMMA_BENCH_FORCE_COMM
echo "== default bench line"
T0=$(date +%s.%N); timeout -k 10 600 python bench.py > $OUT/bench_default.json; T1=$(date +%s.%N); cut -c1-400 $OUT/bench_default.json; echo "bench.py default wall clock: $(python3 -c "print($T1 - $T0)") s"
for s in ${SEEDS:-31 32 33 34}; do
  echo "== stress_assoc seed $s"; timeout -k 10 600 python tools/stress_assoc.py $s
  echo "== stress_rotate seed $s"; timeout -k 10 300 python tools/stress_rotate.py $s | tail -2
done
echo "== robust_small"; timeout -k 10 600 python tools/robust_small.py | tail -5
echo SOAK OK
