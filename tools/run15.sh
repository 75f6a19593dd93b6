#!/bin/bash
# int8 rotation: accuracy against fp64, A/B timing against the fp16 x 2 kernel, rotation tests
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/i8
for v in 1 0; do
  echo "== PG_GENO_I8=$v"; 
  PG_GENO_I8=$v timeout -k 10 200 python tools/geno_accuracy.py 3000 512 || exit 1
  PG_GENO_I8=$v timeout -k 10 200 python tools/bench_rotate_geno.py 10000 100000 || exit 1
done
timeout -k 10 600 python -m pytest tests/test_gpu_rotate.py -x -q -m gpu 2>&1 | tail -n 8
