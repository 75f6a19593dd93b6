#!/bin/bash
# Long randomised soak on one GPU box: many seeds of the stress sweeps, the degenerate-input sweeps, determinism, leak check.  Stops at the first failure.
ROOT=${GRAFT_REPO_ROOT:-/root/repo}; cd $ROOT
set -e
for s in ${SEEDS:-41 42 43 44 45 46 47 48 49 50 51 52}; do
  echo "== seed $s"; timeout -k 10 600 python tools/stress_assoc.py $s | tail -1
  timeout -k 10 300 python tools/stress_rotate.py $s | tail -1
done
timeout -k 10 300 python tools/adversarial_assoc.py 7 | tail -1
timeout -k 10 300 python tools/adversarial_rotate.py | tail -1
timeout -k 10 300 python tools/adversarial_lrt.py | tail -1
timeout -k 10 300 python tools/robust_K.py | tail -1
timeout -k 10 300 python tools/robust_small.py | tail -1
timeout -k 10 600 python tools/determinism.py 10000 20000 10 | tail -2
timeout -k 10 600 python tools/leak_check.py 15 | tail -1
echo SOAK-LONG OK
