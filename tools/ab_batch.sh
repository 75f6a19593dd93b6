#!/bin/bash
# bench.py at a list of resident batch sizes (BATCHES="16384 100000 ..."): step time, rotation and association time per launch.
cd $GRAFT_REPO_ROOT
python3 bench.py --steps 1 --warmup 0 --e2e 0 --cpu-sample 0 --eigh-cache /tmp/eig.npz > /dev/null 2>&1
for b in ${BATCHES:-16384 25000 33334 50000 100000 16384}; do
  python3 bench.py --steps 10 --warmup 1 --e2e 0 --cpu-sample 0 --eigh-cache /tmp/eig.npz --batch $b 2>/dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print($b, 'value %.0f ms/step %.2f rot %.3f assoc %.3f' % (j['value'], j['ms_per_step'], j['roofline_rotate']['avg_launch_ms'], j['roofline_assoc']['avg_launch_ms']))"
done
