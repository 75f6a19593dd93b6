#!/bin/bash
# the round's full GPU check: every -m gpu test, the bench line, the smoke entry
set -o pipefail
mkdir -p gpurun_out/full
cd $GRAFT_REPO_ROOT
timeout -k 10 1000 python -m pytest tests/ -x -q -m gpu > gpurun_out/full/pytest_gpu.log 2>&1; rc=$?; echo "pytest -m gpu rc=$rc"; tail -n 6 gpurun_out/full/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d=json.loads(open('gpurun_out/full/bench.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','eigh_seconds')}, d['roofline']['frac'], d.get('roofline_eigh'), d.get('e2e',{}).get('seconds'))
PY
timeout -k 10 200 python -c "import __graft_entry__ as g; g.smoke()"; echo "smoke rc=$?"
