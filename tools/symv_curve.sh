#!/bin/bash
# per-launch duration of symv_sym_kernel against the trailing size (kernel trace of one solve) -> effective bandwidth curve
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sc; timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sc -o k -- python3 $ROOT/tools/bench_syevd.py ${1:-10000} > /tmp/sc.log 2>&1 || { tail -5 /tmp/sc.log; exit 1; }
python3 - <<'PY'
import csv, collections
rows = [r for r in csv.DictReader(open('/tmp/sc/k_kernel_trace.csv'))]
byk = collections.defaultdict(list)
for r in rows:
    nm = r['Kernel_Name'].split('(')[0].replace('void ', '')
    byk[nm].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
for nm in ('pg::symv_sym_kernel', 'pg::w_update_kernel<true>', 'pg::col_kernel'):
    v = sorted(byk[nm]); n = len(v) // 2      # two solves: second half = the timed one
    v = v[n:]
    N = len(v) + 1
    print(nm, 'launches per solve', len(v))
    for lo in range(0, len(v), len(v) // 10):
        seg = v[lo:lo + len(v) // 10]
        dur = sum(e - s for s, e in seg) / len(seg) / 1e3
        j = lo + len(seg) / 2; m = N - j
        gap = sum(max(0, seg[i + 1][0] - seg[i][1]) for i in range(len(seg) - 1)) / max(1, len(seg) - 1) / 1e3
        extra = f"  lower-triangle bytes {m * m * 4 / 1e6:7.1f} MB -> {m * m * 4 / dur / 1e6:6.2f} TB/s" if 'symv' in nm else ''
        print(f"  columns {lo:5d}..{lo + len(seg):5d} (trailing ~{int(m):5d}): avg {dur:7.2f} us, avg gap to next launch of the same kernel {gap:7.2f} us{extra}")
PY
