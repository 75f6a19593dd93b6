"""Per-phase, per-kernel totals of one pg_syevd_dev solve from a rocprofv3 --kernel-trace CSV (the second solve in the file).
usage: trace_phases.py path/to/kernel_trace.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if 'sym_from_lower' in r['Kernel_Name']]
rows = rows[idx[-1]:]


def name(r):
    return r['Kernel_Name'].split('(')[0].replace('void ', '').replace('pg::', '')


ph = 'stage1'
agg = collections.defaultdict(lambda: [0, 0.0])
gaps = collections.defaultdict(float)
span = {}
last_end = None
for r in rows:
    nm = name(r); st = int(r['Start_Timestamp']); en = int(r['End_Timestamp'])
    if nm == 'band_extract_kernel': ph = 'bc'
    elif nm in ('scatter_leaves_kernel',): ph = 'dc'
    elif nm == 'bt2_prep_kernel': ph = 'bt2'
    elif ph == 'bt2' and nm.startswith('dgemm_kernel<true, false, 4>'): ph = 'bt1'
    elif nm == 'finalize_kernel': ph = 'out'
    k = (ph, nm)
    agg[k][0] += 1; agg[k][1] += (en - st) / 1e6
    if last_end is not None: gaps[ph] += max(0, st - last_end) / 1e6
    last_end = en
    span.setdefault(ph, [st, en]); span[ph][1] = en
for (p, nm), (c, t) in sorted(agg.items(), key=lambda x: (x[0][0], -x[1][1])):
    print(f"{p:7s} {nm:42s} calls {c:5d} total {t:8.2f} ms  avg {1e3 * t / c:8.1f} us")
print("gaps between kernels (ms):", {k: round(v, 2) for k, v in gaps.items()})
print("phase spans (ms):", {k: round((v[1] - v[0]) / 1e6, 2) for k, v in span.items()})
