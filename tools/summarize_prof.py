"""Condense rocprofv3 CSVs under gpurun_out/ into the small tracked summaries under profiles/."""
import csv, json, os, sys, collections
R = "/root/repo"
def pmc(path):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return out
def kstats(path, top=30):
    rows = list(csv.DictReader(open(path)))
    return [{"kernel": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
             "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:top]]
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
summ = {"note": "rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 --warmup 2 --cpu-sample 0` (MI355X, 1 GPU); "
                "PMC passes (separate runs, --kernel-trace + one counter group each) on tools/bench_rotate.py 10000 16384 and "
                "tools/bench_assoc.py 10000 16384 5 / tools/bench_rotate_geno.py 10000 16384 = the same kernels and shapes as the bench step "
                "(bench.py itself segfaults inside rocprofv3's counter-collection mode while it queues the ~10^5 small syevd launches). "
                "FETCH_SIZE/WRITE_SIZE are KB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950, "
                "so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024; the counters sit on the L2's fabric side, i.e. Infinity-Cache hits are included."}
ks = os.path.join(R, "gpurun_out/prof_r01b/r01b_kernel_stats.csv")
if not os.path.exists(ks):
    ks = os.path.join(R, "gpurun_out/prof_r01/r01_kernel_stats.csv")
if os.path.exists(ks):
    summ["kernel_stats"] = kstats(ks)
def avg(d, k, c):
    for name, cs in d.items():
        if k in name and c in cs:
            v = cs[c]; return sum(v) / len(v)
    return None
pm = {}
for key, kern, fdir, wdir in (("rotate_kernel", "rotate_kernel<4>", "pmc_fetch", "pmc_write"), ("rotate_geno_kernel", "rotate_geno_kernel", "pmc_fetch_geno", "pmc_write_geno"),
                              ("assoc_kernel", "assoc_kernel<5>", "pmc_fetch_assoc", "pmc_write_assoc")):
    try:
        f = avg(pmc(os.path.join(R, "gpurun_out", fdir, "f_counter_collection.csv")), kern, "FETCH_SIZE")
        w = avg(pmc(os.path.join(R, "gpurun_out", wdir, "w_counter_collection.csv")), kern, "WRITE_SIZE")
        pm[key] = {"FETCH_SIZE_KB": f, "WRITE_SIZE_KB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024,
                   "hbm_bytes_uncorrected": (f + w) * 1024}
    except Exception as ex:
        pm[key] = {"error": str(ex)}
try:
    l2 = pmc(os.path.join(R, "gpurun_out/pmc_l2/l_counter_collection.csv"))
    h, m = avg(l2, "rotate_kernel<4>", "TCC_HIT_sum"), avg(l2, "rotate_kernel<4>", "TCC_MISS_sum")
    pm["rotate_kernel"]["L2_hit_rate"] = h / (h + m)
except Exception as ex:
    pass
n, B, c = 10000, 16384, 5
pm["rotate_kernel"]["algorithmic_bytes_per_launch"] = 4.0 * n * B + 4.0 * n * n + 4.0 * 10048 * B
pm["assoc_kernel"]["algorithmic_bytes_per_launch"] = (4.0 * 10048 + 36) * B
pm["rotate_geno_kernel"]["algorithmic_bytes_per_launch"] = 2.0 * 10048 * B + 6.0 * n * n + 4.0 * 10048 * B
pm["rotate_geno_kernel"]["note"] = "GEMM kernel only (bf16 codes 2 B/elem in, 3 bf16 planes of U, f32 out); the detect/encode passes read the raw f32 block twice more"
summ["pmc"] = pm
json.dump(summ, open(os.path.join(R, "profiles", f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(pm, indent=1))
