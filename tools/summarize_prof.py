"""Condense the rocprofv3 CSVs that tools/collect_profiles.sh left under gpurun_out/prof_<tag>_final/ into the small tracked
files under profiles/:  <tag>_bench.json (the bench line), <tag>_bench_under_rocprof.json, <tag>_kernel_stats.csv
(--kernel-trace --stats of the same bench command), <tag>_summary.json (PMC on bench.py itself: HBM-side traffic, pipe-busy fractions)."""
import csv, json, os, shutil, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
D = os.path.join(R, "gpurun_out", f"prof_{tag}_final")
P = os.path.join(R, "profiles")

def pmc(name):
    """counter values and durations per kernel name over the dispatches that did the full-batch work: a dispatch counts when its own
    duration is at least half of the longest one of that kernel (drops the ragged last batch and the predicated launches of
    pg_rotate_auto_dev that leave at once)"""
    rows = list(csv.DictReader(open(os.path.join(D, name, "c_counter_collection.csv"))))
    dmax = collections.defaultdict(float)
    for r in rows:
        r["_dur"] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
        dmax[r["Kernel_Name"]] = max(dmax[r["Kernel_Name"]], r["_dur"])
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for r in rows:
        if r["_dur"] >= 0.5 * dmax[r["Kernel_Name"]]:
            out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            dur[r["Kernel_Name"]].append(r["_dur"])
    return out, dur

def avg(d, kern, ctr=None):
    for name, cs in d.items():
        if kern in name:
            v = cs[ctr] if ctr else cs
            return sum(v) / len(v) if v else None
    return None

for f in ("bench.json", "bench_under_rocprof.json"):
    shutil.copy(os.path.join(D, f), os.path.join(P, f"{tag}_{f}"))
rows = list(csv.DictReader(open(os.path.join(D, "stats", "k_kernel_stats.csv"))))
with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows[:40])

_cfg = json.load(open(os.path.join(D, "bench.json")))["config"]
n, B = int(_cfg["n"]), int(_cfg["batch"])          # the PMC passes run bench.py at its defaults: the same batch shape
npad = (n + 63) // 64 * 64
summ = {"note": "MI355X, 1 GPU. kernel_stats: rocprofv3 --kernel-trace --stats of `python3 bench.py --e2e 0 --cpu-sample 0` (the bench's timed command). "
                "pmc: separate rocprofv3 --kernel-trace --pmc passes (one counter group per run) on bench.py ITSELF (--steps 2 --warmup 1, eigenpairs "
                "from --eigh-cache; full batches of the bench's own --batch only (default: one 100 000-SNP batch)). FETCH_SIZE/WRITE_SIZE are KB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide "
                "coalesced reads by 2x on gfx950, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (fabric side of L2: Infinity-Cache hits included). "
                "Pipe-busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8); VALU issue = 4 cycles * SQ_INSTS_VALU / the same; "
                "clock = GRBM_GUI_ACTIVE / 8 / duration. pmc_full_bench: bench.py including the eigensolver under --pmc (the library drains the stream once per panel)."}
summ["batch_shape"] = {"n": n, "batch": B, "c": int(_cfg["c"])}
summ["kernel_stats_top"] = [{"kernel": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                             "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:14]]
pm = {}
npad8 = (n + 127) // 128 * 128
for key, kern, algo in (("rotate_geno_i8_kernel", "rotate_geno_i8_kernel", 1.0 * npad8 * B + 3.0 * n * npad8 + 4.0 * npad * B),     # int8 codes, 3 int8 planes of U, fp32 out
                        ("rotate_geno_kernel", "rotate_geno_kernel", 2.0 * npad * B + 4.0 * n * npad + 4.0 * npad * B),
                        ("assoc_kernel", "assoc_kernel<5", (4.0 * npad + 36) * B)):
    e = {}
    try:
        f, _ = pmc("fetch"); w, _ = pmc("write")
        fs, ws = avg(f, kern, "FETCH_SIZE"), avg(w, kern, "WRITE_SIZE")
        e.update({"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "hbm_bytes_per_launch": (2 * fs + ws) * 1024,
                  "hbm_bytes_uncorrected": (fs + ws) * 1024, "algorithmic_bytes_per_launch": algo})
        s, dur = pmc("sq_valu")
        gui, ms = avg(s, kern, "GRBM_GUI_ACTIVE"), avg(dur, kern)
        e.update({"avg_ms_profiled": ms, "clock_GHz": gui / 8 / (ms * 1e6)})
        iv = avg(s, kern, "SQ_INSTS_VALU")
        if iv is not None:
            e["valu_issue_frac"] = 4 * iv / (1024 * gui / 8)
            e["valu_insts_per_snp"] = iv / B
        s2, dur2 = pmc("sq_mfma")
        gui2, mf = avg(s2, kern, "GRBM_GUI_ACTIVE"), avg(s2, kern, "SQ_VALU_MFMA_BUSY_CYCLES")
        if mf is not None:
            e["mfma_busy_frac"] = mf / (1024 * gui2 / 8)
        wv, wa = avg(s2, kern, "SQ_WAVE_CYCLES"), avg(s2, kern, "SQ_WAIT_INST_ANY")
        if wv and wa:
            e["wait_frac_of_wave_cycles"] = wa / wv
    except Exception as ex:
        e["error"] = repr(ex)
    if key.startswith("rotate_geno") and ("error" in e or e.get("avg_ms_profiled", 0.0) < 1.0):
        continue            # that rotation kernel did not run in this profile, or only as predicated launches that leave at once (PG_GENO_I8 decides)
    pm[key] = e
summ["pmc"] = pm
try:
    summ["pmc_full_bench"] = {"ran": os.path.getsize(os.path.join(D, "pmc_full.json")) > 10,
                              "value": json.load(open(os.path.join(D, "pmc_full.json"))).get("value"),
                              "note": "rocprofv3 --kernel-trace --pmc SQ_WAVES -- python3 bench.py --steps 1 --warmup 0 (eigensolver included)"}
except Exception as ex:
    summ["pmc_full_bench"] = {"ran": False, "error": repr(ex)}
json.dump(summ, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(pm, indent=1))
print(json.dumps(summ["pmc_full_bench"]))
