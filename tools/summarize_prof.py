"""Condense the rocprofv3 CSVs that tools/collect_profiles.sh left under gpurun_out/prof_<tag>/ into the small tracked
files under profiles/:  <tag>_bench.json (the bench line), <tag>_bench_under_rocprof.json, <tag>_kernel_stats.csv
(--kernel-trace --stats of the same bench command), <tag>_summary.json (PMC: HBM-side traffic, pipe-busy fractions)."""
import csv, json, os, shutil, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01c"
D = os.path.join(R, "gpurun_out", f"prof_{tag}_final")
P = os.path.join(R, "profiles")

def pmc(name):
    out = collections.defaultdict(lambda: collections.defaultdict(list))
    path = os.path.join(D, name, "c_counter_collection.csv")
    for r in csv.DictReader(open(path)):
        out[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(os.path.join(D, name, "c_kernel_trace.csv"))):
        dur[r["Kernel_Name"]].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6)
    return out, dur

def avg(d, kern, ctr=None):
    for name, cs in d.items():
        if kern in name:
            v = cs[ctr] if ctr else cs
            return sum(v) / len(v) if v else None
    return None

shutil.copy(os.path.join(D, "bench.json"), os.path.join(P, f"{tag}_bench.json"))
shutil.copy(os.path.join(D, "bench_under_rocprof.json"), os.path.join(P, f"{tag}_bench_under_rocprof.json"))
rows = list(csv.DictReader(open(os.path.join(D, "stats", "k_kernel_stats.csv"))))
with open(os.path.join(P, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
    w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
    w.writeheader()
    w.writerows(rows[:40])

n, B = 10000, 16384
npad = 10048
summ = {"note": "MI355X, 1 GPU. kernel_stats: rocprofv3 --kernel-trace --stats of `python3 bench.py --steps 10 --warmup 2 --cpu-sample 0`. "
                "pmc: separate rocprofv3 --kernel-trace --pmc passes (one counter group per run) on tools/bench_rotate_geno.py / bench_assoc.py / "
                "bench_rotate.py at the bench step's exact shapes (n=10000, 16384 SNPs, c=5): rocprofv3's counter mode segfaults under the ~10^5 "
                "small syevd launches of bench.py itself. FETCH_SIZE/WRITE_SIZE are KB; per MI355X_MICROARCH.md FETCH_SIZE under-reports wide "
                "coalesced reads by 2x on gfx950, so hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (fabric side of L2: Infinity-Cache hits included). "
                "Pipe-busy: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE/8); VALU issue = 4 cycles * SQ_INSTS_VALU / the same; "
                "clock = GRBM_GUI_ACTIVE / 8 / duration."}
summ["kernel_stats_top"] = [{"kernel": r["Name"], "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6,
                             "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:12]]
pm = {}
for key, kern, suffix, algo in (("rotate_geno_kernel", "rotate_geno_kernel", "geno", 2.0 * npad * B + 4.0 * n * npad + 4.0 * npad * B),
                                ("assoc_kernel", "assoc_kernel<5>", "assoc", (4.0 * npad + 36) * B),
                                ("rotate_kernel", "rotate_kernel<4>", "rot", 4.0 * n * B + 4.0 * n * n + 4.0 * npad * B)):
    e = {}
    try:
        f, _ = pmc("fetch_" + suffix); w, _ = pmc("write_" + suffix)
        fs, ws = avg(f, kern, "FETCH_SIZE"), avg(w, kern, "WRITE_SIZE")
        e.update({"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "hbm_bytes_per_launch": (2 * fs + ws) * 1024,
                  "hbm_bytes_uncorrected": (fs + ws) * 1024, "algorithmic_bytes_per_launch": algo})
        s, dur = pmc("sq_" + suffix)
        gui, ms = avg(s, kern, "GRBM_GUI_ACTIVE"), avg(dur, kern)
        e.update({"avg_ms_profiled": ms, "clock_GHz": gui / 8 / (ms * 1e6)})
        mf = avg(s, kern, "SQ_VALU_MFMA_BUSY_CYCLES")
        if mf is not None:
            e["mfma_busy_frac"] = mf / (1024 * gui / 8)
        iv = avg(s, kern, "SQ_INSTS_VALU")
        if iv is not None:
            e["valu_issue_frac"] = 4 * iv / (1024 * gui / 8)
            e["valu_insts_per_snp"] = iv / B
        wv, wa = avg(s, kern, "SQ_WAVE_CYCLES"), avg(s, kern, "SQ_WAIT_INST_ANY")
        if wv and wa:
            e["wait_frac_of_wave_cycles"] = wa / wv
    except Exception as ex:
        e["error"] = repr(ex)
    pm[key] = e
for key, name, flops in (("dgemm_panel_update", "sq_dgemm_panel", 2.0 * 8192 * 8192 * 128), ("dgemm_8192_cubed", "sq_dgemm_big", 2.0 * 8192 ** 3)):
    try:
        sq, dur = pmc(name)
        gui, ms, mf = avg(sq, "dgemm_kernel", "GRBM_GUI_ACTIVE"), avg(dur, "dgemm_kernel"), avg(sq, "dgemm_kernel", "SQ_VALU_MFMA_BUSY_CYCLES")
        pm[key] = {"note": "fp64 MFMA GEMM of the eigensolver (v_mfma_f64_16x16x4_f64), tools/bench_dgemm.py", "avg_ms_profiled": ms,
                   "TFLOPs": flops / ms / 1e9, "frac_of_78.6": flops / ms / 1e9 / 78.6, "clock_GHz": gui / 8 / (ms * 1e6),
                   "mfma_busy_frac": mf / (1024 * gui / 8)}
    except Exception as ex:
        pm[key] = {"error": repr(ex)}
pm["rotate_geno_kernel"]["note"] = "GEMM kernel only (fp16 codes 2 B/elem in, 2 fp16 planes of U, f32 out); the detect/encode passes read the raw f32 block twice more"
summ["pmc"] = pm
json.dump(summ, open(os.path.join(P, f"{tag}_summary.json"), "w"), indent=1)
print(json.dumps(pm, indent=1))
print(open(os.path.join(D, "bench.json")).read()[:1500])
