"""Condense gpurun_out/prof_<tag>_extra/ (tools/collect_profiles_extra.sh) into profiles/<tag>_extra_summary.json:
per kernel beside the bench step, duration under the profiler, clock, MFMA-busy / VALU-issue fraction, HBM-side bytes."""
import csv, json, os, re, sys, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
D = os.path.join(R, "gpurun_out", f"prof_{tag}_extra")

def table(name):
    rows = list(csv.DictReader(open(os.path.join(D, name, "c_counter_collection.csv"))))
    k = collections.defaultdict(lambda: {"n": 0, "ms": 0.0, "ctr": collections.defaultdict(float)})
    seen = set()
    for r in rows:
        nm = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        e = k[nm]
        if r["Dispatch_Id"] not in seen:
            seen.add(r["Dispatch_Id"]); e["n"] += 1
            e["ms"] += (float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e6
        e["ctr"][r["Counter_Name"]] += float(r["Counter_Value"])
    return k

def stdout_line(name):
    return [l.strip() for l in open(os.path.join(D, name + ".log")) if ("TFLOP" in l or "SNPs/s" in l) and "rocprofv3" not in l]

out = {"note": "MI355X, 1 GPU, rocprofv3 --kernel-trace --pmc (one counter group per run) on the tools/ drivers named in `driver`. Sums over all "
               "dispatches of the run (first call + timed repeats). mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * GRBM_GUI_ACTIVE / 8); "
               "valu_issue = 4 * SQ_INSTS_VALU / the same; clock = GRBM_GUI_ACTIVE / 8 / duration; bytes = (2*FETCH_SIZE + WRITE_SIZE) KB "
               "(gfx950 correction of MI355X_MICROARCH.md).  `driver_stdout` is the driver's own wall-clock line under the profiler."}
for name, driver in (("dgemm_panel", "bench_dgemm.py 8192 8192 128 1"), ("dgemm_big", "bench_dgemm.py 8192 8192 8192"),
                     ("kinship", "bench_kinship.py 10000 20000"), ("lrt", "bench_lrt.py 10000 16384 5"), ("rot32", "bench_rotate.py 10000 16384")):
    e = {"driver": driver, "driver_stdout": stdout_line(name), "kernels": {}}
    for nm, v in sorted(table(name).items(), key=lambda kv: -kv[1]["ms"])[:4]:
        c = v["ctr"]; gui = c.get("GRBM_GUI_ACTIVE", 0.0)
        ke = {"launches": v["n"], "avg_ms": v["ms"] / v["n"]}
        if gui:
            ke["clock_GHz"] = gui / 8 / (v["ms"] * 1e6)
            if "SQ_VALU_MFMA_BUSY_CYCLES" in c: ke["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (1024 * gui / 8)
            if "SQ_INSTS_VALU" in c: ke["valu_issue"] = 4 * c["SQ_INSTS_VALU"] / (1024 * gui / 8); ke["valu_insts_per_launch"] = c["SQ_INSTS_VALU"] / v["n"]
        e["kernels"][nm] = ke
    out[name] = e
try:
    f, w = table("rot32_fetch"), table("rot32_write")
    for nm in f:
        if "rotate" in nm or "gemm" in nm:
            fs, ws = f[nm]["ctr"]["FETCH_SIZE"] / f[nm]["n"], w[nm]["ctr"]["WRITE_SIZE"] / w[nm]["n"]
            out["rot32"]["kernels"].setdefault(nm, {}).update({"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "hbm_bytes_per_launch": (2 * fs + ws) * 1024,
                                                               "algorithmic_bytes_per_launch": 4.0 * 10048 * 16384 * 2 + 4.0 * 10000 * 10048})
except Exception as ex:
    out["rot32"]["traffic_error"] = repr(ex)
rows = list(csv.DictReader(open(os.path.join(D, "syevd_stats", "k_kernel_stats.csv"))))
out["syevd_n10000_kernel_stats"] = {"driver": "bench_syevd.py 10000 (first call + timed repeats)", "stdout": [l.strip() for l in open(os.path.join(D, "syevd_stats.log")) if "syevd" in l and "rocprof" not in l][:3],
    "top": [{"kernel": re.sub(r"\(.*", "", r["Name"]), "calls": int(r["Calls"]), "total_ms": float(r["TotalDurationNs"]) / 1e6, "avg_us": float(r["AverageNs"]) / 1e3, "pct": float(r["Percentage"])} for r in rows[:12]]}
json.dump(out, open(os.path.join(R, "profiles", f"{tag}_extra_summary.json"), "w"), indent=1)
print(json.dumps(out, indent=1)[:6000])
