"""Condense gpurun_out/prof_<tag>_syevd/ (tools/collect_syevd_prof.sh) into profiles/<tag>_syevd_summary.json: per kernel of ONE two-stage
solve at n — calls, total ms, and, from the --pmc pass, the MFMA-busy fraction and the clock; for the GEMM-shaped kernels the useful
fp64 TFLOP/s (flops of the algorithm, not of the padded tiles) against the 78.6 TF peak."""
import collections, csv, json, os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, n = sys.argv[1], int(sys.argv[2])
D = os.path.join(R, "gpurun_out", f"prof_{tag}_syevd")


def short(nm):
    return nm.split("(")[0].replace("void ", "").replace("pg::", "")


def one_solve(rows):
    idx = [i for i, r in enumerate(rows) if "sym_from_lower" in r["Kernel_Name"]]
    return rows[idx[-1]:]


tr = one_solve(list(csv.DictReader(open(os.path.join(D, "trace", "t_kernel_trace.csv")))))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in tr:
    k = short(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
pm = collections.defaultdict(lambda: collections.defaultdict(float))
try:
    rows = list(csv.DictReader(open(os.path.join(D, "pmc", "c_counter_collection.csv"))))
    # keep the dispatches of the last solve (dispatch ids after the last sym_from_lower)
    last = max(int(r["Dispatch_Id"]) for r in rows if "sym_from_lower" in r["Kernel_Name"])
    for r in rows:
        if int(r["Dispatch_Id"]) >= last:
            k = short(r["Kernel_Name"])
            pm[k][r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                pm[k]["_ms"] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
except Exception as ex:   # noqa: BLE001
    print("no pmc data:", ex)
b, g = 64, 64
npan = len(range(0, n - b - 1, b))
ms = [n - j - b for j in range(0, n - b - 1, b)]
flops = {
    "dgemm_kernel<false, true, 4, 4, false>": ("rank-128 update of the band reduction, lower triangle: sum_panels 2 (m^2/2) 128", sum(2 * (m * m / 2) * 128 for m in ms)),
    "dgemm_kernel<false, false, 4, 2, true>": ("X = A22 V with A22 symmetric from its lower triangle: sum_panels 2 m^2 64", sum(2 * m * m * 64 for m in ms)),
    "bt2_apply_kernel": ("stage-2 reflector blocks on Z: per block 2 x (2 x 127 x 64 x n)", None),
}
ng = (n - 2 + g - 1) // g
nblk = sum(1 for G in range(ng) for k in range((n + b - 1) // b + 1) if G * g + 1 + k * b < n)
flops["bt2_apply_kernel"] = (flops["bt2_apply_kernel"][0], nblk * 2 * (2 * 127 * 64 * n))
out = {"note": f"MI355X, one pg_syevd_dev solve at n = {n} (two-stage), rocprofv3 --kernel-trace; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CYCLES-normalised "
               "CU cycles) from a separate --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8); clock = GRBM_GUI_ACTIVE / 8 / duration. "
               "useful_TF counts the algorithm's flops only; fp64 MFMA peak 78.6 TF.",
       "timing": open(os.path.join(D, "timing.txt")).read().splitlines()[-14:], "kernels": []}
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    e = {"kernel": k, "calls": c, "total_ms": round(t, 3)}
    if k in pm and pm[k].get("GRBM_GUI_ACTIVE"):
        cyc = pm[k]["GRBM_GUI_ACTIVE"] / 8.0
        e["mfma_busy"] = round(pm[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc), 4)
        e["clock_GHz"] = round(cyc / (pm[k]["_ms"] * 1e6), 3) if pm[k]["_ms"] else None
    if k in flops:
        e["what"] = flops[k][0]; e["useful_TF"] = round(flops[k][1] / (t * 1e-3) / 1e12, 2); e["frac_of_fp64_mfma_peak"] = round(e["useful_TF"] / 78.6, 3)
    out["kernels"].append(e)
out["kernels"] = out["kernels"][:24]
json.dump(out, open(os.path.join(R, "profiles", f"{tag}_syevd_summary.json"), "w"), indent=1)
print(json.dumps(out["kernels"][:8], indent=1))
