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
ms = [n - j - b for j in range(0, n - b - 1, b)]
ng = (n - 2 + g - 1) // g
nblk = sum(1 for G in range(ng) for k in range((n + b - 1) // b + 1) if G * g + 1 + k * b < n)
# stage-1 back-transformation: blocks of 256 reflectors (nref = 64 * panels), rows below the block's first unit
nref = len(ms) * b
bt1 = [(min(256, nref - i0), n - i0 - b) for i0 in range(0, nref, 256) if n - i0 - b > 0]


def phase_of(names):
    """split one solve's kernel list into its phases by position"""
    def first(x, d=None):
        return next((i for i, v in enumerate(names) if v.startswith(x)), d)
    i_be = first("band_extract_kernel")
    i_bc_end = max(i for i, v in enumerate(names) if v.startswith("bc_")) + 1
    bt2 = [i for i, v in enumerate(names) if v.startswith("bt2_apply")]
    return {"dense->band": (0, i_be), "band->tridiagonal": (i_be, i_bc_end), "divide&conquer (+ bt2_prep beside it)": (i_bc_end, bt2[0]),
            "back-transform 2": (bt2[0], bt2[-1] + 1), "back-transform 1": (bt2[-1] + 1, len(names))}


upd = ("rank-128 update A22 -= [V W][W V]', lower triangle: sum_panels 2 (m^2/2) 128", sum(2 * (m * m / 2) * 128 for m in ms))
symx = ("X = A22 V, A22 symmetric from its lower triangle: sum_panels 2 m^2 64", sum(2 * m * m * 64 for m in ms))
bt1a = ("W = V'Z and G = V'V per block of 256 reflectors", sum(2 * m * n * r + 2 * m * m * r for m, r in bt1))
bt1b = ("Z -= V (T W) and T W", sum(2 * r * n * m + 2 * m * m * n for m, r in bt1))
flops = {
    # r3 names (register-staged kernel) and r4 names (LDS-DMA ring kernel: <A k-major, symmetric A, 16-column tiles per wave>)
    ("dense->band", "dgemm_kernel<true, false, 4, 4, false>"): upd,
    ("dense->band", "dgemm_ring_kernel<true, false, 4>"): upd,
    ("dense->band", "dgemm_kernel<false, false, 4, 2, true>"): symx,
    ("dense->band", "dgemm_ring_kernel<false, true, 2>"): symx,
    ("back-transform 1", "dgemm_ring_kernel<true, false, 4>"): bt1a,
    ("back-transform 1", "dgemm_ring_kernel<false, false, 4>"): bt1b,
    ("back-transform 2", "bt2_apply4_kernel"): ("four reflector blocks per slab trip; flops EXECUTED = 46 of the 64 tile products of a zero-padded 128 x 64 block: per block 46/64 x 2 x (2 x 128 x 64 x n)",
                                               nblk * (46 / 64) * 2 * (2 * 128 * 64 * n)),
    ("back-transform 1", "dgemm_kernel<true, false, 4, 4, false>"): ("W = V'Z and G = V'V per block of 256 reflectors", sum(2 * m * n * r + 2 * m * m * r for m, r in bt1)),
    ("back-transform 1", "dgemm_kernel<false, false, 4, 4, false>"): ("Z -= V (T W) and T W", sum(2 * r * n * m + 2 * m * m * n for m, r in bt1)),
}
names = [short(r["Kernel_Name"]) for r in tr]
out = {"note": f"MI355X, one pg_syevd_dev solve at n = {n} (two-stage), rocprofv3 --kernel-trace, kernels grouped by phase (position in the stream); mfma_busy and clock "
               "are per kernel NAME over the whole solve, from a separate --pmc pass: SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8), clock = "
               "GRBM_GUI_ACTIVE / 8 / duration (both only for kernels averaging >= 50 us per dispatch).  useful_TF counts the flops named in `what` only; fp64 MFMA peak 78.6 TF.",
       "timing": open(os.path.join(D, "timing.txt")).read().splitlines()[-14:], "phases": []}
for ph, (a, z) in phase_of(names).items():
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in tr[a:z]:
        k = short(r["Kernel_Name"]); agg[k][0] += 1; agg[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
    span = (int(tr[z - 1]["End_Timestamp"]) - int(tr[a]["Start_Timestamp"])) / 1e6
    ks = []
    for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:10]:
        e = {"kernel": k, "calls": c, "total_ms": round(t, 3)}
        # GRBM_GUI_ACTIVE / 8 / duration reads high on dispatches of a few microseconds (the guide: reliable from ~0.3 ms on): the clock and the
        # busy fraction are given only for kernels that average >= 50 us per dispatch (rows with 2.7 ... 10 GHz)
        if k in pm and pm[k].get("GRBM_GUI_ACTIVE") and t / c >= 0.05:
            cyc = pm[k]["GRBM_GUI_ACTIVE"] / 8.0
            e["mfma_busy"] = round(pm[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (1024.0 * cyc), 4)
            e["clock_GHz"] = round(cyc / (pm[k]["_ms"] * 1e6), 3) if pm[k]["_ms"] else None
        if (ph, k) in flops:
            e["what"] = flops[(ph, k)][0]; e["useful_TF"] = round(flops[(ph, k)][1] / (t * 1e-3) / 1e12, 2); e["frac_of_fp64_mfma_peak"] = round(e["useful_TF"] / 78.6, 3)
        ks.append(e)
    out["phases"].append({"phase": ph, "span_ms": round(span, 2), "kernel_ms": round(sum(v[1] for v in agg.values()), 2), "kernels": ks})
json.dump(out, open(os.path.join(R, "profiles", f"{tag}_syevd_summary.json"), "w"), indent=1)
print(json.dumps([{p["phase"]: p["span_ms"]} for p in out["phases"]]))
