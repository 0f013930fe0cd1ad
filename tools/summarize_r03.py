#!/usr/bin/env python3
"""gpurun_out/prof_r03/ (tools/profile_r03.sh) -> the committed summaries under profiles/: kernel-trace stats per run
(r03_*_kernel_stats.csv) and the HBM traffic of the fused mid-size kernel (r03_mid48_pmc_summary.json; FETCH_SIZE corrected
with the factor calibrated by tools/fetch_calib.hip on this kernel family's 8-byte-per-lane loads, WRITE_SIZE as is)."""
import csv, glob, json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "prof_r03"); dst = os.path.join(R, "profiles")
def latest(p):
    f = sorted(glob.glob(os.path.join(src, p)), key=os.path.getmtime)
    return f[-1] if f else None
def rows(p):
    f = latest(p); return list(csv.DictReader(open(f))) if f else []
for run, name in (("trace", "r03_kernel_stats.csv"), ("trace20", "r03_trace20_kernel_stats.csv"), ("mid33", "r03_mid33_kernel_stats.csv"),
                  ("mid48", "r03_mid48_kernel_stats.csv"), ("mid64", "r03_mid64_kernel_stats.csv"), ("wg2_96", "r03_wg2_96_kernel_stats.csv"),
                  ("wg2_128", "r03_wg2_128_kernel_stats.csv"), ("small16", "r03_small16_kernel_stats.csv"),
                  ("explicit64", "r03_explicit64_kernel_stats.csv"), ("trace5", "r03_trace5_kernel_stats.csv"),
                  ("trace5_route0", "r03_trace5_route0_kernel_stats.csv")):
    st = rows(f"{run}/*/*kernel_stats.csv")
    if not st:
        continue
    with open(os.path.join(dst, name), "w") as fh:
        w = csv.writer(fh); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in st:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    log = os.path.join(src, run + ".log")
    if os.path.exists(log):
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith("{") or ln.startswith("n=m=") or ln.startswith("n=")]
        if lines:
            open(os.path.join(dst, name.replace("_kernel_stats.csv", "_line.txt")), "w").write("\n".join(lines) + "\n")
def pmc(p, counter, kname):
    return [float(r["Counter_Value"]) for r in rows(p) if r["Counter_Name"] == counter and kname in r["Kernel_Name"]]
cal = {}
v = pmc("calib_fetch/*/*counter_collection.csv", "FETCH_SIZE", "calib_read8")
if v: cal["read8"] = (1 << 30) / 1024.0 / v[0]
v = pmc("calib_write/*/*counter_collection.csv", "WRITE_SIZE", "calib_write8")
if v: cal["write8"] = (64 << 20) / 1024.0 / v[0]
f = pmc("pmc_fetch_mid48/*/*counter_collection.csv", "FETCH_SIZE", "avi_solve_schur48")
wv = pmc("pmc_write_mid48/*/*counter_collection.csv", "WRITE_SIZE", "avi_solve_schur48")
summ = {"calibration_factor_known_over_reported": cal}
if f and wv:
    fk = sum(f) / len(f); wk = sum(wv) / len(wv)
    n = m = 48; p = 8; nodes = 4000
    rec = 8 * (n * n + n * p + n + m * n + m * p + 2 * m) * nodes
    outb = (8 * (n + m) + (n + m) + 8 + 4 + 4) * nodes
    summ["avi_solve_schur48 (n = m = 48, 4000 nodes per launch)"] = {
        "launches": len(f), "FETCH_SIZE_KiB_raw": fk, "FETCH_bytes_corrected": fk * cal.get("read8", 2.0) * 1024.0,
        "WRITE_SIZE_bytes": wk * cal.get("write8", 1.0) * 1024.0,
        "hbm_bytes_per_launch": (fk * cal.get("read8", 2.0) + wk * cal.get("write8", 1.0)) * 1024.0,
        "records_bytes_per_launch": rec, "outputs_bytes_per_launch": outb}
sq = {}
for r in rows("pmc_sq_mid48/*/*counter_collection.csv"):
    if "avi_solve_schur48" in r["Kernel_Name"]:
        sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if sq:
    summ["avi_solve_schur48_sq_per_launch"] = {k: sum(v) / len(v) for k, v in sq.items()}
json.dump(summ, open(os.path.join(dst, "r03_mid48_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summ, indent=1)[:3000])

# ---- the bench kernel's counters (same layout as profiles/r02_pmc_summary.json: bench.py reads it)
import subprocess
b = {"calibration_factor_known_over_reported": cal}
f = pmc("pmc_fetch/*/*counter_collection.csv", "FETCH_SIZE", "avi_solve_schur")
wv = pmc("pmc_write/*/*counter_collection.csv", "WRITE_SIZE", "avi_solve_schur")
if f and wv:
    fk = sum(f) / len(f); wk = sum(wv) / len(wv)
    hb = (fk * cal.get("read8", 2.0) + wk * cal.get("write8", 1.0)) * 1024.0
    b["avi_solve_schur"] = {"launches": len(f), "FETCH_SIZE_KiB_raw": fk, "FETCH_KiB_corrected": fk * cal.get("read8", 2.0), "WRITE_SIZE_KiB": wk,
                            "hbm_bytes_per_launch": hb}
    b["avi_solve_hbm_bytes_per_launch"] = hb
sqb = {}
for pat in ("pmc_sq/*/*counter_collection.csv", "pmc_sq_b/*/*counter_collection.csv", "pmc_sq_c/*/*counter_collection.csv"):
    for r in rows(pat):
        if "avi_solve_schur" in r["Kernel_Name"]:
            sqb.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
if sqb:
    b["avi_solve_schur_sq_per_launch"] = {k: sum(v) / len(v) for k, v in sqb.items()}
try:
    b["commit"] = subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    pass
if "avi_solve_schur" in b and "avi_solve_schur_sq_per_launch" in b:
    json.dump(b, open(os.path.join(dst, "r03_pmc_summary.json"), "w"), indent=1)
    print("wrote r03_pmc_summary.json:", b["avi_solve_schur"])
