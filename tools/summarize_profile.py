#!/usr/bin/env python3
"""Turns gpurun_out/prof_rNN/ (tools/profile_r01.sh) into the committed summaries under profiles/:
kernel-trace stats, PMC traffic with the gfx950 corrections of MI355X_MICROARCH.md (FETCH_SIZE x2
for coalesced streaming reads -- re-calibrated here for this kernel's 8 B/lane loads with
tools/fetch_calib.hip -- WRITE_SIZE as is; both reported in KiB), and profiles/traffic.json."""
import collections, csv, glob, json, os, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "prof_" + tag)
dst = os.path.join(R, "profiles"); os.makedirs(dst, exist_ok=True)
def latest(p):
    f = sorted(glob.glob(os.path.join(src, p)), key=os.path.getmtime)
    return f[-1] if f else None
def rows(p):
    f = latest(p); return list(csv.DictReader(open(f))) if f else []
out = []
st = rows("trace/runc/*kernel_stats.csv")
with open(os.path.join(dst, f"{tag}_kernel_stats.csv"), "w") as fh:
    w = csv.writer(fh); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in st:
        w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
def pmc(p, counter, kname):
    return [float(r["Counter_Value"]) for r in rows(p) if r["Counter_Name"] == counter and kname in r["Kernel_Name"]]
cal = {}
for k, known in (("calib_read8", 1 << 30), ("calib_read16", 1 << 30)):
    v = pmc("calib_fetch/runc/*counter_collection.csv", "FETCH_SIZE", k)
    if v: cal[k] = known / 1024.0 / v[0]
v = pmc("calib_write/runc/*counter_collection.csv", "WRITE_SIZE", "calib_write8")
if v: cal["calib_write8"] = (64 << 20) / 1024.0 / v[0]
summ = {"calibration_factor_known_over_reported": cal}
for kern in ("avi_solve_schur", "avi_solve_reg", "assemble_nodes"):
    f = pmc("pmc_fetch/runc/*counter_collection.csv", "FETCH_SIZE", kern)
    wv = pmc("pmc_write/runc/*counter_collection.csv", "WRITE_SIZE", kern)
    if f and wv:
        fk = sum(f) / len(f); wk = sum(wv) / len(wv)
        fcorr = fk * cal.get("calib_read8", 2.0)
        summ[kern] = {"launches": len(f), "FETCH_SIZE_KiB_raw": fk, "FETCH_KiB_corrected": fcorr, "WRITE_SIZE_KiB": wk,
                      "hbm_bytes_per_launch": (fcorr + wk * cal.get("calib_write8", 1.0)) * 1024.0}
DOM = "avi_solve_schur"      # the dominant kernel of the bench step (fused node path)
sq = collections.defaultdict(list)
for sub in ("pmc_sq", "pmc_sq_b", "pmc_sq_c"):
    for r in rows(sub + "/runc/*counter_collection.csv"):
        if DOM in r["Kernel_Name"]: sq[r["Counter_Name"]].append(float(r["Counter_Value"]))
summ[DOM + "_sq_per_launch"] = {k: sum(v) / len(v) for k, v in sq.items()}
import subprocess
try:
    summ["commit"] = subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"]).decode().strip()
except Exception:
    pass
if DOM in summ:
    summ["avi_solve_hbm_bytes_per_launch"] = summ[DOM]["hbm_bytes_per_launch"]
for extra in ("trace20", "trace5"):                       # the driver's own command; the config-5 line
    st2 = rows(extra + "/runc/*kernel_stats.csv")
    if st2:
        with open(os.path.join(dst, f"{tag}_{extra}_kernel_stats.csv"), "w") as fh:
            w = csv.writer(fh); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
            for r in st2:
                w.writerow([r["Name"][:120], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
for lg in ("trace", "trace20", "trace5"):                 # the bench lines those traced runs printed themselves
    f = os.path.join(src, lg + ".log")
    if os.path.exists(f):
        for line in open(f):
            if line.startswith("{"):
                open(os.path.join(dst, f"{tag}_{lg}_bench_line.json"), "w").write(line)
json.dump(summ, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
if DOM in summ:
    json.dump({"avi_solve_hbm_bytes_per_launch": summ[DOM]["hbm_bytes_per_launch"], "kernel": DOM,
               "source": f"profiles/{tag}_pmc_summary.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; "
                         "FETCH_SIZE x calibrated factor)", "solves_per_launch": 10000},
              open(os.path.join(dst, "traffic.json"), "w"), indent=1)
print(json.dumps(summ, indent=1)[:3000])
