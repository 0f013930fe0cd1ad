#!/usr/bin/env python3
"""gpurun_out/prof_r04/ (tools/profile_r04.sh) -> the committed summaries under profiles/: kernel-trace stats per run
(r04_*_kernel_stats.csv, the run's own output lines next to them) and counter summaries per kernel family
(r04_*_pmc_summary.json: FETCH_SIZE corrected with the factor calibrated by tools/fetch_calib.hip on 8-byte-per-lane loads,
as /opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950; WRITE_SIZE as is; SQ counters averaged per launch)."""
import csv, glob, json, os, subprocess
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(R, "gpurun_out", "prof_r04"); dst = os.path.join(R, "profiles")


def latest(p):
    f = sorted(glob.glob(os.path.join(src, p)), key=os.path.getmtime)
    return f[-1] if f else None


def rows(p):
    f = latest(p)
    return list(csv.DictReader(open(f))) if f else []


def copy_stats(run, name, keep=("{", "verify ", "n=")):
    st = rows(f"{run}/*/*kernel_stats.csv")
    if not st:
        return
    with open(os.path.join(dst, name), "w") as fh:
        w = csv.writer(fh); w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
        for r in st:
            w.writerow([r["Name"][:140], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"], r["MaxNs"]])
    log = os.path.join(src, run + ".log")
    if os.path.exists(log):
        lines = [ln for ln in open(log).read().splitlines() if ln.startswith(keep)]
        if lines:
            open(os.path.join(dst, name.replace("_kernel_stats.csv", "_line.txt")), "w").write("\n".join(lines) + "\n")


def pmc(p, counter, kname):
    return [float(r["Counter_Value"]) for r in rows(p) if r["Counter_Name"] == counter and kname in r["Kernel_Name"]]


cal = {}
v = pmc("calib_fetch/*/*counter_collection.csv", "FETCH_SIZE", "calib_read8")
if v: cal["read8"] = (1 << 30) / 1024.0 / v[0]
v = pmc("calib_write/*/*counter_collection.csv", "WRITE_SIZE", "calib_write8")
if v: cal["write8"] = (64 << 20) / 1024.0 / v[0]
try:
    commit = subprocess.check_output(["git", "-C", R, "rev-parse", "--short", "HEAD"], text=True).strip()
except Exception:
    commit = None


def family(tag, kname, extra=None):
    out = {"calibration_factor_known_over_reported": cal, "kernel": kname, "commit": commit}
    f = pmc(f"{tag}_fetch/*/*counter_collection.csv", "FETCH_SIZE", kname)
    wv = pmc(f"{tag}_write/*/*counter_collection.csv", "WRITE_SIZE", kname)
    if f and wv:
        fk = sum(f) / len(f); wk = sum(wv) / len(wv)
        out["hbm"] = {"launches": len(f), "FETCH_SIZE_KiB_raw": fk, "FETCH_bytes_corrected": fk * cal.get("read8", 2.0) * 1024.0,
                      "WRITE_SIZE_bytes": wk * cal.get("write8", 1.0) * 1024.0,
                      "hbm_bytes_per_launch": (fk * cal.get("read8", 2.0) + wk * cal.get("write8", 1.0)) * 1024.0}
    sq = {}
    for pat in (f"{tag}_sq/*/*counter_collection.csv", f"{tag}_sq_b/*/*counter_collection.csv", f"{tag}_sq_c/*/*counter_collection.csv"):
        for r in rows(pat):
            if kname in r["Kernel_Name"]:
                sq.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    if sq:
        out["sq_per_launch"] = {k: sum(v) / len(v) for k, v in sq.items()}
    if extra:
        out.update(extra)
    return out


# ---- A8 verify
for run, name in (("verify_trace", "r04_verify_kernel_stats.csv"), ("verify_trace_mode0", "r04_verify_solution_kernel_stats.csv"),
                  ("verify_trace_mode2", "r04_verify_fallback_kernel_stats.csv"), ("verify_trace_48x48", "r04_verify_48_kernel_stats.csv"),
                  ("verify_trace_64x64", "r04_verify_64_kernel_stats.csv"), ("verify_trace_64x96", "r04_verify_64x96_kernel_stats.csv"),
                  ("verify_trace_256x256", "r04_verify_256_kernel_stats.csv"), ("verify_trace_128x200", "r04_verify_128x200_kernel_stats.csv"),
                  ("trace", "r04_kernel_stats.csv"),
                  ("trace20", "r04_trace20_kernel_stats.csv"), ("trace5", "r04_trace5_kernel_stats.csv")):
    copy_stats(run, name)
for f_ in ("verify_rate.txt", "verify_rate_other.txt"):
    p = os.path.join(src, f_)
    if os.path.exists(p):
        lines = [ln for ln in open(p).read().splitlines() if ln.startswith("verify ")]
        open(os.path.join(dst, "r04_" + f_), "w").write("\n".join(lines) + "\n")
if latest("verify_fetch/*/*counter_collection.csv"):
    n = m = 32; p = 8; nodes = 10000
    rec = 8 * (n * n + n * p + n + m * n + m * p + 2 * m + n + p) * nodes
    outb = (8 * m + 8) * nodes
    s = family("verify", "verify_node32", {"workload": "10 000 nodes, n = m = 32, p = 8, at the solution (least-squares path on every node)",
                                           "records_and_point_bytes_per_launch": rec, "outputs_bytes_per_launch": outb})
    json.dump(s, open(os.path.join(dst, "r04_verify_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(s, indent=1)[:2500])
if latest("bench_fetch/*/*counter_collection.csv"):
    s = family("bench", "avi_solve_schur")
    # (same keys as profiles/r03_pmc_summary.json: bench.py reads them)
    if "hbm" in s:
        s["avi_solve_schur"] = {"launches": s["hbm"]["launches"], "FETCH_SIZE_KiB_raw": s["hbm"]["FETCH_SIZE_KiB_raw"],
                                "FETCH_KiB_corrected": s["hbm"]["FETCH_bytes_corrected"] / 1024.0, "WRITE_SIZE_KiB": s["hbm"]["WRITE_SIZE_bytes"] / 1024.0,
                                "hbm_bytes_per_launch": s["hbm"]["hbm_bytes_per_launch"]}
        s["avi_solve_hbm_bytes_per_launch"] = s["hbm"]["hbm_bytes_per_launch"]
    if "sq_per_launch" in s:
        s["avi_solve_schur_sq_per_launch"] = s["sq_per_launch"]
    json.dump(s, open(os.path.join(dst, "r04_pmc_summary.json"), "w"), indent=1)
if latest("c5_fetch/*/*counter_collection.csv"):
    import re
    out = {"calibration_factor_known_over_reported": cal, "commit": commit,
           "workload": "bench.py --config 5: 512 nodes, n = m = 256, one sweep = the launches below", "kernels": {}}
    shorts = sorted({re.search(r"(schur_big\w*(?:<\d+>)?)", r["Kernel_Name"]).group(1)
                     for r in rows("c5_fetch/*/*counter_collection.csv") if "schur_big" in r["Kernel_Name"]})
    # per SWEEP: a kernel's bytes summed over all its launches, divided by the number of sweeps (= launches of the finish kernel;
    # some kernels run twice a sweep, the Lemke kernel's launches are empty when block principal pivoting finished every node)
    sweeps = max(1, len(pmc("c5_fetch/*/*counter_collection.csv", "FETCH_SIZE", "schur_big2_finish(")))
    out["sweeps_profiled"] = sweeps
    total = 0.0
    for kn in shorts:
        fam = family("c5", kn + "(")
        ent = {k: v for k, v in fam.items() if k in ("hbm", "sq_per_launch")}
        f = pmc("c5_fetch/*/*counter_collection.csv", "FETCH_SIZE", kn + "(")
        wv = pmc("c5_write/*/*counter_collection.csv", "WRITE_SIZE", kn + "(")
        if f and wv:
            ent["launches_per_sweep"] = len(f) / sweeps
            ent["hbm_bytes_per_sweep"] = (sum(f) * cal.get("read8", 2.0) + sum(wv) * cal.get("write8", 1.0)) * 1024.0 / sweeps
            total += ent["hbm_bytes_per_sweep"]
        out["kernels"][kn] = ent
    out["hbm_bytes_per_sweep"] = total
    json.dump(out, open(os.path.join(dst, "r04_c5_pmc_summary.json"), "w"), indent=1)
    print(json.dumps({k: v.get("hbm_bytes_per_sweep") for k, v in out["kernels"].items()}, indent=1), total)

# ---- mid-size classes and the wide verify: counters per launch of the class's kernel
for n_, kname, nodes in ((48, "avi_solve_schur48", 4000), (64, "schur_wg_nodes", 4000), (96, "schur_wg2_nodes", 1024), (128, "schur_wg2_nodes", 1024)):
    copy_stats(f"mid{n_}_trace", f"r04_mid{n_}_kernel_stats.csv")
    if latest(f"mid{n_}_fetch/*/*counter_collection.csv"):
        rec = 8 * (n_ * n_ + n_ * 8 + n_ + n_ * n_ + n_ * 8 + 2 * n_) * nodes
        s_ = family(f"mid{n_}", kname, {"workload": f"{nodes} resident node records, n = m = {n_}, p = 8 (tools/mid_rate.py)",
                                        "records_bytes_per_launch": rec, "outputs_bytes_per_launch": (8 * 2 * n_ + 2 * n_ + 16 + 8 * n_) * nodes})
        json.dump(s_, open(os.path.join(dst, f"r04_mid{n_}_pmc_summary.json"), "w"), indent=1)
if latest("vwide_fetch/*/*counter_collection.csv"):
    n_ = 256; nodes = 512
    s_ = family("vwide", "verify_wide_node", {"workload": "512 nodes, n = m = 256, p = 8, at the solution (least-squares path on every node)",
                                              "records_and_point_bytes_per_launch": 8 * (2 * n_ * n_ + 2 * n_ * 8 + 3 * n_ + n_ + 8) * nodes,
                                              "outputs_bytes_per_launch": (8 * n_ + 8) * nodes})
    json.dump(s_, open(os.path.join(dst, "r04_verify_wide_pmc_summary.json"), "w"), indent=1)
