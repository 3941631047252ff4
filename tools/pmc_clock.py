#!/usr/bin/env python3
"""Effective clock per kernel from a rocprofv3 `--kernel-trace --pmc GRBM_GUI_ACTIVE` pass (the power-bound reading of the
matrix kernels was an inference from s_memtime stamps).  GRBM_GUI_ACTIVE counts graphics-clock cycles while the GPU is busy; a counter pass
runs the dispatches one at a time.  Per dispatch the counter covers the kernel PLUS a fixed window around it (counter start / stop), and it is
reported summed over the chip's XCDs, so a plain  cycles / duration  over-reads short kernels (se_fc, 11 us: 34 cycles/ns).  A kernel family
whose dispatches differ in length gives both numbers: the least-squares line  cycles = slope * duration + intercept  over its dispatches has
slope = XCDs * clock and intercept = XCDs * clock * window.  Families with too little spread in duration get the window of the best-fitted
family instead.  Usage: pmc_clock.py <counter_collection.csv> <kernel_trace.csv> <out.csv>"""
import csv
import sys
from collections import defaultdict

cc, kt, out = sys.argv[1], sys.argv[2], sys.argv[3]
XCDS = 8
dur = {}
with open(kt) as f:
    for r in csv.DictReader(f):
        did = r.get("Dispatch_Id") or r.get("Correlation_Id")
        dur[did] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
cyc = defaultdict(float)
with open(cc) as f:
    for r in csv.DictReader(f):
        if r["Counter_Name"] != "GRBM_GUI_ACTIVE":
            continue
        did = r.get("Dispatch_Id") or r.get("Correlation_Id")
        cyc[did] += float(r["Counter_Value"])
pts = defaultdict(list)
for did, c in cyc.items():
    if did in dur:
        ns, name = dur[did]
        pts[name.split("(")[0].replace("void ", "")].append((float(ns), c))


def fit(p):
    n = len(p)
    sx = sum(a for a, _ in p); sy = sum(b for _, b in p)
    sxx = sum(a * a for a, _ in p); sxy = sum(a * b for a, b in p)
    den = n * sxx - sx * sx
    if n < 8 or den <= 0:
        return None
    slope = (n * sxy - sx * sy) / den
    icpt = (sy - slope * sx) / n
    lo, hi = min(a for a, _ in p), max(a for a, _ in p)
    if hi < 2.0 * lo or slope <= 0:
        return None
    ss_res = sum((b - slope * a - icpt) ** 2 for a, b in p)
    ss_tot = sum((b - sy / n) ** 2 for _, b in p) or 1.0
    return slope, icpt, 1.0 - ss_res / ss_tot


fits = {k: fit(p) for k, p in pts.items()}
# a usable fit: tight (R^2 > 0.97) with a positive intercept (the window around a dispatch cannot be negative)
good = {k: v for k, v in fits.items() if v and v[2] > 0.97 and v[1] > 0}
# the window (ns) for families without a usable fit: the median over the usable fits' own windows
wins = sorted(v[1] / v[0] for v in good.values())
window_ns = wins[len(wins) // 2] if wins else 0.0
ref = "the median of %d fitted families" % len(wins)
rows = sorted(pts.items(), key=lambda kv: -sum(a for a, _ in kv[1]))
with open(out, "w") as f:
    f.write(f"# GRBM_GUI_ACTIVE (summed over {XCDS} XCDs) vs kernel duration, one counter pass of the bench's step, dispatches serialised by the profiler.\n")
    f.write(f"# fit: cycles = slope * ns + intercept per family (>= 8 dispatches, durations spread >= 2x, R^2 > 0.98); clock = slope / {XCDS}.\n")
    f.write(f"# no fit: clock = cycles / (ns + window) / {XCDS} with the window of {ref}: {window_ns:.0f} ns.\n")
    f.write("kernel,dispatches,total_ms,avg_us,raw_cycles_per_ns,fit_R2,effective_GHz,method\n")
    for k, p in rows:
        tns = sum(a for a, _ in p); tc = sum(b for _, b in p)
        v = good.get(k)
        if v:
            ghz, r2, how = v[0] / XCDS, f"{v[2]:.4f}", "fit"
        else:
            fv = fits.get(k)
            ghz, r2, how = tc / (tns + window_ns * len(p)) / XCDS, (f"{fv[2]:.4f}" if fv else ""), "window"
        f.write(f"{k.replace(',', ';')},{len(p)},{tns / 1e6:.3f},{tns / len(p) / 1e3:.1f},{tc / tns:.3f},{r2},{ghz:.3f},{how}\n")
with open(out.replace(".csv", "_points.csv"), "w") as f:
    f.write("kernel,duration_ns,gui_active_cycles\n")
    for k, p in rows[:10]:
        for a, b in p:
            f.write(f"{k.replace(',', ';')},{a:.0f},{b:.0f}\n")
print(open(out).read()[:5000])
