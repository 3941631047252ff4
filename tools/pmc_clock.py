#!/usr/bin/env python3
"""Effective clock per kernel from a rocprofv3 `--kernel-trace --pmc GRBM_GUI_ACTIVE` pass (the power-bound reading of the
matrix kernels was an inference from s_memtime stamps): GRBM_GUI_ACTIVE counts the graphics clock's cycles while the GPU is busy, a counter
pass runs the dispatches one at a time, so  sum(GRBM_GUI_ACTIVE) / sum(kernel duration)  over a kernel's dispatches is the clock that kernel
gets when it has the chip to itself.  The counter is reported summed over the XCDs on some stacks: the table carries the raw ratio and the
ratio divided by the XCD count that makes the bandwidth-bound kernels (which do not throttle) land at the chip's peak clock.
Usage: pmc_clock.py <counter_collection.csv> <kernel_trace.csv> <out.csv>"""
import csv
import sys
from collections import defaultdict

cc, kt, out = sys.argv[1], sys.argv[2], sys.argv[3]
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
agg = defaultdict(lambda: [0, 0.0, 0.0])
for did, c in cyc.items():
    if did not in dur:
        continue
    ns, name = dur[did]
    k = name.split("(")[0].replace("void ", "")
    a = agg[k]
    a[0] += 1
    a[1] += c
    a[2] += ns
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
# XCD normalisation: the highest raw ratio among kernels with >= 1 ms of total time is taken to be a kernel running at the peak clock
raw = {k: (v[1] / v[2] if v[2] else 0.0) for k, v in rows}
top = max([r for k, r in raw.items() if agg[k][2] > 1e6] or [1.0])
PEAK = 2.4
nx = max(1, round(top / PEAK))
with open(out, "w") as f:
    f.write(f"# GRBM_GUI_ACTIVE / duration per kernel; divided by {nx} (counter summed over {nx} XCDs: highest raw ratio {top:.2f} cycles/ns)\n")
    f.write("kernel,dispatches,total_ms,gui_active_cycles,cycles_per_ns_raw,effective_GHz\n")
    for k, v in rows:
        f.write(f"{k.replace(',', ';')},{v[0]},{v[2] / 1e6:.3f},{v[1]:.6g},{raw[k]:.4f},{raw[k] / nx:.3f}\n")
print(open(out).read()[:4000])
