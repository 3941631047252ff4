#!/usr/bin/env python3
"""Fill the R4_* placeholders of DESIGN.md's round-4 measurement paragraph from profiles/r4_*.json (run after tools/collect_profiles.sh r4
and the copy into profiles/)."""
import csv
import json
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
r = json.loads((ROOT / "profiles" / "r4_bench.json").read_text())
res = json.loads((ROOT / "profiles" / "r4_bench_resident_one_set.json").read_text())
c, roof, bb, f = r["config"], r["roofline"], r["backbone"], r["formula"]
lines = (ROOT / "profiles" / "r4_bench_per_kernel.csv").read_text().splitlines()
keys = lines[0].split(",")
rows = [dict(zip(keys, l.rsplit(",", len(keys) - 1))) for l in lines[1:10]]       # (kernel names may hold commas: split from the right)
table = " · ".join("%s %.1f (%s)" % (x["kernel"].replace("_kernel", ""), float(x["total_ms"]),
                                     ("%.0f TFLOP/s" % float(x["TFLOPs"])) if float(x["TFLOPs"]) > 20 else ("%.1f TB/s" % (float(x["GBs"]) / 1e3)))
                   for x in rows)
sub = {
    "R4_VALUE": "%.1f" % r["value"], "R4_MS": "%.2f" % r["ms_per_step"], "R4_THR": "%.1f" % r["throughput_rec_batching"]["pages_s"],
    "R4_FP32": "%.1f" % r["fp32_precision_mode"]["pages_s"], "R4_BBTF": "%.1f" % bb["roofline"]["achieved"], "R4_BBFRAC": "%.3f" % bb["roofline"]["frac"],
    "R4_BB": "%.0f" % bb["pages_s"], "R4_FENC": "%.0f" % f["encoder_b32"]["tflops"], "R4_FDEC8": "%.3f" % f["decode_b8"]["ms_per_step"],
    "R4_FTOK32": "%.0f" % f["decode_b32"]["tokens_s"], "R4_RES": "%.1f" % res["value"], "R4_HOST": "%.1f" % c["host_ms_per_step_max_over_ranks"],
    "R4_H2D": "%.2f ms" % c["h2d_ms_per_batch_alone"], "R4_MISS": str(c["plan_cache_misses"]), "R4_CPU": "%.2f" % r["cpu_baseline"]["value"],
    "R4_DOMUS": "%.1f" % roof["avg_launch_us"], "R4_DOMTF": "%.1f" % roof["achieved"], "R4_DOMFRAC": "%.4f" % roof["frac"], "R4_DOM": roof["kernel"],
    "R4_TRAFFIC": ("%.0f" % (roof["traffic"] / 1e6)) if roof.get("traffic") else "n/a", "R4_TABLE": table,
}
p = ROOT / "DESIGN.md"
s = p.read_text()
for k in sorted(sub, key=len, reverse=True):
    s = s.replace(k, sub[k])
p.write_text(s)
print({k: v for k, v in sub.items() if k != "R4_TABLE"})
print(table)
