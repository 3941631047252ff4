#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: sum of each counter and dispatch count."""
import csv
import sys
from collections import defaultdict

path, out = sys.argv[1], sys.argv[2]
agg = defaultdict(lambda: defaultdict(float))
calls = defaultdict(set)
with open(path) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0].replace("void ", "")
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        calls[k].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
names = sorted({c for v in agg.values() for c in v})
with open(out, "w") as f:
    f.write("kernel,dispatches," + ",".join(names) + "\n")
    for k in sorted(agg, key=lambda k: -sum(agg[k].values())):
        f.write(k.replace(",", ";") + f",{len(calls[k])}," + ",".join("%.6g" % agg[k].get(c, 0) for c in names) + "\n")
print(open(out).read()[:3000])
