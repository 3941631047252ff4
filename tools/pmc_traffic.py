#!/usr/bin/env python3
"""HBM traffic per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE), following
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; on gfx950 FETCH_SIZE reports 1/2 of
the bytes of a wide (16 B/lane) coalesced streaming read, so it is doubled; WRITE_SIZE is taken as is.

    python tools/pmc_traffic.py gpurun_out/pmc_r1_FETCH_SIZE.csv gpurun_out/pmc_r1_WRITE_SIZE.csv profiles/pmc_traffic.json [note] [steps]

`steps`: how many steps of bench.py the counter passes ran (default 1): every row also gets dispatches_per_step, which bench.py prints
next to its own launches_per_step - "bytes per launch" only compares with the algorithmic figure when the two launch mixes agree.
"""
import csv
import json
import re
import sys


def load(path, col):
    out = {}
    for row in csv.DictReader(open(path)):
        name = row["kernel"].replace("rd::", "").replace("; ", ",")
        out[name] = (float(row[col]), int(row["dispatches"]))
    return out


def canon(name):
    m = re.match(r"conv_igemm_kernel<128,(\d+),\d,\d,(true|false)>", name)
    if m:
        return "conv_igemm_kernel<128x%s,%s>" % (m.group(1), "1x1" if m.group(2) == "true" else "kxk")
    m = re.match(r"conv_igemm_h3_kernel<(\d+),(\d+),\d,\d,(true|false)>", name)
    if m:
        return "conv_igemm_h3_kernel<%sx%s,%s>" % (m.group(1), m.group(2), "1x1" if m.group(3) == "true" else "kxk")
    m = re.match(r"lc_mixer_h3_kernel<(\d+),0>", name)
    if m:
        return "lc_mixer_h3_kernel<%s>" % m.group(1)
    m = re.match(r"lc_mixer_ws_kernel<(\d+),", name)
    if m:
        return "lc_mixer_ws_kernel<%s>" % m.group(1)
    if name.startswith("gemm_h1_kernel<"):
        return "gemm_h1_kernel"
    m = re.match(r"lc_mixer_res_kernel<(\d+),", name)
    if m:
        return "lc_mixer_res_kernel<%s>" % m.group(1)
    m = re.match(r"lc_mixer_kernel<(\d+),0>", name)
    if m:
        return "lc_mixer_kernel<%s>" % m.group(1)
    return name   # lc_mixer_h3_kernel<C>, ctc_head_h3_kernel, ... already match bench.py's names


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
acc = {}
for k in fetch:
    if k not in write:
        continue
    f, n = fetch[k]
    w, _ = write[k]
    a = acc.setdefault(canon(k), [0, 0.0, 0.0])     # template variants of one kernel (gated / ungated) are pooled
    a[0] += n; a[1] += f; a[2] += w
res = {"_collected": sys.argv[4] if len(sys.argv) > 4 else "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes"}
steps = int(sys.argv[5]) if len(sys.argv) > 5 else 1
for k, (n, f, w) in acc.items():
    res[k] = {"dispatches": n, "dispatches_per_step": n / steps, "fetch_KiB_raw": f, "write_KiB_raw": w,
              "hbm_bytes_per_launch": round((2.0 * f + w) * 1024.0 / n),
              "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as counted"}
json.dump(res, open(sys.argv[3], "w"), indent=1)
print(json.dumps({k: v["hbm_bytes_per_launch"] for k, v in list(res.items())[1:9]}, indent=1))
