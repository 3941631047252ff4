#!/bin/bash
# GPU box: wall-clock table of tools/probe_mfma_valu2.hip + SQ counters per arm (two rocprofv3 --pmc passes, kernel-trace only).
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 $R/tools/probe_mfma_valu2.hip -o /tmp/pmv2 || exit 1
/tmp/pmv2 > $O/r4_probe2_wall.txt 2>&1
cat $O/r4_probe2_wall.txt
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z0-9_]*" | sort -u > $O/r4_sq_counters.txt
pick() { for c in "$@"; do grep -qx "$c" $O/r4_sq_counters.txt && echo -n "$c "; done; }
S1=$(pick SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_WAIT_INST_ANY)
S2=$(pick SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES)
echo "pass 1: $S1"; echo "pass 2: $S2"
for n in 1 2; do
  eval S=\$S$n
  rm -rf /tmp/pp$n
  timeout 300 rocprofv3 --kernel-trace --pmc $S --output-format csv -d /tmp/pp$n -o a -- /tmp/pmv2 > /tmp/pp$n.log 2>&1
  f=$(find /tmp/pp$n -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" $O/r4_probe2_pmc$n.csv <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.OrderedDict()
for r in rows:
    k = r.get("Kernel_Name") or r.get("Kernel Name")
    d = int(r.get("Dispatch_Id", 0))
    agg.setdefault((k, d), {})[r["Counter_Name"]] = agg.get((k, d), {}).get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
# two dispatches per arm (10 warm-up iterations, then 4000): keep the larger one per kernel
best = collections.OrderedDict()
for (k, d), c in agg.items():
    tot = sum(c.values())
    if k not in best or tot > best[k][0]:
        best[k] = (tot, c)
names = sorted({n for _t, c in best.values() for n in c})
with open(sys.argv[2], "w") as f:
    f.write("kernel," + ",".join(names) + "\n")
    for k, (_t, c) in best.items():
        f.write(k.replace(",", ";") + "," + ",".join("%.0f" % c.get(n, 0) for n in names) + "\n")
print(open(sys.argv[2]).read())
PY
done
