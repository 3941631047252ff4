#!/bin/bash
# A/B of environment switches on the bench's steady state (GPU box): one warm-up process, then every configuration twice, interleaved.
# usage: bash tools/ab_env_bench.sh "RD_A=1 RD_B=0" "RD_A=0" ...   (each argument = the env assignments of one configuration, "" = defaults)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
run() { env $1 RD_BENCH_STEP_TIMES=1 python bench.py --steps 16 --warmup 6 --no-extra-passes --no-cpu-baseline 2> /tmp/ab.err > /tmp/ab.json
  python - "$1" <<'PY'
import json, re, sys
r = json.load(open("/tmp/ab.json"))
steps = [float(x) for x in re.findall(r"step ([0-9.]+) ms", open("/tmp/ab.err").read())]
tail = sorted(steps[-10:])
print("%-40s value %7.2f  ms/step %6.2f  median of last 10 steps %6.2f  min %6.2f" % (sys.argv[1] or "(defaults)", r["value"], r["ms_per_step"], tail[len(tail) // 2], tail[0]))
PY
}
env RD_BENCH_STEP_TIMES=0 python bench.py --steps 10 --warmup 3 --no-extra-passes --no-cpu-baseline > /dev/null 2>&1
for rep in 1 2; do for cfg in "$@"; do run "$cfg"; done; done
