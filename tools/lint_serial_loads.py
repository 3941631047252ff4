#!/usr/bin/env python3
"""ISA lint: loads that are waited for one by one.  `x = ok ? p[i] : 0` compiles to `global_load ; s_waitcnt vmcnt(0) ; v_cndmask`
per element, so a run of conditional loads is a run of serial memory round trips.  Compiles every .hip under
rapiddoc_amd/csrc to gfx950 assembly (no GPU needed) and reports, per kernel, how many loads are followed within five instructions by
an `s_waitcnt vmcnt(0)` before another load is issued.  Usage: python tools/lint_serial_loads.py [min_pairs]"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rapiddoc_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"
min_pairs = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for src in sorted(glob.glob(os.path.join(ROOT, "*.hip"))):
    out = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only", f"-I{ROOT}", src, "-o", "-"],
                         capture_output=True, text=True).stdout
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\n\s*\.end_amdhsa_kernel", out, re.S | re.M):
        name, body = m.group(1), m.group(2).split("\n")
        ins = [l.strip() for l in body if l.strip() and not l.strip().startswith((";", "."))]
        pairs = loads = 0
        for i, ls in enumerate(ins):
            if ls.startswith(("global_load_dword", "buffer_load_dword")) and "lds" not in ls:
                loads += 1
                for nx in ins[i + 1: i + 6]:
                    if nx.startswith(("global_load", "buffer_load")):
                        break
                    if nx.startswith("s_waitcnt") and "vmcnt(0)" in nx:
                        pairs += 1
                        break
        if pairs >= min_pairs:
            print(f"{os.path.basename(src):28s} {name[:90]:90s} loads {loads:4d}  waited-at-once {pairs}")
