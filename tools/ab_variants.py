#!/usr/bin/env python3
"""Developer A/B of compile-time variants of the library.  HERE (no GPU): `python tools/ab_variants.py build tag=flag,flag ...` builds
tools/variants/lib.<tag>.so from the same sources.  On the GPU box: `python tools/ab_variants.py run tag ...` copies each variant over
the in-tree library of the box's scratch copy, runs `bench.py --no-extra-passes --no-cpu-baseline` and prints pages/s + the per-kernel table."""
import json
import shutil
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
VAR = ROOT / "tools" / "variants"
sys.path.insert(0, str(ROOT))


def main():
    mode, specs = sys.argv[1], sys.argv[2:]
    if mode == "build":
        from rapiddoc_amd import build as B
        VAR.mkdir(exist_ok=True)
        for spec in specs:
            tag, _, flags = spec.partition("=")
            B.build(force=True, verbose=True, extra_flags=tuple(f for f in flags.split(",") if f), out=VAR / f"lib.{tag}.so")
        return
    lib = ROOT / "rapiddoc_amd" / "librapiddoc_mi355.so"
    if mode == "cmd":            # python tools/ab_variants.py cmd "<shell command>" tag ...: the command once per variant library
        keep = lib.with_suffix(".so.keep")
        shutil.copy2(lib, keep)
        try:
            for tag in specs[1:]:
                shutil.copy2(keep if tag == "tree" else VAR / f"lib.{tag}.so", lib)
                lib.touch()
                r = subprocess.run(specs[0], shell=True, capture_output=True, text=True, cwd=ROOT)
                for line in (r.stdout + r.stderr[-600:] * (r.returncode != 0)).splitlines():
                    print(f"[{tag}] {line}")
        finally:
            shutil.copy2(keep, lib)
            keep.unlink()
        return
    keep = lib.with_suffix(".so.keep")
    shutil.copy2(lib, keep)
    extra = [a for a in specs if a.startswith("--")]
    try:
        for tag in [a for a in specs if not a.startswith("--")]:
            if tag != "tree":
                shutil.copy2(VAR / f"lib.{tag}.so", lib)
            else:
                shutil.copy2(keep, lib)
            lib.touch()
            prof = ROOT / "gpurun_out" / f"ab_{tag}.csv"
            r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "10", "--warmup", "3", "--no-extra-passes", "--no-cpu-baseline",
                                "--dump-profile", str(prof), *extra], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print(tag, "FAILED", r.stderr[-800:])
                continue
            j = json.loads(line[-1])
            print("%-10s %8.2f pages/s  %7.2f ms/step  crc %s  dominant %s frac %.3f (%.1f us)" % (
                tag, j["value"], j["ms_per_step"], j["config"]["result_crc32"], j["roofline"]["kernel"], j["roofline"]["frac"], j["roofline"]["avg_launch_us"]))
            rows = prof.read_text().splitlines()[1:9]
            for row in rows:
                k, n, ms = row.split(",")[:3]
                print("      %-46s %4s launches %8s ms" % (k[:46], n, ms))
    finally:
        shutil.copy2(keep, lib)
        keep.unlink()


if __name__ == "__main__":
    main()
