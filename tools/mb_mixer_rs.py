"""Round 5 A/B of the mixers' residual source (one process per setting - the switches are read once):
    RD_WS_RS  = 0 / 1   ws mixer, C = 192 (variant 400: the prefetching form the engine launches)
    RD_RES_RS = 0 / 1   resident-weights mixer, C = 96 (variant 300)
0 = the tile is read a second time for the residual (rounds 2-4), 1 = the residual is re-formed from the split fragments."""
import importlib.util
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    kind, tag = sys.argv[2], sys.argv[3]
    sys.argv = [sys.argv[0]]
    spec = importlib.util.spec_from_file_location("mb", os.path.join(HERE, "microbench.py"))
    mb = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mb)
    shapes = ((192, 105600, 400), (192, 131072, 400), (192, 52800, 400), (192, 190000, 400)) if kind == "ws" else \
             ((96, 104448, 300), (96, 211200, 300), (96, 262144, 300), (96, 26112, 300))
    for C_, M, v in shapes:
        ms, tf, err = mb.mixer(C_, M, v, check=True)
        gb = 8.0 * M * C_ / 1e9
        print(f"{tag:12s} mixer C={C_} M={M} variant {v}: {ms * 1e3:8.1f} us {tf:7.1f} TF/s {gb / ms:5.2f} TB/s algorithmic  max abs err vs fp64 {err:.2e}", flush=True)
else:
    for var, kind in (("RD_WS_RS", "ws"), ("RD_RES_RS", "res")):
        for v in ("0", "1", "0", "1"):
            env = dict(os.environ, **{var: v})
            r = subprocess.run([sys.executable, __file__, "child", kind, f"{var}={v}"], env=env, capture_output=True, text=True)
            print(r.stdout, end="", flush=True)
            if r.returncode != 0:
                print(r.stderr[-1500:])
