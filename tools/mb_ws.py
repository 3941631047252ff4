"""Quick A/B of the ws mixer (C = 192) at the bench's typical launch sizes: python tools/mb_ws.py [variant ...]"""
import sys
sys.argv = [sys.argv[0]] + [a for a in sys.argv[1:]]
import importlib.util, os
spec = importlib.util.spec_from_file_location("mb", os.path.join(os.path.dirname(__file__), "microbench.py"))
mb = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mb)
variants = [int(a) for a in sys.argv[1:] if a.isdigit()] or [200]
for M in (105600, 131072, 52800, 190000, 33000):
    for v in variants:
        ms, tf, err = mb.mixer(192, M, v, check=True)
        print(f"mixer C=192 M={M} variant {v}: {ms*1e3:8.1f} us {tf:7.1f} TF/s  max abs err vs fp64 {err:.2e}", flush=True)
