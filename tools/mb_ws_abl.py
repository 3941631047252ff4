import sys, importlib.util, os
spec = importlib.util.spec_from_file_location("mb", os.path.join(os.path.dirname(__file__), "microbench.py"))
mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
names = {0: "full", 1: "no DMA", 2: "no MFMA (reads kept)", 4: "no reads, no MFMA", 8: "no GELU", 16: "no barrier", 17: "no DMA, no barrier",
         25: "MFMA + reads + tile IO only (no DMA/GELU/barrier)", 27: "reads + tile IO only", 29: "tile IO only", 13: "barriers + tile IO", 12: "DMA + barriers + tile IO"}
names.update({32 + k: 'PF ' + v for k, v in list(names.items())})
names.update({60: 'PF tile IO + DMA (no reads / MFMA / GELU / barrier)', 96: 'PF fragments read once (no re-reads)', 104: 'PF fragments read once, no GELU', 120: 'PF fragments read once, no GELU, no barrier'})
for M in (105600,):
    for bits in [int(a) for a in sys.argv[1:]] or sorted(names):
        ms, tf = mb.mixer(192, M, 1000 + 256 * bits, iters=30)
        print(f"ws M={M} ABL {bits:2d} {names.get(bits,''):52s}: {ms*1e3:8.1f} us", flush=True)
