"""A/B of the DMA GEMMs' output-tile order (RD_GEMM_ORDER, read once per process) at the step's big shapes:
    python tools/mb_gemm_order.py            # runs itself once per order in a subprocess"""
import os, subprocess, sys, importlib.util
HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = ((131072, 384, 768, 4), (131072, 768, 384, 0), (65536, 384, 768, 4), (65536, 768, 384, 0), (80000, 512, 1024, 0), (80000, 2176, 512, 0),
          (131072, 384, 384, 0), (320000, 256, 512, 0), (32768, 4096, 4096, 0))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    spec = importlib.util.spec_from_file_location("mb", os.path.join(HERE, "microbench.py"))
    mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
    for M, K, N, act in SHAPES:
        ms, tf, err = mb.gemm(M, K, N, act=act, iters=20, h3=True, check=(act == 0))
        print(f"order {os.environ.get('RD_GEMM_ORDER', '0')} M={M} K={K} N={N} act={act}: {ms*1e3:8.1f} us {tf:6.1f} TF/s err {err}", flush=True)
else:
    for order in sys.argv[1:] or ["0", "1"]:
        env = dict(os.environ, RD_GEMM_ORDER=order)
        r = subprocess.run([sys.executable, __file__, "child"], env=env, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode != 0:
            print(r.stderr[-1500:])
