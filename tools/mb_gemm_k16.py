"""8-wavefront (RD_H3_DMA16=0) vs 16-wavefront (=1) DMA GEMM at the step's K <= 384 shapes, each in its own process."""
import os, subprocess, sys, importlib.util
HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = ((131072, 384, 768, 4), (65536, 384, 768, 4), (43056, 384, 768, 4), (131072, 384, 384, 0), (64512, 384, 384, 0), (80000, 192, 192, 0),
          (320000, 256, 512, 0), (101376, 192, 192, 0), (50688, 192, 384, 4), (337920, 96, 96, 0), (130000, 192, 384, 4), (26112, 384, 768, 4))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    spec = importlib.util.spec_from_file_location("mb", os.path.join(HERE, "microbench.py"))
    mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
    for M, K, N, act in SHAPES:
        ms, tf, err = mb.gemm(M, K, N, act=act, iters=20, h3=True, check=(act == 0))
        print(f"DMA16={os.environ.get('RD_H3_DMA16')} M={M} K={K} N={N} act={act}: {ms*1e3:8.1f} us {tf:6.1f} TF/s err {err}", flush=True)
else:
    for v in ("1", "0"):
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, RD_H3_DMA16=v), capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode != 0:
            print(r.stderr[-1500:])
