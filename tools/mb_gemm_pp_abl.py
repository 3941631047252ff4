"""Ablations of the ping-pong DMA GEMM (RD_GEMM_PP=1, RD_GEMM_PP_ABL bits: 1 no DMA in the K loop, 2 no MFMAs, 4 no fragment reads / split, 8 no split only, 16 no LDS reads only), each in its
own process, at K = 4096 (the K loop alone) and K = 768."""
import os, subprocess, sys, importlib.util
HERE = os.path.dirname(os.path.abspath(__file__))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    spec = importlib.util.spec_from_file_location("mb", os.path.join(HERE, "microbench.py"))
    mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
    for M, K, N in ((32768, 4096, 4096), (131072, 768, 384)):
        ms, tf, _ = mb.gemm(M, K, N, act=0, iters=10, h3=True)
        tiles = -(-M // 256) * -(-N // 128)
        per_cu = -(-tiles // 256)
        print(f"PP={os.environ.get('RD_GEMM_PP')} ABL={os.environ.get('RD_GEMM_PP_ABL', '0')} M={M} K={K} N={N}: {ms*1e3:8.1f} us {tf:6.1f} TF/s   = {ms*1e6/(per_cu*(K//32)):7.1f} ns per K tile and CU", flush=True)
else:
    sel = [a for a in sys.argv[1:]] or ["0", "1", "2", "3", "5", "6", "7", "9", "17", "25"]
    for pp, abl in [("0", "0")] + [("1", a) for a in sel]:
        r = subprocess.run([sys.executable, __file__, "child"], env=dict(os.environ, RD_GEMM_PP=pp, RD_GEMM_PP_ABL=abl), capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode != 0:
            print(r.stderr[-800:])
