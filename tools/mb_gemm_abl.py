"""Ablations of gemm_h3_dma16_kernel at the recogniser's shapes (RD_GEMM_DBG is read once per process): python tools/mb_gemm_abl.py"""
import os, sys, importlib.util
spec = importlib.util.spec_from_file_location("mb", os.path.join(os.path.dirname(__file__), "microbench.py"))
mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
for M, K, N, act in ((50688, 384, 768, 2), (50688, 384, 384, 0), (101376, 192, 192, 0), (101376, 96, 192, 2), (50688, 192, 384, 2)):
    ms, tf, _ = mb.gemm(M, K, N, act=act, iters=30, h3=True)
    print(f"dbg {os.environ.get('RD_GEMM_DBG','0'):>2s} M={M} K={K} N={N} act={act}: {ms*1e3:7.1f} us {tf:6.1f} TF/s", flush=True)
