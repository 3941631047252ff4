"""Phase stamps of gemm_h3_dma_kernel (8 wavefronts) for workgroup 0 (RD_GEMM_TRACE=1, printed to stderr by the launcher):
    RD_GEMM_TRACE=1 RD_H3_DMA16=0 python tools/mb_gemm_trace.py [M K N act]"""
import os, sys, importlib.util
os.environ.setdefault("RD_GEMM_TRACE", "1")
os.environ.setdefault("RD_H3_DMA16", "0")          # force the 8-wavefront kernel also for K <= 384
HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("mb", os.path.join(HERE, "microbench.py"))
mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
a = [int(v) for v in sys.argv[1:5]] or [131072, 768, 384, 0]
ms, tf, err = mb.gemm(a[0], a[1], a[2], act=a[3] if len(a) > 3 else 0, iters=1, h3=True)
print("traced launch (synchronous, not a timing):", a, file=sys.stderr)
