"""A/B of one environment switch of the DMA GEMMs (each value in its own process - the switches are read once):
    python tools/mb_gemm_ab.py RD_GEMM_IL 0 1 [--k16]       (--k16: also force the 8-wavefront kernel at K <= 384 with RD_H3_DMA16=0)"""
import os, subprocess, sys, importlib.util
HERE = os.path.dirname(os.path.abspath(__file__))
SHAPES = ((131072, 768, 384, 0), (65536, 768, 384, 0), (80000, 512, 1024, 0), (80000, 2176, 512, 0), (80000, 1024, 192, 0), (320000, 704, 256, 0),
          (131072, 384, 768, 4), (131072, 384, 384, 0), (32768, 4096, 4096, 0))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    spec = importlib.util.spec_from_file_location("mb", os.path.join(HERE, "microbench.py"))
    mb = importlib.util.module_from_spec(spec); spec.loader.exec_module(mb)
    tag = sys.argv[2]
    for M, K, N, act in SHAPES:
        ms, tf, err = mb.gemm(M, K, N, act=act, iters=20, h3=True, check=(act == 0))
        print(f"{tag} M={M} K={K} N={N} act={act}: {ms*1e3:8.1f} us {tf:6.1f} TF/s err {err}", flush=True)
else:
    var, vals = sys.argv[1], [a for a in sys.argv[2:] if not a.startswith("--")]
    for v in vals:
        env = dict(os.environ, **{var: v})
        if "--k16" in sys.argv:
            env["RD_H3_DMA16"] = "0"
        r = subprocess.run([sys.executable, __file__, "child", f"{var}={v}"], env=env, capture_output=True, text=True)
        print(r.stdout, end="")
        if r.returncode != 0:
            print(r.stderr[-1500:])
