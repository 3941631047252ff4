"""Single-accumulator split GEMM (kernels_gemm_h1.hip): fp64 checks and timings at the step's shapes, against the round-3 DMA kernel.

    python tools/mb_gemm_h1.py check                 fp64 error at edge shapes, strides, activations, residual, small magnitudes, range flag
    python tools/mb_gemm_h1.py perf [ENV=V ...]      one process per configuration: default, RD_GEMM_H1=0 (round-3 kernel), and the given ones
"""
import ctypes as C
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

SHAPES = ((131072, 768, 384, 0), (65536, 768, 384, 0), (131072, 384, 768, 2), (65536, 384, 768, 2), (131072, 384, 384, 0), (80000, 512, 1024, 0),
          (80000, 2176, 512, 0), (80000, 192, 192, 0), (320000, 704, 256, 0), (320000, 256, 512, 0), (131072, 192, 384, 2), (262080, 96, 192, 2),
          (21120, 768, 384, 0), (32768, 4096, 4096, 0))


def _lib():
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_time_gemm.restype = C.c_float
    lib.rd_debug_time_gemm.argtypes = [C.c_int] * 5 + [C.c_void_p] * 6
    lib.rd_debug_gemm_h1.restype = C.c_float
    lib.rd_debug_gemm_h1.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    return lib


def act_ref(v, act):
    import torch
    if act == 1:
        return torch.relu(v)
    if act == 2:
        return torch.nn.functional.gelu(v)
    if act == 3:
        return torch.nn.functional.silu(v)
    if act == 5:
        return torch.clamp(v / 6 + 0.5, 0, 1)
    return v


def run_h1(lib, M, K, N, act=0, xs=1.0, ws=0.1, xld=None, yld=None, res=False, bias=True, seed=0, iters=0, spike=None):
    """-> (ms, max abs err / max |ref| vs fp64, range flag)"""
    import torch
    g = torch.Generator(device="cuda").manual_seed(seed)
    xld = xld or K
    yld = yld or N
    xb = (torch.rand((M, xld), device="cuda", generator=g) - 0.5) * 2 * xs
    if spike is not None:
        xb[M // 2, 3] = spike
    w = (torch.rand((N, K), device="cuda", generator=g) - 0.5) * 2 * ws
    # per-channel magnitudes three decades apart: the per-channel pre-scale has something to do
    w *= torch.logspace(-2, 1, N, device="cuda")[torch.randperm(N, device="cuda", generator=g)][:, None]
    b = (torch.rand(N, device="cuda", generator=g) - 0.5) if bias else None
    r = (torch.rand((M, yld), device="cuda", generator=g) - 0.5) if res else None
    y = torch.full((M, yld), 7.0, device="cuda")
    flag = C.c_int(0)
    ms = lib.rd_debug_gemm_h1(M, K, N, act, iters, xb.data_ptr(), xld, w.data_ptr(), b.data_ptr() if bias else None,
                              r.data_ptr() if res else None, yld, y.data_ptr(), yld, C.byref(flag))
    torch.cuda.synchronize()
    rows = torch.cat([torch.arange(0, min(M, 1024)), torch.arange(max(M - 1024, 0), M)]).unique().cuda()
    ref = xb[rows][:, :K].double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    ref = act_ref(ref, act)
    if res:
        ref = ref + r[rows][:, :N].double()
    err = float((y[rows][:, :N].double() - ref).abs().max() / ref.abs().max())
    untouched = bool((y[:, N:] == 7.0).all()) if yld > N else True
    return ms, err, flag.value, untouched


def check():
    import torch
    lib = _lib()
    ok = True
    cases = [
        dict(M=2048, K=64, N=96), dict(M=2048 + 77, K=96, N=128), dict(M=4096 + 255, K=768, N=384), dict(M=4096 + 1, K=384, N=768, act=2),
        dict(M=5000, K=192, N=200, act=1, res=True), dict(M=6000, K=128, N=360, xld=160, yld=400, res=True), dict(M=3000, K=2176, N=130, bias=False), dict(M=2500, K=96, N=96, act=5),
        dict(M=70000, K=256, N=512, act=3), dict(M=131072, K=768, N=384, res=True),
        # small magnitudes: the low planes are fp16 subnormals (absolute error 2^-25 per operand)
        dict(M=4096, K=768, N=384, xs=1e-2), dict(M=4096, K=768, N=384, xs=1e-3), dict(M=4096, K=768, N=384, xs=1e-4), dict(M=4096, K=768, N=384, xs=30.0),
        dict(M=4096, K=768, N=384, ws=1e-4), dict(M=4096, K=768, N=384, ws=50.0),
    ]
    for c in cases:
        ms, err, flag, untouched = run_h1(lib, **c)
        # activations of magnitude xs keep an ABSOLUTE error of 2^-25 per element below 2^-3 (subnormal low plane): the bound relaxes as 1 / xs
        thr = 2e-6 * max(1.0, 0.06 / c.get("xs", 1.0))
        good = ms >= 0 and err < thr and flag == 0 and untouched
        ok &= good
        print(f"{'ok  ' if good else 'FAIL'} {c}: rel err {err:.2e} flag {flag} pad-columns-untouched {untouched}", flush=True)
    ms, err, flag, _ = run_h1(lib, M=4096, K=768, N=384, spike=1e5)
    good = flag == 1
    ok &= good
    print(f"{'ok  ' if good else 'FAIL'} range flag on an activation of 1e5: flag {flag}")
    print("ALL OK" if ok else "SOME FAILED")
    return ok


def perf_child(tag):
    import torch
    lib = _lib()
    for M, K, N, act in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(0)
        x = torch.rand((M, K), device="cuda", generator=g) - 0.5
        w = (torch.rand((N, K), device="cuda", generator=g) - 0.5) * 0.1
        if os.environ.get("MB_ZEROS") == "1":       # zero operands: same instruction stream, far fewer toggling bits - what the power budget costs
            x.zero_()
            w.zero_()
        if os.environ.get("MB_ZEROS") == "2":       # small integers: hi planes busy, lo planes all zero
            x = torch.randint(-8, 9, (M, K), device="cuda", generator=g).float()
            w = torch.randint(-8, 9, (N, K), device="cuda", generator=g).float()
        b = torch.zeros(N, device="cuda")
        y = torch.empty((M, N), device="cuda")
        Kp = (K + 31) // 32 * 32
        hi = w.half()
        lo = ((w - hi.float()) * 2048.0).half()
        wh = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wh[:, :K] = hi
        wl = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wl[:, :K] = lo
        ms = lib.rd_debug_time_gemm(M, K, N, act, 20, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), wh.data_ptr(), wl.data_ptr())
        err = None
        if act == 0:
            ref = x[:2048].double() @ w.double().t()
            err = float((y[:2048].double() - ref).abs().max() / ref.abs().max())
        gb = 4.0 * (M * K + M * N) / 1e9
        print(f"{tag:28s} M={M:6d} K={K:4d} N={N:4d} act={act}: {ms * 1e3:8.1f} us {2.0 * M * K * N / ms / 1e9:6.1f} TF/s {gb / ms:5.2f} TB/s err {err if err is None else format(err, '.1e')}", flush=True)


if __name__ == "__main__":
    if sys.argv[1] == "check":
        sys.exit(0 if check() else 1)
    if sys.argv[1] == "child":
        perf_child(sys.argv[2])
        sys.exit(0)
    configs = [""] + ["RD_GEMM_H1=0"] + sys.argv[2:]
    for cfg in configs:
        env = dict(os.environ)
        for kv in cfg.split(","):
            if kv:
                k, v = kv.split("=")
                env[k] = v
        r = subprocess.run([sys.executable, __file__, "child", cfg or "default"], env=env, capture_output=True, text=True)
        print(r.stdout, end="", flush=True)
        for ln in r.stderr.splitlines():
            if ln.startswith("h1 trace"):          # RD_GEMM1_DBG=32 / 34: the library's cycle accounting
                print(ln, flush=True)
        if r.returncode != 0:
            print(r.stderr[-2000:])
