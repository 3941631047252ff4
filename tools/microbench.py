#!/usr/bin/env python3
"""Kernel micro-benchmarks / ablations on the GPU box (developer tool):  python tools/microbench.py"""
import ctypes as C
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rapiddoc_amd import _lib

lib = _lib.load()
lib.rd_debug_time_mixer.restype = C.c_float
lib.rd_debug_time_mixer.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6
lib.rd_debug_time_gemm.restype = C.c_float
lib.rd_debug_time_gemm.argtypes = [C.c_int] * 5 + [C.c_void_p] * 6


def mixer(C_, M, variant, iters=20, check=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((M, C_), device="cuda", generator=g) - 0.5
    y = torch.empty_like(x)
    w1 = (torch.rand((2 * C_, C_), device="cuda", generator=g) - 0.5) * 0.1
    w2 = (torch.rand((C_, 2 * C_), device="cuda", generator=g) - 0.5) * 0.1
    b1 = torch.zeros(2 * C_, device="cuda")
    b2 = torch.zeros(C_, device="cuda")
    b1 += 0.05
    b2 -= 0.02
    ms = lib.rd_debug_time_mixer(C_, M, variant, iters, x.data_ptr(), y.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr())
    if check:     # every 37th row plus the first and last 1024 (tile edges, the tail of the last workgroup)
        sel = torch.unique(torch.cat([torch.arange(0, min(M, 1024)), torch.arange(0, M, 37), torch.arange(max(0, M - 1024), M)])).cuda()
        xs = x[sel].double()
        ref = xs + torch.nn.functional.gelu(xs @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
        return ms, 8.0 * M * C_ * C_ / ms / 1e9, float((y[sel].double() - ref).abs().max())
    return ms, 8.0 * M * C_ * C_ / ms / 1e9


def ctc(M, Ccls=18385, K=120, nsplit=0, iters=20, split=True):
    """Fused CTC head in isolation (weights prepared here the way engine.cpp does: bias in column K, hi/lo fp16 split)."""
    lib.rd_debug_time_ctc.restype = C.c_float
    lib.rd_debug_time_ctc.argtypes = [C.c_int] * 4 + [C.c_void_p] * 7 + [C.c_int, C.c_void_p]
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((M, K), device="cuda", generator=g) - 0.5
    wp = torch.zeros((Ccls, 128), device="cuda")
    wp[:, :K] = (torch.rand((Ccls, K), device="cuda", generator=g) - 0.5) * 0.2
    wp[:, K] = torch.rand((Ccls,), device="cuda", generator=g) - 0.5
    hi = wp.half()
    lo = ((wp - hi.float()) * 2048.0).half()
    part = torch.empty((M * 64 * 4,), device="cuda")
    idx = torch.empty((M,), dtype=torch.int32, device="cuda")
    prob = torch.empty((M,), device="cuda")
    ns = C.c_int(0)
    ms = lib.rd_debug_time_ctc(M, K, Ccls, iters, x.data_ptr(), wp.data_ptr(), hi.data_ptr() if split else None, lo.data_ptr() if split else None,
                               part.data_ptr(), idx.data_ptr(), prob.data_ptr(), nsplit, C.byref(ns))
    lg = (x.double() @ wp[:, :K].double().t() + wp[:, K].double())[:512]
    ok = bool((lg.argmax(1).int() == idx[:512]).all())
    perr = float((torch.softmax(lg, 1).max(1).values - prob[:512].double()).abs().max())
    return ms, 2.0 * M * K * Ccls / ms / 1e9, ns.value, ok, perr


if __name__ == "__main__" and "--ctc" in sys.argv:
    for M in (2176, 4352, 6528, 8704, 4352 * 23):
        for ns in (0, 8):
            ms, tf, n, ok, perr = ctc(M, nsplit=ns)
            print(f"ctc head M={M} nsplit {n:2d}: {ms*1e3:8.1f} us {tf:7.1f} TF/s  argmax ok {ok}  prob err {perr:.1e}", flush=True)
    ms, tf, n, ok, perr = ctc(4352, split=False)
    print(f"ctc head fp32 MFMA M=4352 nsplit {n}: {ms*1e3:8.1f} us {tf:7.1f} TF/s  argmax ok {ok}  prob err {perr:.1e}")
    sys.exit(0)

if __name__ == "__main__" and "--mixer-res" in sys.argv:
    # narrow blocks: resident-weights kernel (300) vs weight-streaming (200) vs the round-1 split kernel (100)
    for C_, M in ((96, 104448), (96, 211200), (96, 26112), (96, 5000)):
        for v in (100, 200, 300):
            ms, tf, err = mixer(C_, M, v, check=True)
            print(f"mixer C={C_} M={M} variant {v}: {ms*1e3:8.1f} us {tf:7.1f} TF/s  max abs err vs fp64 {err:.2e}", flush=True)
    sys.exit(0)

if __name__ == "__main__" and "--mixer-ws" in sys.argv:
    # round 2: weight-streaming mixer (variant 200 + bits: 1 lock step instead of per-wavefront phases, 2 flipped residual
    # policy) vs the round-1 kernel (100); then ablations (+ 256 * bits, C = 192, garbage results)
    for C_, M in ((192, 105600), (192, 131072), (192, 33000), (192, 4000), (96, 211200), (96, 262144)):
        for v in (100, 200, 201, 204, 208):
            ms, tf, err = mixer(C_, M, v, check=True)
            print(f"mixer C={C_} M={M} variant {v}: {ms*1e3:8.1f} us {tf:7.1f} TF/s  max abs err vs fp64 {err:.2e}", flush=True)
    names = {1: "no DMA", 4: "no frag reads, no MFMA", 8: "no GELU", 12: "no GELU / reads / MFMA (DMA + barriers only)",
             13: "barriers only", 16: "no barrier", 29: "empty loop"}
    for bits in (1, 4, 8, 12, 13, 16, 29):
        ms, tf = mixer(192, 105600, 1000 + 256 * bits)
        print(f"ws ablation M=105600 {names[bits]:45s}: {ms*1e3:8.1f} us", flush=True)
    sys.exit(0)

if __name__ == "__main__" and "--mixer-h3" in sys.argv:
    for C_ in (48, 96, 192):
        for M in (32768 + 77, 131072 * 192 // C_):
            for v in (0, 100):
                ms, tf, err = mixer(C_, M, v, check=True)
                print(f"mixer C={C_} M={M} {'h3  ' if v else 'fp32'}: {ms*1e3:8.1f} us {tf:7.1f} TF/s(nominal) max abs err vs fp64 {err:.2e}")
    for v in (100, 101, 102, 104, 108, 110, 111, 115):
        ms, tf = mixer(192, 131072, v)
        print(f"mixer-h3 C=192 ablation bits {v-100:2d}: {ms*1e3:8.1f} us")
    sys.exit(0)


def gemm(M, K, N, act=0, iters=20, h3=False, check=False):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((M, K), device="cuda", generator=g) - 0.5
    w = (torch.rand((N, K), device="cuda", generator=g) - 0.5) * 0.1
    b = torch.zeros(N, device="cuda")
    y = torch.empty((M, N), device="cuda")
    wh = wl = None
    if h3:
        Kp = (K + 31) // 32 * 32
        hi = w.half()
        lo = ((w - hi.float()) * 2048.0).half()
        wh = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wh[:, :K] = hi
        wl = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wl[:, :K] = lo
    ms = lib.rd_debug_time_gemm(M, K, N, act, iters, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(),
                                wh.data_ptr() if h3 else None, wl.data_ptr() if h3 else None)
    err = None
    if check:
        ref = (x[:4096].double() @ w.double().t())
        err = float((y[:4096].double() - ref).abs().max() / ref.abs().max())
    return ms, 2.0 * M * K * N / ms / 1e9, err


if __name__ == "__main__" and "--small" in sys.argv:
    # the short / narrow layers of the recogniser's neck (LightSVTR): launch- and latency-bound
    for (M, K, N) in ((4352, 120, 120), (4352, 120, 360), (4352, 240, 120), (4352, 384, 120), (10240, 120, 240), (10240, 384, 120)):
        for h3 in (False, True):
            ms, tf, err = gemm(M, K, N, 0, iters=50, h3=h3, check=True)
            print(f"small gemm M={M} K={K} N={N} {'h3  ' if h3 else 'fp32'}: {ms*1e3:7.1f} us  {tf:6.1f} TF/s  err {err:.1e}")
    sys.exit(0)

if __name__ == "__main__" and "--rec-gemms" in sys.argv:
    # the pointwise layers of one 64-line recogniser batch (stage 2 / 3 of PPLCNetV4 + the LightSVTR neck's first layer)
    for (M, K, N, act) in ((52224, 96, 192, 4), (52224, 192, 192, 0), (26112, 192, 384, 4), (26112, 384, 384, 0), (26112, 384, 768, 4),
                           (26112, 768, 384, 0), (4352, 384, 120, 0), (104448, 384, 768, 4)):
        ms, tf, err = gemm(M, K, N, act, iters=50, h3=True, check=(act == 0))
        print(f"rec gemm M={M} K={K} N={N} act {act}: {ms*1e3:7.1f} us  {tf:6.1f} TF/s  err {err}", flush=True)
    sys.exit(0)

if __name__ == "__main__" and "--kxk" in sys.argv:
    # the k x k layers of the step (NHWC geometry, Cin, Cout, k, pads): direct split kernel vs the fp32 MFMA implicit GEMM
    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
    from test_gpu_parity import _debug_conv
    cases = (("B4 stages.0 3x3", 32, 200, 200, 48, 48, 3, (1, 1, 1, 1)), ("B4 stages.1 3x3", 32, 100, 100, 96, 96, 3, (1, 1, 1, 1)),
             ("rec stem2a 2x2", 64, 24, 272, 48, 24, 2, (0, 0, 1, 1)), ("rec stem2b 2x2", 64, 24, 272, 24, 48, 2, (0, 0, 1, 1)),
             ("det conv_down 3x3", 32, 240, 176, 96, 24, 3, (1, 1, 1, 1)), ("layout stem2b 2x2", 32, 400, 400, 16, 32, 2, (0, 0, 1, 1)),
             ("rec stem4 1x1", 64, 12, 136, 48, 96, 1, (0, 0, 0, 0)), ("det stem2a 2x2", 32, 480, 352, 24, 12, 2, (0, 0, 1, 1)))
    for name, N, H, W_, Cin, Cout, k, pads in cases:
        x = torch.rand((N, H, W_, Cin), device="cuda") - 0.5
        w = (torch.rand((Cout, Cin, k, k), device="cuda") - 0.5) * 0.1
        b = torch.zeros(Cout, device="cuda")
        for split in (True, False):
            y, used, ms = _debug_conv(x, w, b, 1, pads, 1, None, split=split, iters=20, force_direct="--force-direct" in sys.argv)
            fl = 2.0 * y.numel() * Cin * k * k
            print(f"{name:20s} M={y.numel() // Cout:8d} K={Cin * k * k:4d} N={Cout:3d} {'split' if split else 'fp32 '} direct={used}: {ms * 1e3:8.1f} us {fl / ms / 1e9:7.1f} TF/s", flush=True)
    sys.exit(0)

if __name__ == "__main__" and "--h3only" in sys.argv:
    for (M, K, N) in ((131072, 192, 384), (131072, 384, 768), (131072, 768, 384), (81920, 2176, 512), (32768, 4096, 4096)):
        ms, tf, err = gemm(M, K, N, 0, h3=True, check=True)
        print(f"h3 gemm M={M} K={K} N={N}: {ms*1e3:8.1f} us  {tf:7.1f} TF/s  err {err:.1e}")
    sys.exit(0)

if __name__ == "__main__":
    names = {0: "full", 1: "no GELU", 2: "no weight stream/barriers", 3: "no GELU, no stream", 4: "GEMM2 only", 8: "GEMM1 only"}
    for M in (256 * 128, 256 * 128 * 4):
        for v in (0, 1, 2, 3, 4, 8):
            ms, tf = mixer(192, M, v)
            print(f"mixer C=192 M={M:7d} variant {v} ({names[v]:26s}): {ms*1e3:8.1f} us  {tf:6.1f} TF/s(nominal)")
    for (M, K, N) in ((131072, 192, 384), (131072, 384, 192), (131072, 384, 768), (131072, 768, 384), (81920, 2176, 512),
                      (131072, 1024, 1024), (32768, 4096, 4096), (131072, 96, 192), (1310720, 48, 96)):
        for h3 in (False, True):
            ms, tf, err = gemm(M, K, N, 0, h3=h3, check=True)
            print(f"gemm M={M} K={K} N={N} {'h3  ' if h3 else 'fp32'}: {ms*1e3:8.1f} us  {tf:7.1f} TF/s   max rel err vs fp64 {err:.2e}")
