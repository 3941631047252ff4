#!/usr/bin/env python3
"""Depthwise 3x3 micro-benchmark on the recogniser's map shapes (developer tool): RD_DW_LDS=0 python tools/mb_dw.py for the
row-tiled register kernel, default = the LDS-DMA-staged kernel."""
import ctypes as C
import os
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
from rapiddoc_amd import _lib

lib = _lib.load()
lib.rd_debug_dwconv.restype = C.c_float
lib.rd_debug_dwconv.argtypes = [C.c_int] * 8 + [C.c_void_p] * 8

for (N, H, W, Cn, gap_on) in ((128, 6, 132, 192, True), (128, 6, 132, 192, False), (128, 12, 132, 96, False), (128, 3, 132, 384, True),
                              (64, 6, 400, 192, True), (24, 6, 800, 192, True), (512, 6, 40, 192, True)):
    x = torch.rand((N, H, W, Cn), device="cuda") - 0.5
    w = torch.rand((9, Cn), device="cuda") - 0.5
    b = torch.rand((Cn,), device="cuda")
    y = torch.empty_like(x)
    gap = torch.empty((N * 1024 * Cn,), device="cuda")
    ch = C.c_int(0)
    ms = lib.rd_debug_dwconv(N, H, W, Cn, 3, 1, 0, 50, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None,
                             gap.data_ptr() if gap_on else None, C.byref(ch))
    by = 8.0 * N * H * W * Cn
    print(f"RD_DW_LDS={os.environ.get('RD_DW_LDS', '1')} dw3x3 N={N} H={H} W={W} C={Cn} gap={int(gap_on)} chunks={ch.value}: {ms*1e3:7.1f} us  {by/ms/1e9:7.2f} TB/s", flush=True)

# 5x5 / 7x7 (PPHGNetV2 light blocks 32 x 50 x 50 x 192 and 32 x 25 x 25 x 384; the detector's RepLK neck 32 x 240 x 176 x 96 ...)
for (N, H, W, Cn, K) in ((32, 50, 50, 192, 5), (32, 25, 25, 384, 5), (32, 240, 176, 96, 7), (32, 120, 88, 96, 7), (32, 60, 44, 96, 7)):
    x = torch.rand((N, H, W, Cn), device="cuda") - 0.5
    w = torch.rand((K * K, Cn), device="cuda") - 0.5
    b = torch.rand((Cn,), device="cuda")
    y = torch.empty_like(x)
    ch = C.c_int(0)
    ms = lib.rd_debug_dwconv(N, H, W, Cn, K, 1, 0, 30, x.data_ptr(), w.data_ptr(), b.data_ptr(), None, y.data_ptr(), None, None, C.byref(ch))
    by = 8.0 * N * H * W * Cn
    print(f"RD_DW_LDS={os.environ.get('RD_DW_LDS', '1')} dw{K}x{K} N={N} H={H} W={W} C={Cn} staged={int(ch.value == -1)}: {ms*1e3:7.1f} us  {by/ms/1e9:7.2f} TB/s", flush=True)
