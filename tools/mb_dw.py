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
