#!/usr/bin/env python3
"""Fused stem tail (kernels_stem34.hip) at the three networks' bench shapes: us per launch, TFLOP/s, GB/s of algorithmic traffic."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tests"))
sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch
from test_gpu_stem34 import _case, _run

for name, (N, H, W_, Cin, N1, N2, yld) in {"rec stem (136 lines 24x320)": (136, 24, 320, 96, 48, 96, None), "rec stem (140 lines 24x624)": (140, 24, 624, 96, 48, 96, None),
                                            "det stem (32 pages 480x352)": (32, 480, 352, 48, 24, 48, None), "B4 stem (32 pages 400x400)": (32, 400, 400, 64, 32, 48, 384)}.items():
    x, w3, b3, w4, b4 = _case(N, H, W_, Cin, N1, N2, seed=1)
    _, ms = _run(x, w3, b3, w4, b4, yld=yld, iters=20)
    OH, OW = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    fl = 2.0 * N * OH * OW * (9 * Cin * N1 + N1 * N2)
    by = 4.0 * (N * H * W_ * Cin + N * OH * OW * N2)
    print(f"{name:30s} {ms*1e3:8.1f} us  {fl/ms/1e9:7.1f} TFLOP/s  {by/ms/1e6:7.0f} GB/s")
