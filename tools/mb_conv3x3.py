"""The 3x3 / stride-1 layers of the step in isolation (B4 stages.0 / stages.1, the DB head's conv_down, B6's stage layers), against fp64:
    python tools/mb_conv3x3.py            (RD_CONV3X3_H1=0: the round-2..5 kernel, kernels_conv_direct_h3.hip)"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from test_gpu_parity import _debug_conv
CASES = [("B4 stages.0 3x3", 32, 200, 200, 48, 48), ("B4 stages.1 3x3", 32, 100, 100, 96, 96), ("B4 stages.1 layers.0", 32, 100, 100, 128, 96),
         ("det conv_down 3x3", 32, 240, 176, 96, 24), ("B6 stages.0 3x3", 32, 96, 96, 96, 96)]
for name, N, H, W_, cin, cout in CASES:
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((N, H, W_, cin), device="cuda", generator=g) - 0.5
    w = (torch.rand((cout, cin, 3, 3), device="cuda", generator=g) - 0.5) * 0.1
    b = torch.rand(cout, device="cuda", generator=g) - 0.5
    y, used, ms = _debug_conv(x, w, b, 1, (1, 1, 1, 1), 1, None, split=True, iters=20)
    ref = torch.relu(torch.nn.functional.conv2d(x[:2].permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    err = float((y[:2].double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * y.numel() * cin * 9
    print(f"{name:22s} M={y.numel() // cout:8d} K={9 * cin:5d} N={cout:3d} route {used}: {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF/s  rel err vs fp64 {err:.2e}", flush=True)
