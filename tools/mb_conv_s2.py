"""The recogniser's stride-2 3x3 (stem3: 96 -> 48 channels, K = 864; conv_igemm_h3_kernel<256x64,kxk>) in isolation, against fp64:
    python tools/mb_conv_s2.py            (RD_CONV_FAST_EPI=0: the round-2..4 epilogue)"""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from test_gpu_parity import _debug_conv
for N, H, W_ in ((340, 24, 160), (136, 24, 320), (64, 24, 544)):
    g = torch.Generator(device="cuda").manual_seed(0)
    x = torch.rand((N, H, W_, 96), device="cuda", generator=g) - 0.5
    w = (torch.rand((48, 96, 3, 3), device="cuda", generator=g) - 0.5) * 0.1
    b = torch.rand(48, device="cuda", generator=g) - 0.5
    y, used, ms = _debug_conv(x, w, b, 2, (1, 1, 1, 1), 1, None, split=True, iters=20)
    ref = torch.relu(torch.nn.functional.conv2d(x[:8].permute(0, 3, 1, 2).double(), w.double(), b.double(), stride=2, padding=1)).permute(0, 2, 3, 1)
    err = float((y[:8].double() - ref).abs().max())
    fl = 2.0 * y.numel() * 96 * 9
    print(f"stem3 3x3 s2 N={N} {H}x{W_}: M={y.numel() // 48} {ms * 1e3:8.1f} us {fl / ms / 1e9:6.1f} TF/s  max abs err vs fp64 {err:.2e}", flush=True)
