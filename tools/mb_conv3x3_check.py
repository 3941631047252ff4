"""Developer: repeated full-tensor checks of the one-accumulator 3x3 kernel against fp64 (races show up as run-to-run differences)."""
import sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
from test_gpu_parity import _debug_conv
for (N, H, W_, cin, cout) in ((40, 64, 128, 48, 48), (16, 100, 100, 96, 96), (32, 96, 96, 96, 24), (40, 64, 128, 64, 64)):
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand((N, H, W_, cin), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((cout, cin, 3, 3), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand((cout,), device="cuda", generator=g) - 0.5
    ref = torch.relu(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)).permute(0, 2, 3, 1)
    ys = []
    for rep in range(4):
        y, used, _ = _debug_conv(x, w, b, 1, (1, 1, 1, 1), 1, None, iters=3)
        e = (y.double() - ref).abs()
        bad = e > 2e-6 * float(ref.abs().max())
        ys.append(y)
        print(f"cin{cin} cout{cout} rep {rep}: route {used} max err {float(e.max()):.3e} bad {int(bad.sum())} of {bad.numel()}  nan {int(torch.isnan(y).sum())}"
              + (f"  first bad (n,h,w,c) {torch.nonzero(bad)[0].tolist()}" if bad.any() else "") + f"  equal to rep0 {bool(torch.equal(y, ys[0]))}", flush=True)
