"""GPU: the fused stem tail (csrc/kernels_stem34.hip, round 6) - 3x3 / stride 2 / pad 1 conv + bias + act, then the 1x1 conv + bias + act
that consumes it (stem3 -> stem4: rec_lcnetv4.py:154-169, rec_pphgnetv2.py:1040-1056) - against torch conv2d in float64 through
`rd_debug_stem34`, which prepares the two weight images the way the engine does.  The engine does not route through this kernel by default
(it measures level with the two-kernel path, profiles/r6_stem34.txt; RD_STEM34=1 opts in - the last test runs the three networks' golden
parity tests that way in a child process); it stays under test so that the next attempt starts from a correct kernel.

Covered: the three networks' own widths (recogniser 96 -> 48 -> 96, detector 48 -> 24 -> 48, layout / formula backbones 64 -> 32 -> 48) at
odd map sizes (tile edges in both directions: 4 x 32 / 2 x 64 output tiles, odd and even input heights / widths - the last input row /
column is or is not read), row strides wider than the channel count on both sides (channel slices of concat buffers), images smaller than
a tile, many tiles per workgroup of the persistent grid, activations; bit-exactness of an image's rows whatever the launch holds; the
range flag."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_stem34.restype = C.c_float
    lib.rd_debug_stem34.argtypes = [C.c_int] * 11 + [C.c_void_p] * 7
    return lib


def _run(x, w3, b3, w4, b4, act3=1, act4=1, xld=None, yld=None, iters=0, flag=None):
    """x [N,H,W,Cin] fp32 cuda; w3 [N1,Cin,3,3]; w4 [N2,N1].  Returns y [N,OH,OW,N2] (a view into a yld-wide buffer) and ms."""
    N, H, W_, Cin = x.shape
    N1, N2 = w3.shape[0], w4.shape[0]
    xld = xld or Cin
    yld = yld or N2
    xb = torch.full((N, H, W_, xld), 7.0, device="cuda")
    xb[..., :Cin] = x
    OH, OW = (H - 1) // 2 + 1, (W_ - 1) // 2 + 1
    yb = torch.full((N, OH, OW, yld), 555.0, device="cuda")
    w3k = w3.permute(0, 2, 3, 1).reshape(N1, 9 * Cin).contiguous()          # k = (kh * 3 + kw) * Cin + ci
    ms = _lib().rd_debug_stem34(N, H, W_, Cin, xld, N1, N2, yld, act3, act4, iters, xb.data_ptr(), w3k.data_ptr(), b3.data_ptr(),
                                w4.contiguous().data_ptr(), b4.data_ptr(), yb.data_ptr(), flag.data_ptr() if flag is not None else None)
    torch.cuda.synchronize()
    assert ms >= 0, "shape not covered by the fused kernel"
    assert float((yb[..., N2:] - 555.0).abs().max()) == 0.0 if yld > N2 else True      # nothing written beside the view
    return yb[..., :N2], ms


def _act(t, a):
    return {0: t, 1: torch.relu(t), 2: torch.nn.functional.gelu(t), 3: torch.nn.functional.silu(t)}[a]


def _ref(x, w3, b3, w4, b4, act3, act4):
    h = _act(torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w3.double(), b3.double(), stride=2, padding=1), act3)
    o = _act(torch.nn.functional.conv2d(h, w4.double()[:, :, None, None], b4.double()), act4)
    return o.permute(0, 2, 3, 1)


def _case(N, H, W_, Cin, N1, N2, seed):
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand((N, H, W_, Cin), device="cuda", generator=g) * 2 - 1
    w3 = (torch.rand((N1, Cin, 3, 3), device="cuda", generator=g) - 0.5) * 0.2
    w3 *= torch.logspace(-2, 0.5, N1, device="cuda")[torch.randperm(N1, device="cuda", generator=g)][:, None, None, None]
    b3 = torch.rand((N1,), device="cuda", generator=g) - 0.3
    w4 = (torch.rand((N2, N1), device="cuda", generator=g) - 0.5) * 0.4
    b4 = torch.rand((N2,), device="cuda", generator=g) - 0.5
    return x, w3, b3, w4, b4


@pytest.mark.parametrize("N,H,W_,Cin,N1,N2,act3,act4,xld,yld", [
    (3, 24, 320, 96, 48, 96, 1, 1, None, None),      # the recogniser's stem tail (24 x 320 -> 12 x 160)
    (2, 24, 323, 96, 48, 96, 1, 1, None, None),      # odd width: the last input column is read by the last output column's centre tap only
    (2, 37, 45, 48, 24, 48, 1, 1, None, None),       # the detector's widths: 24 of a block's 32 channels, odd height
    (2, 50, 66, 64, 32, 48, 1, 1, None, 384),        # the B4 stem: output = channel slot 0 of stage 1's 384-wide concat buffer
    (1, 9, 200, 96, 48, 96, 1, 1, 128, None),        # input row stride wider than its channel count, 5 output rows (a 1-row edge tile)
    (5, 3, 5, 16, 8, 8, 0, 0, None, None),           # images smaller than a tile, one 16-channel pass, no activations
    (2, 16, 64, 32, 64, 64, 2, 3, None, None),       # two full blocks -> two blocks, GELU then SiLU
    (1, 130, 70, 80, 40, 72, 1, 0, None, None),      # 5 passes, 40 = 1.25 blocks, 72 = 2.25 blocks
])
def test_matches_fp64(N, H, W_, Cin, N1, N2, act3, act4, xld, yld):
    x, w3, b3, w4, b4 = _case(N, H, W_, Cin, N1, N2, seed=N * 1000 + Cin + N1 + N2)
    y, _ = _run(x, w3, b3, w4, b4, act3, act4, xld, yld)
    ref = _ref(x, w3, b3, w4, b4, act3, act4)
    assert y.shape == ref.shape
    err = float((y.double() - ref).abs().max())
    assert err < 4e-6 * max(1.0, float(ref.abs().max())), err       # fp32 class (two chained products)


def test_persistent_grid_runs_many_tiles_per_workgroup():
    """40 images x (48 / 4) x (320 / 32) = 4800 tiles over at most 2 x CUs workgroups: every workgroup loops, the weight stream wraps around
    its 54-slab cycle many times."""
    x, w3, b3, w4, b4 = _case(40, 96, 640, 96, 48, 96, seed=77)
    y, _ = _run(x, w3, b3, w4, b4)
    ref = _ref(x, w3, b3, w4, b4, 1, 1)
    assert float((y.double() - ref).abs().max()) < 4e-6 * max(1.0, float(ref.abs().max()))


def test_rows_do_not_depend_on_the_launch():
    """Image 0 inside a launch of 30 images == image 0 alone, bit for bit (one kernel, one accumulation order per output element)."""
    x, w3, b3, w4, b4 = _case(30, 24, 200, 96, 48, 96, seed=5)
    many, _ = _run(x, w3, b3, w4, b4)
    one, _ = _run(x[:1].contiguous(), w3, b3, w4, b4)
    assert torch.equal(many[:1], one)


def test_range_flag():
    """stem3's activations feed the fp16 split: one beyond 65504 raises the flag (the engine then repeats the call in fp32); ordinary
    magnitudes leave it alone."""
    x, w3, b3, w4, b4 = _case(1, 16, 64, 32, 32, 32, seed=9)
    flag = torch.zeros(4, dtype=torch.int32, device="cuda")
    _run(x, w3, b3, w4, b4, flag=flag)
    assert int(flag[0]) == 0
    _run(x * 1e6, w3, b3, w4, b4, flag=flag)
    assert int(flag[0]) != 0


def test_engine_route_under_RD_STEM34(tmp_path):
    """RD_STEM34=1 (read once per process: a child process): the recogniser, the detector and the layout backbone built through
    Builder::stem_tail's fused op reproduce their reference-minted golden vectors like the default route does."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parents[1]
    env = dict(os.environ, RD_STEM34="1")
    r = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "test_gpu_parity.py"), "-x", "-q", "-k",
                        "matches_golden and not formula"], capture_output=True, text=True, env=env, cwd=root, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert " passed" in r.stdout
