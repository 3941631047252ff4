"""GPU: the one-accumulator direct 3x3 convolution (csrc/kernels_conv3x3_h1.hip, round 6) against torch conv2d in float64 through
`rd_debug_conv` - which prepares the slab-ordered weight image the way the engine does and reports route 3 when this kernel ran.
Ragged sizes (tile edges in both directions: 8 x 32 output tiles), every channel split the pass structure knows (full passes of 32 input
channels, a 16-channel tail pass, one to three 32-wide output blocks incl. partial ones), activations, residual; the two properties the
arithmetic rests on - activations of small magnitude keep an ABSOLUTE error of 2^-25 per element (unscaled low plane, fp16 subnormals kept
by the matrix cores), weights of any magnitude keep 22 bits relative to the matrix' largest (power-of-two pre-scale) - and the
bit-exactness of an image's rows whatever the launch holds (also covered by tests/test_gpu_launch_invariance.py)."""
import pytest
import torch

from test_gpu_parity import _debug_conv

pytestmark = pytest.mark.gpu


def _ref(x, w, b, act, res):
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), padding=1)
    ref = {0: ref, 1: torch.relu(ref), 2: torch.nn.functional.gelu(ref), 3: torch.nn.functional.silu(ref)}[act]
    if res is not None:
        ref = ref + res.permute(0, 3, 1, 2).double()
    return ref.permute(0, 2, 3, 1)


@pytest.mark.parametrize("N,H,W_,Cin,Cout,act,with_res", [
    (2, 50, 67, 48, 48, 1, False),       # B4 stages.0: one full pass + the 16-channel tail, two output blocks (the second half empty)
    (2, 41, 90, 96, 96, 1, True),        # B4 stages.1: three passes, three blocks, residual
    (1, 40, 33, 128, 96, 1, False),      # stages.1 layers.0: Cin 128
    (2, 13, 333, 64, 32, 0, False),      # one block, OH = 13 (a 5-row edge tile), many column tiles
    (1, 96, 96, 96, 24, 1, False),       # the DB head's conv_down: 24 of a block's 32 channels
    (3, 8, 32, 32, 64, 3, True),         # exactly one tile per image, SiLU, residual
    (2, 9, 31, 80, 72, 2, False),        # Cin 80 = two passes + tail, 72 channels = 2.25 blocks, GELU, W < 32
    (5, 3, 5, 32, 8, 0, False),          # images smaller than a tile
    (1, 200, 200, 48, 48, 1, False),     # > 2 tiles per workgroup of the persistent grid on any chip? no - but many rows: 25 x 7 tiles
])
def test_matches_fp64(N, H, W_, Cin, Cout, act, with_res):
    g = torch.Generator(device="cuda").manual_seed(N * 1000 + Cin + Cout)
    x = torch.rand((N, H, W_, Cin), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((Cout, Cin, 3, 3), device="cuda", generator=g) - 0.5) * 0.2
    w *= torch.logspace(-2, 0.5, Cout, device="cuda")[torch.randperm(Cout, device="cuda", generator=g)][:, None, None, None]
    b = torch.rand((Cout,), device="cuda", generator=g) - 0.5
    res = torch.rand((N, H, W_, Cout), device="cuda", generator=g) if with_res else None
    y, used, _ = _debug_conv(x, w, b, 1, (1, 1, 1, 1), act, res)
    assert used == 3, "the one-accumulator 3x3 kernel did not take this geometry"
    ref = _ref(x, w, b, act, res)
    err = float((y.double() - ref).abs().max())
    assert err < 4e-6 * max(1.0, float(ref.abs().max())), err       # fp32 class: the fp32 MFMA kernels measure 1-2e-6 on these shapes


def test_persistent_grid_runs_many_tiles_per_workgroup():
    """4 x 64 x 64 tiles x 40 images = 2560 tiles over a grid of at most 2 x CUs workgroups: every workgroup loops, the weight stream
    wraps around its cycle many times."""
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand((40, 64, 128, 48), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((48, 48, 3, 3), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand((48,), device="cuda", generator=g) - 0.5
    y, used, _ = _debug_conv(x, w, b, 1, (1, 1, 1, 1), 1, None)
    assert used == 3
    ref = _ref(x[::13], w, b, 1, None)
    assert float((y[::13].double() - ref).abs().max()) < 4e-6 * float(ref.abs().max())       # (the maximum over 6 M outputs)
    y2, _, _ = _debug_conv(x, w, b, 1, (1, 1, 1, 1), 1, None)
    assert torch.equal(y, y2)                                  # deterministic: no run-to-run scheduling in the arithmetic


@pytest.mark.parametrize("xs", [30.0, 1.0, 1e-2, 1e-4])
def test_small_activations_keep_an_absolute_error_of_2_to_the_minus_25(xs):
    g = torch.Generator(device="cuda").manual_seed(3)
    x = (torch.rand((2, 20, 40, 48), device="cuda", generator=g) * 2 - 1) * xs
    w = (torch.rand((48, 48, 3, 3), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.zeros((48,), device="cuda")
    y, used, _ = _debug_conv(x, w, b, 1, (1, 1, 1, 1), 0, None)
    assert used == 3
    ref = _ref(x, w, b, 0, None)
    # per output: sum over K = 432 products of |w| <= 0.1: the low plane's absolute error 2^-25 per activation (below |x| = 2^-3), 2^-22 |x|
    # relative above, plus the fp32 accumulation
    bound = 432 * 0.1 * max(2.0 ** -25, 2.0 ** -22 * xs) * 0.5 + 4e-7 * float(ref.abs().max())
    assert float((y.double() - ref).abs().max()) < bound


def test_weights_three_decades_apart_keep_their_precision():
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand((1, 24, 40, 64), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((64, 64, 3, 3), device="cuda", generator=g) - 0.5) * 0.2
    w[:32] *= 1e-3                                             # half of the channels 1000 x smaller than the matrix' largest weight
    b = torch.zeros((64,), device="cuda")
    y, used, _ = _debug_conv(x, w, b, 1, (1, 1, 1, 1), 0, None)
    assert used == 3
    ref = _ref(x, w, b, 0, None)
    small, large = (y[..., :32].double() - ref[..., :32]).abs().max() / ref[..., :32].abs().max(), (y[..., 32:].double() - ref[..., 32:]).abs().max() / ref[..., 32:].abs().max()
    assert float(large) < 2e-6 and float(small) < 2e-5         # (2^-17 of the largest weight still has a normal low plane)


def test_rows_do_not_depend_on_the_number_of_images():
    g = torch.Generator(device="cuda").manual_seed(5)
    x = torch.rand((70, 24, 40, 96), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((96, 96, 3, 3), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand((96,), device="cuda", generator=g) - 0.5
    big, used, _ = _debug_conv(x, w, b, 1, (1, 1, 1, 1), 1, None)
    one, used1, _ = _debug_conv(x[:1].contiguous(), w, b, 1, (1, 1, 1, 1), 1, None)
    assert used == used1 == 3 and torch.equal(one[0], big[0])
