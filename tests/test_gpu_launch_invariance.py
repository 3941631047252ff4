"""GPU: the arithmetic of an image does not depend on the launch it rides in.

The reference runs one process and gives one answer (rapid_ocr.py:404-449); a page-sharded run gives every rank a different number of
pages and lines, so a result that changes with the launch size is a parity gap between N = 1 and N = 8.  Every kernel family fixes the
product order and the accumulator scheme of an output element by the LAYER (kernel size, K, N), never by the number of rows M, the batch
or the launch tensor's width.  Checked here bit for bit, in the default (`auto`, split-fp16) mode and in `fp32`:

  * rec: a line inside a launch of many lines (of other widths, in a wider launch tensor) == the same line as a launch of its own -
    backbone tokens, and the fused head's (argmax, probability) per time step;
  * det: page 0 of a batch of six == page 0 alone;
  * layout backbone: image 0 of a batch of six == image 0 alone (its deep stages have 16 pixels per image: M = 16 vs 96);
  * one layer at a time through `rd_debug_conv` (the routes of launch_conv_igemm_h3 that look at M): the rows of image 0 in an
    M = 70 000+ launch and in an M = 500-row launch are the same bits, for pointwise, 3x3, 2x2, strided and narrow-N layers.

The only size-dependent routes left are address-width guards (a layer whose activation tensor exceeds 2^32 elements = 16 GB takes the
64-bit-addressed fp32 kernel): far above any launch the planners make, stated in DESIGN.md."""
import ctypes as C

import numpy as np
import pytest
import torch

from rapiddoc_amd import weights as W

pytestmark = pytest.mark.gpu


def _state(golden_dir, kind):
    return W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{kind}.json"), 0)


def _lines_input(widths, W_launch, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros((len(widths), 3, 48, W_launch))
    for b, w in enumerate(widths):
        x[b, :, :, :w] = torch.rand((3, 48, w), generator=g) * 2 - 1
    return x


@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_rec_line_in_a_launch_of_many_equals_the_line_alone_bit_for_bit(golden_dir, precision):
    from rapiddoc_amd.engine import RdEngine, rec_line_table
    eng = RdEngine("ppocrv6_rec").load_weights(_state(golden_dir, "ppocrv6_rec"))
    eng.set_precision(precision)
    widths = [320, 320, 323, 401, 517, 640, 638, 77, 16, 48] + [320 + 8 * i for i in range(40)]       # 50 lines: M of the late layers >> 2048
    W_launch = 672
    x = _lines_input(widths, W_launch, seed=5)
    w = np.asarray(widths)
    T = (((w - 1) // 2 + 1 - 1) // 2 + 1) // 2
    first = np.cumsum(T) - T
    tab = torch.from_numpy(rec_line_table(w, first)).cuda()
    tokens = torch.zeros((int(first[-1] + T[-1]), eng.rec_token_dim), device="cuda")
    eng.rec_backbone_forward_lines(x.cuda(), tab, tokens)
    torch.cuda.synchronize()
    got = tokens.cpu()
    for b in (0, 2, 3, 5, 7, 8, 9, 30, 49):
        wb = widths[b]
        alone = eng.rec_backbone_forward(x[b:b + 1, :, :, :wb].contiguous().cuda()).cpu()[0]     # M = 1 line: every layer far below 2048 rows
        mine = got[first[b]: first[b] + T[b]]
        assert torch.equal(mine, alone), (precision, b, wb, float((mine - alone).abs().max()))
    assert not eng.range_overflow()


@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_rec_forward_of_one_line_equals_its_rows_in_a_batch(golden_dir, precision):
    """rd_rec_forward (backbone + SVTR neck + fused CTC head): (argmax, max-probability) of line 0 / lines 0-2 alone == inside 24 lines."""
    from rapiddoc_amd.engine import RdEngine
    eng = RdEngine("ppocrv6_rec").load_weights(_state(golden_dir, "ppocrv6_rec"))
    eng.set_precision(precision)
    x = (torch.rand((24, 3, 48, 480), generator=torch.Generator().manual_seed(11)) * 2 - 1).cuda()
    idx, prob, _ = eng.rec_forward(x)
    idx, prob = idx.clone(), prob.clone()
    for n in (1, 3):
        i1, p1, _ = eng.rec_forward(x[:n].contiguous())
        assert torch.equal(i1, idx[:n]) and torch.equal(p1, prob[:n]), (precision, n, float((p1 - prob[:n]).abs().max()))
    assert not eng.range_overflow()


@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_det_page_alone_equals_the_page_in_a_batch(golden_dir, precision):
    from rapiddoc_amd.engine import RdEngine
    eng = RdEngine("ppocrv6_det").load_weights(_state(golden_dir, "ppocrv6_det"))
    eng.set_precision(precision)
    x = torch.randn((6, 3, 160, 224), generator=torch.Generator().manual_seed(3)).cuda()
    full = eng.det_forward(x).clone()
    one = eng.det_forward(x[:1].contiguous())
    assert torch.equal(one[0], full[0]), (precision, float((one[0] - full[0]).abs().max()))
    assert not eng.range_overflow()


@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_layout_backbone_image_alone_equals_the_image_in_a_batch(golden_dir, precision):
    from rapiddoc_amd.engine import RdEngine
    eng = RdEngine("pphgnetv2_b4").load_weights(_state(golden_dir, "pphgnetv2_b4"))
    eng.set_precision(precision)
    x = torch.rand((6, 3, 128, 128), generator=torch.Generator().manual_seed(4)).cuda()
    full = [f.clone() for f in eng.backbone_forward(x)]
    one = eng.backbone_forward(x[:1].contiguous())
    for lvl, (a, b) in enumerate(zip(one, full)):
        assert torch.equal(a[0], b[0]), (precision, lvl, float((a[0] - b[0]).abs().max()))
    assert not eng.range_overflow()


def _split(w):
    """fp32 [rows][K] -> (hi, lo) fp16 images with rows padded to ceil32(K), lo = fp16((w - hi) * 2^11): csrc split_weights_h3."""
    rows, K = w.shape
    Kp = (K + 31) // 32 * 32
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    H = torch.zeros((rows, Kp), dtype=torch.float16, device=w.device)
    L = torch.zeros((rows, Kp), dtype=torch.float16, device=w.device)
    H[:, :K], L[:, :K] = hi, lo
    return H.contiguous(), L.contiguous()


CONV_CASES = [
    # (Cin, Cout, k, stride, pad, image H, W, images in the big launch)   rows per image = OH * OW
    dict(cin=192, cout=384, k=1, s=1, pad=0, H=12, W=40, n_big=150),      # pointwise, LDS-DMA GEMM family            M 480 vs 72 000
    dict(cin=100, cout=128, k=1, s=1, pad=0, H=12, W=40, n_big=150),      # pointwise, K % 32 != 0
    dict(cin=48, cout=48, k=3, s=1, pad=1, H=20, W=25, n_big=150),        # 3x3 stride 1, <= 96 channels: direct kernel   M 500 vs 75 000
    dict(cin=64, cout=64, k=3, s=1, pad=1, H=20, W=25, n_big=150),        # 3x3, N in (32, 64]: the 256x64 / 128x64 pick by M
    dict(cin=96, cout=192, k=3, s=2, pad=1, H=24, W=40, n_big=300),       # 3x3 stride 2 (B4 down-sampling): implicit GEMM 256x128
    dict(cin=32, cout=32, k=3, s=1, pad=1, H=20, W=25, n_big=150),        # narrow: 128x32 tile
    dict(cin=24, cout=48, k=2, s=1, pad=0, H=21, W=26, n_big=150),        # 2x2 stem layer (stream kernel family, small K)
    dict(cin=128, cout=128, k=3, s=1, pad=1, H=20, W=25, n_big=150),      # 3x3 wide: 256x128
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "cin%d_cout%d_k%d_s%d" % (c["cin"], c["cout"], c["k"], c["s"]))
def test_one_layer_rows_do_not_depend_on_the_launch_size(case):
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_conv.restype = C.c_float
    lib.rd_debug_conv.argtypes = [C.c_int] * 14 + [C.c_void_p] * 8
    cin, cout, k, s, pad, H, Wd, n_big = (case[n] for n in ("cin", "cout", "k", "s", "pad", "H", "W", "n_big"))
    g = torch.Generator(device="cuda").manual_seed(cin * 1000 + cout + k)
    x = torch.rand((n_big, H, Wd, cin), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((cout, k * k * cin), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand(cout, device="cuda", generator=g) - 0.5
    wh, wl = _split(w)
    OH, OW = (H + 2 * pad - k) // s + 1, (Wd + 2 * pad - k) // s + 1

    def run(n):
        y = torch.zeros((n, OH, OW, cout), device="cuda")
        used = C.c_int(0)
        ms = lib.rd_debug_conv(n, H, Wd, cin, cout, k, k, s, pad, pad, pad, pad, 1, 0, x.data_ptr(), w.data_ptr(), wh.data_ptr(), wl.data_ptr(),
                               b.data_ptr(), None, y.data_ptr(), C.byref(used))
        torch.cuda.synchronize()
        assert ms >= 0
        return y, used.value
    big, route_big = run(n_big)
    assert n_big * OH * OW >= 65536 + 1000
    ref = torch.nn.functional.conv2d(x[:2].permute(0, 3, 1, 2).double(), w.view(cout, k, k, cin).permute(0, 3, 1, 2).double(), b.double(),
                                     stride=s, padding=pad).relu().permute(0, 2, 3, 1)
    assert float((big[:2].double() - ref).abs().max()) < 2e-5 * float(ref.abs().max())
    for n in (1, 5):                                       # M = rows of one image (~500) and of five (M % tile != 0 in another place)
        small, route_small = run(n)
        assert route_small == route_big, (route_small, route_big)
        assert torch.equal(small, big[:n]), (case, n, float((small - big[:n]).abs().max()))


@pytest.mark.parametrize("K,N,act", [(192, 384, 2), (96, 192, 2), (384, 768, 2), (768, 384, 0), (100, 128, 1), (2176, 512, 1), (64, 96, 0)],
                         ids=lambda v: str(v))
def test_pointwise_gemm_rows_do_not_depend_on_m(K, N, act):
    """launch_conv_igemm_h3 on a pointwise layer with the operands the engine prepares (one-accumulator image included): the first 517
    rows of an M = 70 001 launch == an M = 517 launch == the first rows of an M = 2 049 launch, bit for bit."""
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_time_gemm.restype = C.c_float
    lib.rd_debug_time_gemm.argtypes = [C.c_int] * 5 + [C.c_void_p] * 6
    g = torch.Generator(device="cuda").manual_seed(K + N)
    M = 70001
    x = torch.rand((M, K), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((N, K), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand(N, device="cuda", generator=g) - 0.5
    wh, wl = _split(w)

    def run(m):
        y = torch.zeros((m, N), device="cuda")
        lib.rd_debug_time_gemm(m, K, N, act, 1, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), wh.data_ptr(), wl.data_ptr())
        torch.cuda.synchronize()
        return y
    big = run(M)
    ref = x[:64].double() @ w.double().t() + b.double()
    ref = torch.relu(ref) if act == 1 else torch.nn.functional.gelu(ref) if act == 2 else ref
    assert float((big[:64].double() - ref).abs().max()) < 2e-5 * max(1.0, float(ref.abs().max()))
    for m in (517, 2049, 31, 1):
        small = run(m)
        assert torch.equal(small, big[:m]), (K, N, act, m, float((small - big[:m]).abs().max()))
