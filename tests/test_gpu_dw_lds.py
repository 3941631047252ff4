"""The LDS-DMA-staged depthwise 3x3 (kernels_dw_lds.hip; reference rec_lcnetv4.py:187-206) against torch's fp64 grouped conv,
through the library's own launcher (api.cpp rd_debug_dwconv): every instantiation, strips that end inside a tile, one-column maps,
per-line widths, the residual and activation epilogue and the fused squeeze-excite partial sums."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ACT_NONE, ACT_RELU = 0, 1


def _run(N, H, W, Cn, act=ACT_NONE, res=False, widths=None, bias=True, seed=0, K=3, SH=1, use_gap=True):
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_dwconv.restype = C.c_float
    lib.rd_debug_dwconv.argtypes = [C.c_int] * 8 + [C.c_void_p] * 8
    g = torch.Generator(device="cuda").manual_seed(seed)
    x = torch.rand((N, H, W, Cn), device="cuda", generator=g) - 0.5
    w = torch.rand((K * K, Cn), device="cuda", generator=g) - 0.5
    b = (torch.rand((Cn,), device="cuda", generator=g) - 0.5) if bias else None
    OH = (H + 2 * (K // 2) - K) // SH + 1
    r = (torch.rand((N, OH, W, Cn), device="cuda", generator=g) - 0.5) if res else None
    y = torch.full((N, OH, W, Cn), float("nan"), device="cuda")
    lw = torch.tensor(widths, dtype=torch.int32, device="cuda") if widths is not None else None
    gap = torch.full((N * ((W + 3) // 4 + 8) * H * Cn,), float("nan"), device="cuda")
    chunks = C.c_int(-1)
    lib.rd_debug_dwconv(N, H, W, Cn, K, SH, act, 0, x.data_ptr(), w.data_ptr(), b.data_ptr() if bias else None,
                        r.data_ptr() if res else None, y.data_ptr(), lw.data_ptr() if lw is not None else None, gap.data_ptr() if use_gap else None,
                        C.byref(chunks))
    torch.cuda.synchronize()
    xd = x.double().clone()
    if widths is not None:
        for n, wl in enumerate(widths):
            xd[n, :, wl:, :] = 0
    ref = torch.nn.functional.conv2d(xd.permute(0, 3, 1, 2), w.double().t().reshape(Cn, 1, K, K), b.double() if bias else None,
                                     padding=K // 2, stride=(SH, 1), groups=Cn).permute(0, 2, 3, 1)
    if act == ACT_RELU:
        ref = ref.clamp_min(0)
    if res:
        ref = ref + r.double()
    return y, ref, gap, chunks.value


@pytest.mark.parametrize("shape", [(5, 6, 67, 192), (3, 12, 50, 96), (4, 3, 33, 384), (2, 6, 16, 64), (1, 6, 1, 192), (2, 12, 17, 32),
                                   (1, 3, 2, 64), (9, 6, 401, 192)])
@pytest.mark.parametrize("epi", ["plain", "relu_res"])
def test_dw_lds_matches_fp64(shape, epi):
    N, H, W, Cn = shape
    y, ref, gap, chunks = _run(N, H, W, Cn, act=ACT_RELU if epi == "relu_res" else ACT_NONE, res=epi == "relu_res")
    assert chunks == (W + 15) // 16, "the staged kernel did not take this geometry"
    assert torch.isfinite(y).all()
    assert float((y.double() - ref).abs().max()) < 2e-6
    part = gap[: N * chunks * Cn].view(N, chunks, Cn).double().sum(1)
    assert float((part - ref.sum((1, 2))).abs().max()) < 1e-3 * max(1.0, H * W / 100.0)


def test_dw_lds_line_widths():
    N, H, W, Cn = 6, 6, 70, 192
    widths = [70, 1, 16, 17, 33, 64]
    y, ref, gap, chunks = _run(N, H, W, Cn, widths=widths, seed=3)
    assert chunks == 5
    assert float((y.double() - ref).abs().max()) < 2e-6
    part = gap[: N * chunks * Cn].view(N, chunks, Cn).double().sum(1)
    want = torch.stack([ref[n, :, :wl, :].sum((0, 1)) for n, wl in enumerate(widths)])
    assert float((part - want).abs().max()) < 1e-3


def test_other_geometries_keep_the_register_kernel():
    # 5 x 5 and maps that are not 3 / 6 / 12 rows high stay on the row-tiled kernel: its chunk count is not ceil(W / 16) here
    y, ref, gap, chunks = _run(2, 6, 40, 192, K=5)
    assert float((y.double() - ref).abs().max()) < 2e-6
    y, ref, gap, chunks = _run(2, 8, 40, 192)
    assert float((y.double() - ref).abs().max()) < 2e-6


@pytest.mark.parametrize("shape", [(5, 12, 67, 96), (3, 6, 50, 192), (2, 12, 16, 32), (1, 6, 1, 64), (7, 12, 130, 96)])
@pytest.mark.parametrize("epi", ["plain", "relu_res"])
def test_dw_lds_stride_2_rows(shape, epi):
    """The stride-(2, 1) depthwise conv in front of a stage's first block (rec_lcnetv4.py: `token_conv` of the down-sampling blocks):
    12 -> 6 and 6 -> 3 rows from the same full-height strips."""
    N, H, W, Cn = shape
    widths = [W, 1, 17, W - 1, 16][:N] + [W] * max(0, N - 5) if N > 1 else None
    y, ref, gap, chunks = _run(N, H, W, Cn, act=ACT_RELU if epi == "relu_res" else ACT_NONE, res=epi == "relu_res", SH=2, widths=widths, seed=5)
    assert y.shape[1] == H // 2 and chunks == (W + 15) // 16, "the staged kernel did not take this geometry"
    assert torch.isfinite(y).all()
    assert float((y.double() - ref).abs().max()) < 2e-6
    part = gap[: N * chunks * Cn].view(N, chunks, Cn).double().sum(1)
    wl = widths if widths is not None else [W] * N
    want = torch.stack([ref[n, :, :wl[n], :].sum((0, 1)) for n in range(N)])
    assert float((part - want).abs().max()) < 1e-3 * max(1.0, H * W / 100.0)


@pytest.mark.parametrize("K", [5, 7])
@pytest.mark.parametrize("shape", [(2, 50, 50, 192), (3, 25, 25, 384), (1, 40, 44, 96), (2, 8, 16, 32), (1, 1, 1, 32), (1, 9, 17, 64), (2, 33, 5, 96)])
@pytest.mark.parametrize("epi", ["plain", "relu_res"])
def test_dw_kxk_lds_matches_fp64(K, shape, epi):
    """5x5 / 7x7 stride-1 depthwise convs (PPHGNetV2's light blocks, the detector's RepLK neck) on the one-channel-per-lane staged kernel:
    maps smaller than a tile, tiles cut by the right / bottom border, 32 / 64 / 96 / 192 / 384 channels."""
    N, H, W, Cn = shape
    y, ref, _gap, chunks = _run(N, H, W, Cn, act=ACT_RELU if epi == "relu_res" else ACT_NONE, res=epi == "relu_res", K=K, use_gap=False, seed=7)
    assert chunks == -1, "the staged k x k kernel did not take this geometry"
    assert torch.isfinite(y).all()
    assert float((y.double() - ref).abs().max()) < 5e-6


def test_dw_kxk_with_se_sums_stays_on_the_register_kernel():
    y, ref, gap, chunks = _run(2, 20, 24, 64, K=5, use_gap=True)
    assert chunks > 0 and float((y.double() - ref).abs().max()) < 5e-6
    part = gap[: 2 * chunks * 64].view(2, chunks, 64).double().sum(1)
    assert float((part - ref.sum((1, 2))).abs().max()) < 1e-3
