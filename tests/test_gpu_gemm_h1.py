"""GPU: the single-accumulator split GEMM (csrc/kernels_gemm_h1.hip) against fp64 through its developer entry `rd_debug_gemm_h1`
(strides, bias, activation, residual, range flag) - edge tiles in M and N, K of 2 .. 68 K tiles, row strides wider than the tensor,
every activation the engine fuses, and the two properties the arithmetic rests on: activations of small magnitude keep an ABSOLUTE
error of 2^-25 per element (their low plane is an fp16 subnormal, which gfx950's matrix cores keep), weights of any magnitude keep
22 bits relative to the matrix' largest (power-of-two pre-scale).  The layers of the three networks reach the same kernel through
`launch_conv_igemm_h3` (every pointwise layer with K % 32 == 0, K >= 64, N >= 96, M >= 2048) and are covered by the network parity
tests; the round-3 tests `test_split_gemm_*` (tests/test_gpu_parity.py, test_gpu_fullsize.py) run on it unchanged."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_gemm_h1.restype = C.c_float
    lib.rd_debug_gemm_h1.argtypes = [C.c_int] * 5 + [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    return lib


def _act(v, act):
    if act == 1:
        return torch.relu(v)
    if act == 2:
        return torch.nn.functional.gelu(v)
    if act == 3:
        return torch.nn.functional.silu(v)
    if act == 4:
        return torch.sigmoid(v)
    if act == 5:
        return torch.clamp(v / 6 + 0.5, 0, 1)
    return v


def _run(M, K, N, act=0, xs=1.0, ws=0.1, xld=None, yld=None, res=False, bias=True, spike=None, seed=0):
    lib = _lib()
    g = torch.Generator(device="cuda").manual_seed(seed + M + K + N)
    xld, yld = xld or K, yld or N
    x = (torch.rand((M, xld), device="cuda", generator=g) - 0.5) * 2 * xs
    if spike is not None:
        x[M // 2, 3] = spike
    w = (torch.rand((N, K), device="cuda", generator=g) - 0.5) * 2 * ws
    w *= torch.logspace(-2, 1, N, device="cuda")[torch.randperm(N, device="cuda", generator=g)][:, None]      # rows three decades apart
    b = (torch.rand(N, device="cuda", generator=g) - 0.5) if bias else None
    r = (torch.rand((M, yld), device="cuda", generator=g) - 0.5) if res else None
    y = torch.full((M + 3, yld), 7.0, device="cuda")          # guard rows and, with yld > N, guard columns
    flag = C.c_int(0)
    ms = lib.rd_debug_gemm_h1(M, K, N, act, 0, x.data_ptr(), xld, w.data_ptr(), b.data_ptr() if bias else None,
                              r.data_ptr() if res else None, yld, y.data_ptr(), yld, C.byref(flag))
    torch.cuda.synchronize()
    assert ms >= 0, "the kernel did not take the shape"
    ref = x[:, :K].double() @ w.double().t()
    if bias:
        ref = ref + b.double()
    ref = _act(ref, act)
    if res:
        ref = ref + r[:, :N].double()
    err = float((y[:M, :N].double() - ref).abs().max() / ref.abs().max())
    assert float((y[M:] - 7.0).abs().max()) == 0.0                                   # nothing past M
    if yld > N:
        assert float((y[:, N:] - 7.0).abs().max()) == 0.0                            # nothing past N
    return err, flag.value


@pytest.mark.parametrize("case", [
    dict(M=2048, K=64, N=96), dict(M=2048 + 77, K=96, N=128), dict(M=4096 + 255, K=768, N=384), dict(M=4096 + 1, K=384, N=768, act=2),
    dict(M=5000, K=192, N=200, act=1, res=True), dict(M=6000, K=128, N=360, xld=160, yld=400, res=True), dict(M=3000, K=2176, N=130, bias=False),
    dict(M=2500, K=96, N=96, act=5), dict(M=2500, K=160, N=104, act=4), dict(M=70000, K=256, N=512, act=3), dict(M=131072, K=768, N=384, res=True),
], ids=lambda c: "M%d_K%d_N%d_act%d" % (c["M"], c["K"], c["N"], c.get("act", 0)))
def test_matches_fp64_on_edge_shapes_strides_activations_residual(case):
    err, flag = _run(**case)
    assert err < 2e-6 and flag == 0, (case, err, flag)


@pytest.mark.parametrize("xs", [30.0, 1.0, 1e-2, 1e-3, 1e-4])
def test_small_activations_keep_an_absolute_error_of_2_to_the_minus_25(xs):
    """|x| < 2^-3: the low plane is an fp16 subnormal (spacing 2^-24).  Kept by the matrix cores, it bounds the error per element at
    2^-25 ABSOLUTE - relative to the tensor's magnitude the bound relaxes as 1 / xs; flushed, the error would be the low plane itself
    (2^-11 relative: 1e-4 and more on this scale)."""
    err, flag = _run(4096, 768, 384, xs=xs)
    assert flag == 0 and err < 2e-6 * max(1.0, 0.06 / xs), (xs, err)


@pytest.mark.parametrize("ws", [1e-4, 0.1, 50.0])
def test_weight_magnitude_does_not_matter(ws):
    err, flag = _run(4096, 768, 384, ws=ws)
    assert flag == 0 and err < 2e-6, (ws, err)


def test_an_activation_beyond_the_fp16_range_raises_the_range_flag():
    _err, flag = _run(4096, 768, 384, spike=1e5)
    assert flag == 1
