"""GPU parity at the sizes BASELINE.json `configs[1]` actually runs ("What's weak" #1): the persistent 256x128
tiles, XCD swizzles, LDS-DMA pipelines and ~1 GB arena plans only show up at these shapes.  HIP (through the C-ABI) vs
the CPU oracle (`oracle/nets.py`, pinned bit-exact to the reference modules by tests/test_oracle_golden.py) on the same
seeded inputs, in the default `auto` precision AND in native `fp32`.

Bar (north_star): logits / maps / features within 1e-3 absolute, CTC argmax identical wherever the oracle's top-2 logit
gap exceeds 1e-2 (a tie inside fp32 noise is not a parity failure)."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import nets as O
from rapiddoc_amd import weights as W

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _engine(golden_dir, kind, precision):
    from rapiddoc_amd.engine import RdEngine
    st = W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{kind}.json"), 0)
    eng = RdEngine(kind, guard="off").load_weights(st)
    if precision != "auto":
        eng.set_precision(precision)
    return eng, O.as_torch_state(st)


@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_det_fullsize_960x704(golden_dir, precision, monkeypatch):
    """rapidocr DetPreProcess output size of a 1684x1191 page (limit_side_len 960): [2,3,960,704] (rapid_ocr.py:517-518)."""
    monkeypatch.setenv("RD_PRECISION", "auto")
    eng, st = _engine(golden_dir, "ppocrv6_det", precision)
    x = torch.from_numpy(np.random.default_rng(11).standard_normal((2, 3, 960, 704)).astype(np.float32))
    ref = O.det_forward(st, x).numpy()
    y = eng.det_forward(x.cuda()).cpu().numpy()
    assert not eng.range_overflow()
    assert y.shape == ref.shape == (2, 1, 960, 704)
    assert np.abs(y - ref).max() < TOL


@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_b4_fullsize_800x800(golden_dir, precision, monkeypatch):
    """PP-DocLayout-L/plus-L/V2/V3 input size (pp_doclayout/main.py:17-29): [2,3,800,800] -> 4 feature maps."""
    monkeypatch.setenv("RD_PRECISION", "auto")
    eng, st = _engine(golden_dir, "pphgnetv2_b4", precision)
    x = torch.from_numpy(np.random.default_rng(12).uniform(0, 1, (2, 3, 800, 800)).astype(np.float32))
    ref = O.pphgnetv2_features(st, x)
    feats = eng.backbone_forward(x.cuda())
    assert not eng.range_overflow()
    assert [tuple(f.shape[1:]) for f in feats] == [(128, 200, 200), (512, 100, 100), (1024, 50, 50), (2048, 25, 25)]
    for r, f in zip(ref, feats):
        assert np.abs(f.cpu().numpy() - r.numpy()).max() < TOL


@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_rec_fullsize_64x1088(golden_dir, precision, monkeypatch):
    """One rec batch of the benchmark: 64 lines padded to width 1088 -> logits [64,136,18710], argmax, max-prob."""
    from rapiddoc_amd.engine import REC_WANT_LOGITS
    monkeypatch.setenv("RD_PRECISION", "auto")
    eng, st = _engine(golden_dir, "ppocrv6_rec", precision)
    x = torch.from_numpy(np.random.default_rng(13).uniform(-1, 1, (64, 3, 48, 1088)).astype(np.float32))
    x[40:, :, :, 700:] = 0.0                         # zero right-padding of shorter lines (rapidocr resize_norm_img)
    lg = O.rec_forward(st, x)
    ridx, rprob = O.ctc_greedy_stats(lg)
    top2 = torch.topk(lg, 2, dim=2).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-2).numpy()
    idx, prob, _ = eng.rec_forward(x.cuda())                   # fused CTC head (the product path)
    assert idx.shape == (64, 136)
    assert (idx.cpu().numpy() == ridx.numpy())[safe].all()
    assert np.abs(prob.cpu().numpy() - rprob.numpy())[safe].max() < TOL
    _, _, full = eng.rec_forward(x[:8].cuda(), REC_WANT_LOGITS)    # raw logits of the first 8 lines (82 MB)
    lg8 = O.rec_forward(st, x[:8])                                  # batch changes nothing: every op is per-sample
    assert np.abs(full.cpu().numpy() - lg8.numpy()).max() < TOL
    assert not eng.range_overflow()


def test_pipeline_rec_equals_oracle_on_the_crops_it_made(golden_dir):
    """End to end on 2 synthetic pages: the tensors `crop_batch_kernel` produced are pulled back and the oracle's
    recogniser is run on those very tensors; (idx, prob) of every rec batch must match (rapid_ocr.py:443-449)."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0)
              for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, rec_batch_num=32)
    pipe.keep_rec_inputs = True
    pages_np, boxes = synth_batch(5, 2)
    pages = torch.from_numpy(pages_np).cuda()
    det_hw = pipe.det_forward(pages[:1])[1]
    maps = render_text_maps(boxes, pages_np.shape[1:3], det_hw, pages.device)
    res = pipe.run_batch(pages, None, det_maps_override=maps)
    assert [len(r.lines) for r in res] == [45, 45]
    st = O.as_torch_state(states["ppocrv6_rec"])
    assert len(pipe.last_rec_batches) >= 3 and sum(len(c) for c, *_ in pipe.last_rec_batches) == 90
    from rapiddoc_amd import ocr_host
    flat = [ln for r in res for ln in r.lines]
    for chunk, x, idx, prob in pipe.last_rec_batches:
        lg = O.rec_forward(st, x.cpu())
        ridx, rprob = O.ctc_greedy_stats(lg)
        top2 = torch.topk(lg, 2, dim=2).values
        safe = ((top2[..., 0] - top2[..., 1]) > 1e-2).numpy()
        assert (idx.cpu().numpy() == ridx.numpy())[safe].all()
        assert np.abs(prob.cpu().numpy() - rprob.numpy())[safe].max() < TOL
        # and the strings the pipeline returned are the decode of exactly these indices
        dec = ocr_host.ctc_decode(idx.cpu().numpy(), prob.cpu().numpy(), pipe.characters)
        for j, i in enumerate(chunk.tolist()):
            assert flat[i][1] == dec[j][0]


# ---------------------------------------------------------------------------------------------------------------------
# split-fp16 arithmetic on adversarial operand ranges: magnitudes from 1e-5 to 1e4 in one product
# ---------------------------------------------------------------------------------------------------------------------
def _wide_range(shape, lo_exp, hi_exp, gen):
    mag = 10.0 ** (torch.rand(shape, device="cuda", generator=gen) * (hi_exp - lo_exp) + lo_exp)
    sign = torch.where(torch.rand(shape, device="cuda", generator=gen) < 0.5, -1.0, 1.0)
    return (mag * sign).float()


@pytest.mark.parametrize("M,K,N", [(4096, 192, 384), (3000, 768, 200)])
def test_split_gemm_wide_operand_range(M, K, N):
    """x in +-[1e-5, 1e4], w in +-[1e-5, 1]: the split product must hold a componentwise bound of the same class as an
    fp32 FMA chain: |y - ref| <= 2e-6 * (|x| @ |w|^T) (each operand keeps 22 of 24 mantissa bits; fp32 accumulate)."""
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_time_gemm.restype = C.c_float
    lib.rd_debug_time_gemm.argtypes = [C.c_int] * 5 + [C.c_void_p] * 6
    g = torch.Generator(device="cuda").manual_seed(M + K)
    x = _wide_range((M, K), -5, 4, g)
    w = _wide_range((N, K), -5, 0, g)
    b = torch.zeros(N, device="cuda")
    Kp = (K + 31) // 32 * 32
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    wh = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wh[:, :K] = hi
    wl = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wl[:, :K] = lo
    y = torch.empty((M, N), device="cuda")
    lib.rd_debug_time_gemm(M, K, N, 0, 1, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), wh.data_ptr(), wl.data_ptr())
    torch.cuda.synchronize()
    ref = x.double() @ w.double().t()
    bound = 2e-6 * (x.double().abs() @ w.double().abs().t()) + 1e-30
    assert bool(((y.double() - ref).abs() <= bound).all()), float(((y.double() - ref).abs() / bound).max())
    # the fp32-MFMA kernel under the same bound (what the reference's fp32 arithmetic class gives)
    y32 = torch.empty((M, N), device="cuda")
    lib.rd_debug_time_gemm(M, K, N, 0, 1, x.data_ptr(), w.data_ptr(), b.data_ptr(), y32.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert bool(((y32.double() - ref).abs() <= bound).all())


def test_split_mixer_wide_operand_range():
    """The fused mixer with activations spanning 1e-5 .. 1e3 and weights 1e-4 .. 0.3 against fp64."""
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_time_mixer.restype = C.c_float
    lib.rd_debug_time_mixer.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6
    C_, M = 192, 5000
    g = torch.Generator(device="cuda").manual_seed(77)
    x = _wide_range((M, C_), -5, 3, g)
    w1 = _wide_range((2 * C_, C_), -4, -0.5, g)
    w2 = _wide_range((C_, 2 * C_), -4, -0.5, g)
    b1 = torch.rand(2 * C_, device="cuda", generator=g) - 0.5
    b2 = torch.rand(C_, device="cuda", generator=g) - 0.5
    y = torch.empty((M, C_), device="cuda")
    lib.rd_debug_time_mixer(C_, M, 100, 1, x.data_ptr(), y.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr())
    torch.cuda.synchronize()
    xd = x.double()
    pre = xd @ w1.double().t() + b1.double()
    h = torch.nn.functional.gelu(pre)
    ref = xd + h @ w2.double().t() + b2.double()
    # first-order error bound: the hidden pre-activation carries 2e-6 * |x| |w1| (GELU' <= 1.13), the second product the same
    # relative class on |h| |w2|, the fast erf adds 5e-7 per hidden unit
    e1 = 2e-6 * (xd.abs() @ w1.double().abs().t()) + 5e-7
    bound = (1.13 * e1) @ w2.double().abs().t() + 2e-6 * (h.abs() @ w2.double().abs().t()) + 2e-7 * ref.abs() + 1e-6
    err = (y.double() - ref).abs()
    assert bool((err <= bound).all()), float((err / bound).max())


def test_ctc_head_small_dictionary(golden_dir):
    """a Latin-size dictionary (C = 40 classes) used to leave empty class splits whose (-inf, 0) partials
    merged to NaN.  Fused head == unfused statistics == oracle."""
    from rapiddoc_amd.engine import REC_UNFUSED_CTC, RdEngine
    man = [(n, (40, 120) if n == "head.head.weight" else (40,) if n == "head.head.bias" else s, d)
           for n, s, d in W.load_manifest(golden_dir / "manifest_ppocrv6_rec.json")]
    st = W.synth_state_dict(man, 0)
    eng = RdEngine("ppocrv6_rec").load_weights(st)
    assert eng.num_classes == 40
    x = torch.from_numpy(np.random.default_rng(5).uniform(-1, 1, (3, 3, 48, 200)).astype(np.float32))
    lg = O.rec_forward(O.as_torch_state(st), x)
    ridx, rprob = O.ctc_greedy_stats(lg)
    top2 = torch.topk(lg, 2, dim=2).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-2).numpy()
    for flags in (0, REC_UNFUSED_CTC):
        idx, prob, _ = eng.rec_forward(x.cuda(), flags)
        assert bool(torch.isfinite(prob).all())
        assert (idx.cpu().numpy() == ridx.numpy())[safe].all()
        assert np.abs(prob.cpu().numpy() - rprob.numpy())[safe].max() < TOL


# ---------------------------------------------------------------------------------------------------------------------
# range guard on every entry point: B4 session, B6 encoder, RegionOcr / rec_forward_lines
# ---------------------------------------------------------------------------------------------------------------------
def _blow_up_stem(st, gain):
    big = dict(st)
    stem = [k for k in big if k.endswith("weight") and big[k].ndim == 4 and big[k].shape[1] == 3][0]
    big[stem] = big[stem] * gain
    return big


def test_range_guard_b4_and_b6_engines(golden_dir, monkeypatch):
    from rapiddoc_amd.engine import RdEngine
    monkeypatch.setenv("RD_PRECISION", "auto")
    # 256x256: the split kernels only take layers with M >= 2048 pixels (stride-4 stage: 64x64 = 4096)
    for kind, fwd, shape in (("pphgnetv2_b4", "backbone_forward", (1, 3, 256, 256)),
                             ("pphgnetv2_b6_formula", "formula_encoder_forward", (1, 3, 256, 256))):
        big = _blow_up_stem(W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{kind}.json"), 0), 3.0e6)
        x = torch.from_numpy(np.random.default_rng(1).uniform(0, 1, shape).astype(np.float32)).cuda()
        ref_eng = RdEngine(kind, guard="off").load_weights(big).set_precision("fp32")
        ref = getattr(ref_eng, fwd)(x)
        eng = RdEngine(kind).load_weights(big)            # default guard="sync": the forward itself falls back
        got = getattr(eng, fwd)(x)
        assert eng.precision == "fp32" and eng.range_fallbacks == 1
        for a, b in zip(got if isinstance(got, list) else [got], ref if isinstance(ref, list) else [ref]):
            assert torch.equal(a, b)


def test_range_guard_on_rec_forward_lines_path(golden_dir, monkeypatch):
    """analyze.RegionOcr / RegionTextModel call pipe.rec_forward_lines directly: the guard must live there."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, boxes_to_quads
    monkeypatch.setenv("RD_PRECISION", "auto")
    st = {"ppocrv6_det": W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppocrv6_det.json"), 0),
          "ppocrv6_rec": _blow_up_stem(W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppocrv6_rec.json"), 0), 3.0e6)}
    pipe = PagePipeline(st, rec_batch_num=8, n_rec_streams=2)
    pages_np, boxes = synth_batch(0, 1)
    pages = torch.from_numpy(pages_np).cuda()
    quads = [boxes_to_quads(np.asarray(boxes[0])[:10])]
    out = pipe.rec_forward_lines(pages, quads)
    assert len(out[0]) == 10
    assert pipe.stats.get("range_fallbacks", 0) >= 1
    assert all(e.precision == "fp32" for e in pipe.rec_engines + [pipe.rec_tail] if e.range_fallbacks)
    assert any(e.range_fallbacks for e in pipe.rec_engines)
    again = pipe.rec_forward_lines(pages, quads)      # now in fp32: stable, no further fallbacks
    assert again == out
