"""GPU parity: HIP engine (through the C-ABI) vs the oracle and vs the reference-minted golden vectors.
Tolerance: BASELINE.json north_star -> bbox/logits within 1e-3 (fp32), decoded indices identical."""
import numpy as np
import pytest
import torch

from oracle import nets as O
from rapiddoc_amd import weights as W

pytestmark = pytest.mark.gpu
TOL = 1e-3


@pytest.fixture(scope="module")
def engines(golden_dir):
    from rapiddoc_amd.engine import RdEngine
    out = {}
    for kind in ("ppocrv6_det", "ppocrv6_rec", "pphgnetv2_b4"):
        st = W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{kind}.json"), 0)
        out[kind] = (RdEngine(kind).load_weights(st), O.as_torch_state(st))
    return out


@pytest.mark.parametrize("tag", ["64x96", "b2_96x160"])
def test_det_matches_golden(engines, golden_dir, tag):
    eng, _ = engines["ppocrv6_det"]
    g = np.load(golden_dir / f"det_seed0_{tag}.npz")
    y = eng.det_forward(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert y.shape == g["maps"].shape
    assert np.abs(y - g["maps"]).max() < TOL


@pytest.mark.parametrize("tag", ["b2_w320", "b1_w96", "b3_w640"])
@pytest.mark.parametrize("flags", [0, 1])
def test_rec_matches_golden(engines, golden_dir, tag, flags):
    eng, _ = engines["ppocrv6_rec"]
    g = np.load(golden_dir / f"rec_seed0_{tag}.npz")
    idx, prob, _ = eng.rec_forward(torch.from_numpy(g["x"]).cuda(), flags)
    idx, prob = idx.cpu().numpy(), prob.cpu().numpy()
    safe = g["top2gap"] > 1e-2
    assert (idx == g["idx"])[safe].all()
    assert np.abs(prob - g["prob"])[safe].max() < TOL


def test_rec_logits_match_golden(engines, golden_dir):
    from rapiddoc_amd.engine import REC_WANT_LOGITS, REC_WANT_SOFTMAX
    eng, st = engines["ppocrv6_rec"]
    g = np.load(golden_dir / "rec_seed0_b2_w320.npz")
    x = torch.from_numpy(g["x"])
    _, _, lg = eng.rec_forward(x.cuda(), REC_WANT_LOGITS)
    lg = lg.cpu().numpy()
    assert np.abs(lg[:, :, ::61] - g["logits_sub"]).max() < TOL
    assert np.abs(lg[:, 0, :] - g["logits_t0"]).max() < TOL
    _, _, sm = eng.rec_forward(x.cuda(), REC_WANT_SOFTMAX)
    ref = torch.softmax(O.rec_forward(st, x), dim=2).numpy()  # what the reference session returns
    assert np.abs(sm.cpu().numpy() - ref).max() < TOL


def test_b4_matches_golden(engines, golden_dir):
    eng, _ = engines["pphgnetv2_b4"]
    g = np.load(golden_dir / "b4_seed0_64x96.npz")
    feats = eng.backbone_forward(torch.from_numpy(g["x"]).cuda())
    for i, f in enumerate(feats):
        assert np.abs(f.cpu().numpy() - g[f"feat{i}"]).max() < TOL


@pytest.mark.parametrize("shape", [(1, 3, 32, 32), (3, 3, 160, 224), (1, 3, 352, 96)])
def test_det_matches_oracle_on_odd_shapes(engines, shape):
    eng, st = engines["ppocrv6_det"]
    x = torch.from_numpy(np.random.default_rng(7).standard_normal(shape).astype(np.float32))
    ref = O.det_forward(st, x).numpy()
    y = eng.det_forward(x.cuda()).cpu().numpy()
    assert np.abs(y - ref).max() < TOL


@pytest.mark.parametrize("shape", [(1, 3, 48, 16), (5, 3, 48, 328), (2, 3, 48, 1200)])
def test_rec_matches_oracle_on_ragged_widths(engines, shape):
    eng, st = engines["ppocrv6_rec"]
    x = torch.from_numpy(np.random.default_rng(8).uniform(-1, 1, shape).astype(np.float32))
    lg = O.rec_forward(st, x)
    ridx, rp = O.ctc_greedy_stats(lg)
    top2 = torch.topk(lg, 2, dim=2).values
    safe = ((top2[..., 0] - top2[..., 1]) > 1e-2).numpy()
    idx, prob, _ = eng.rec_forward(x.cuda())
    assert (idx.cpu().numpy() == ridx.numpy())[safe].all()
    assert np.abs(prob.cpu().numpy() - rp.numpy())[safe].max() < TOL


def test_b4_matches_oracle_batch(engines):
    eng, st = engines["pphgnetv2_b4"]
    x = torch.from_numpy(np.random.default_rng(9).uniform(0, 1, (2, 3, 128, 96)).astype(np.float32))
    ref = O.pphgnetv2_features(st, x)
    feats = eng.backbone_forward(x.cuda())
    for r, f in zip(ref, feats):
        assert np.abs(f.cpu().numpy() - r.numpy()).max() < TOL


def test_pipeline_end_to_end_boxes_from_db_postprocess(engines, golden_dir):
    """Whole device-resident path on 2 synthetic pages: det forward, DB post-process (on maps rendered from the
    generator's boxes), crops, rec with fused CTC, decode.  (That the rec results equal the oracle run on the very same
    crops is asserted by tests/test_gpu_fullsize.py::test_pipeline_rec_equals_oracle_on_the_crops_it_made.)"""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0)
              for k in ("ppocrv6_det", "ppocrv6_rec", "pphgnetv2_b4")}
    pipe = PagePipeline(states, rec_batch_num=16, keep_feats=True)
    pages_np, boxes = synth_batch(5, 2)
    pages = torch.from_numpy(pages_np).cuda()
    det_hw = pipe.det_forward(pages[:1])[1]
    maps = render_text_maps(boxes, pages_np.shape[1:3], det_hw, pages.device)
    res = pipe.run_batch(pages, None, det_maps_override=maps)
    assert [len(r.lines) for r in res] == [45, 45]
    for r, gb in zip(res, boxes):
        got = np.array([[q[:, 0].min(), q[:, 1].min(), q[:, 0].max(), q[:, 1].max()] for q, _, _ in r.lines])
        assert np.abs(got - gb).max() <= 3          # DB boxes ~ generator boxes
        assert all(isinstance(t, str) and 0.0 <= s <= 1.0 for _, t, s in r.lines)
        assert r.layout_feats[3].shape == (2048, 25, 25)
    # det map of the real network was produced too
    assert pipe.last_det[0].shape == (2, 1, det_hw[0], det_hw[1])


def test_crop_kernel_axis_aligned_matches_torch_interpolate(engines):
    """rd_crop_resize_norm_batch on an axis-aligned quad == bilinear resize of the crop (align_corners=False)."""
    import ctypes as C
    from rapiddoc_amd import _lib
    from rapiddoc_amd.pipeline import CROP_DTYPE, quad_to_crop_matrix
    lib = _lib.load()
    rng = np.random.default_rng(3)
    page = torch.from_numpy(rng.integers(0, 256, (1, 200, 300, 3), dtype=np.uint8)).cuda()
    quad = np.array([[40, 50], [240, 50], [240, 90], [40, 90]], np.float32)
    m, cw, ch = quad_to_crop_matrix(quad)
    d = np.zeros(1, dtype=CROP_DTYPE)
    d["page"], d["out_w"], d["crop_w"], d["crop_h"], d["m"] = 0, 240, cw, ch, m
    dd = torch.from_numpy(d.view(np.uint8)).cuda()
    out = torch.empty((1, 3, 48, 256), device="cuda")
    mean = (C.c_float * 3)(0.5, 0.5, 0.5)
    std = (C.c_float * 3)(0.5, 0.5, 0.5)
    assert lib.rd_crop_resize_norm_batch(0, page.data_ptr(), 1, 200, 300, dd.data_ptr(), 1, 48, 256, mean, std, 1 / 255.0, 0,
                                         out.data_ptr(), torch.cuda.current_stream().cuda_stream) == 0
    crop = page[0, 50:90, 40:240].permute(2, 0, 1)[None].float()   # sample points x in [40, 240), y in [50, 90)
    ref = torch.nn.functional.interpolate(crop, size=(48, 240), mode="bilinear", align_corners=False)
    ref = (ref / 255.0 - 0.5) / 0.5
    got = out[:, :, :, :240]
    # interior only: at the crop border the kernel clamps to the crop rectangle exactly like the resize does
    assert float((got - ref).abs().max()) < 2e-3
    assert float(out[:, :, :, 240:].abs().max()) == 0.0


def test_h3_precision_mode_matches_golden(golden_dir, monkeypatch):
    """Experimental RD_PRECISION=h3 (dense layers on the fp16 matrix cores with hi/lo operand splitting, 3 MFMAs per
    product) must meet the same 1e-3 bar as the fp32-MFMA default."""
    from rapiddoc_amd.engine import REC_WANT_LOGITS, RdEngine
    monkeypatch.setenv("RD_PRECISION", "h3")
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppocrv6_rec.json"), 0)
    eng = RdEngine("ppocrv6_rec").load_weights(st)
    g = np.load(golden_dir / "rec_seed0_b2_w320.npz")
    _, _, lg = eng.rec_forward(torch.from_numpy(g["x"]).cuda(), REC_WANT_LOGITS)
    assert np.abs(lg.cpu().numpy()[:, 0, :] - g["logits_t0"]).max() < TOL
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_pphgnetv2_b4.json"), 0)
    eng = RdEngine("pphgnetv2_b4").load_weights(st)
    g = np.load(golden_dir / "b4_seed0_64x96.npz")
    for i, f in enumerate(eng.backbone_forward(torch.from_numpy(g["x"]).cuda())):
        assert np.abs(f.cpu().numpy() - g[f"feat{i}"]).max() < TOL


def test_native_fp32_precision_mode_matches_golden(golden_dir, monkeypatch):
    """precision 'fp32' (native fp32 MFMA only, no split-fp16 mixer) meets the same bar, and the default 'auto' mode
    (split-fp16 channel mixers) agrees with it to fp32 noise."""
    from rapiddoc_amd.engine import REC_WANT_LOGITS, RdEngine
    monkeypatch.setenv("RD_PRECISION", "auto")      # the suite may be run under RD_PRECISION=fp32 / h3
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppocrv6_rec.json"), 0)
    eng = RdEngine("ppocrv6_rec").load_weights(st)
    g = np.load(golden_dir / "rec_seed0_b3_w640.npz")
    x = torch.from_numpy(g["x"]).cuda()
    auto = eng.rec_forward(x, REC_WANT_LOGITS)[2].cpu().numpy()
    kinds_auto = {r["kind"] for r in _profile(eng, lambda: eng.rec_forward(x))}
    eng.set_precision("fp32")
    nat = eng.rec_forward(x, REC_WANT_LOGITS)[2].cpu().numpy()
    kinds_nat = {r["kind"] for r in _profile(eng, lambda: eng.rec_forward(x))}
    split_kinds = {"mixer_fused_h3", "mixer_fused_ws", "mixer_fused_res"}     # round-1 / weight-streaming / resident-weights kernels
    assert kinds_auto & split_kinds and not (kinds_nat & split_kinds) and "mixer_fused" in kinds_nat
    assert not eng.range_overflow()
    assert np.abs(nat[:, 0, :] - g["logits_t0"]).max() < TOL
    assert np.abs(auto - nat).max() < 2e-4
    idx, prob, _ = eng.rec_forward(x)
    safe = g["top2gap"] > 1e-2
    assert (idx.cpu().numpy() == g["idx"])[safe].all()


def _profile(eng, fn):
    eng.set_profiling(True)
    eng.profile_log.clear()
    fn()
    eng.set_profiling(False)
    return list(eng.profile_log)


def test_range_guard_falls_back_to_fp32(golden_dir, monkeypatch):
    """Split-fp16 operands must stay below 65504.  Activations beyond that raise the range flag (never a silently wrong
    answer) and the session repeats the call in native fp32; the result equals a pure-fp32 engine's bit for bit."""
    from rapiddoc_amd.engine import RdEngine
    from rapiddoc_amd.session import Mi355DetSession
    monkeypatch.setenv("RD_PRECISION", "auto")
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppocrv6_det.json"), 0)
    big = dict(st)
    stem = [k for k in big if k.endswith("weight") and big[k].ndim == 4 and big[k].shape[1] == 3][0]
    big[stem] = big[stem] * 3.0e5          # blow the stem up: every later activation is ~1e5 x larger
    x = np.load(golden_dir / "det_seed0_64x96.npz")["x"]
    ref_eng = RdEngine("ppocrv6_det", guard="off").load_weights(big).set_precision("fp32")
    ref = ref_eng.det_forward(torch.from_numpy(x).cuda()).cpu().numpy()
    eng = RdEngine("ppocrv6_det", guard="off").load_weights(big)   # guard off: look at the raw flag
    eng.det_forward(torch.from_numpy(x).cuda())
    assert eng.range_overflow() and not eng.range_overflow()      # raised once, cleared by the read
    sess = Mi355DetSession(big)
    got = sess(x)
    assert sess.engine.precision == "fp32"
    assert np.array_equal(got, ref)


@pytest.mark.parametrize("tag", ["b2_c1_64x96", "b1_c3_96x64"])
def test_formula_encoder_matches_golden(golden_dir, tag):
    """PP-FormulaNet_plus-M encoder (PPHGNetV2_B6_Formula) vs vectors minted from the reference backbone."""
    from rapiddoc_amd.engine import RdEngine
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_pphgnetv2_b6_formula.json"), 0)
    eng = RdEngine("pphgnetv2_b6_formula").load_weights(st)
    g = np.load(golden_dir / f"b6_seed0_{tag}.npz")
    enc = eng.formula_encoder_forward(torch.from_numpy(g["x"]).cuda()).cpu().numpy()
    assert enc.shape == g["enc"].shape
    assert np.abs(enc - g["enc"]).max() < TOL


def _safe_prefix_equal(ids, ref, gaps, min_gap=1e-2):
    """Token ids must match up to (not including) the first step whose top-2 logit gap is inside fp32 noise."""
    for b in range(ref.shape[0]):
        unsafe = np.nonzero(gaps[b] < min_gap)[0]
        n = (unsafe[0] if len(unsafe) else gaps.shape[1]) + 1   # +1: the start token column
        assert (ids[b, :n] == ref[b, :n]).all(), (b, ids[b], ref[b])
    return True


@pytest.mark.parametrize("tag,max_new", [("dec_a", 16), ("dec_b", 24)])
def test_formula_decoder_matches_reference_golden(golden_dir, tag, max_new):
    """GPU MBart decoder (KV cache, hoisted cross-attention K/V) vs the reference head's generate_export token ids."""
    from rapiddoc_amd.engine import RdEngine
    g = np.load(golden_dir / f"formula_seed0_{tag}.npz")
    st = W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_ppformulanet_head_{tag}.json"), 0)
    st["head.decoder.lm_head.weight"][2] *= float(g["eos_gain"])
    eng = RdEngine("ppformulanet_head").load_weights(st)
    assert eng.formula_max_new_tokens == max_new
    ids = eng.formula_decode(torch.from_numpy(g["enc"]).cuda(), max_new).cpu().numpy()
    assert ids.shape == g["ids"].shape
    _safe_prefix_equal(ids, g["ids"], g["top2gap"])
    if (g["top2gap"] >= 1e-2).all():
        assert (ids == g["ids"]).all()


def test_formula_full_model_image_to_ids(golden_dir):
    """encoder handle + decoder handle composed = the reference BaseModel(image) -> ids."""
    from rapiddoc_amd.engine import RdEngine
    g = np.load(golden_dir / "formula_seed0_m8.npz")
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppformulanet_plus_m_m8.json"), 0)
    enc_eng = RdEngine("pphgnetv2_b6_formula").load_weights({k: v for k, v in st.items() if k.startswith("backbone.")})
    dec_eng = RdEngine("ppformulanet_head").load_weights({k: v for k, v in st.items() if k.startswith("head.")})
    enc = enc_eng.formula_encoder_forward(torch.from_numpy(g["x"]).cuda())
    assert np.abs(enc.cpu().numpy() - g["enc"]).max() < TOL
    ids = dec_eng.formula_decode(enc, 8).cpu().numpy()
    assert ids.shape == g["ids"].shape
    _safe_prefix_equal(ids, g["ids"], g["top2gap"])


def test_formula_recognizer_batch_predict(golden_dir):
    """FormulaRecognizer.batch_predict (CustomBaseModel-shaped driver): crops -> ids, equal to the oracle on the same tensors."""
    from oracle import formula as OF
    from rapiddoc_amd import formula_host as FH
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppformulanet_plus_m_m8.json"), 0)
    rec = FH.FormulaRecognizer(st, max_new_tokens=6)
    rng = np.random.default_rng(0)
    imgs = [np.full((60, 200, 3), 255, np.uint8), np.full((90, 90, 3), 255, np.uint8), np.full((50, 50, 3), 128, np.uint8)]
    imgs[0][20:40, 30:170] = rng.integers(0, 120, (20, 140, 3))
    imgs[1][10:80, 10:80] = rng.integers(0, 120, (70, 70, 3))
    out = rec.batch_predict(imgs, batch_size=2)
    assert len(out) == 3 and all(isinstance(o, list) for o in out)
    xs = FH.preprocess(imgs)
    tst = O.as_torch_state(st)
    for i, x in enumerate(xs):
        enc = O.formula_encoder_forward(tst, torch.from_numpy(x))
        ids, lgs = OF.formula_decode(tst, enc, 6, return_logits=True)
        gaps = torch.stack([torch.topk(l, 2, dim=-1).values for l in lgs], 1)
        gaps = (gaps[..., 0] - gaps[..., 1]).numpy()
        ref = ids[0, 1:].tolist()
        n = int(np.nonzero(gaps[0] < 1e-2)[0][0]) if (gaps[0] < 1e-2).any() else len(ref)
        assert out[i][:n] == ref[:n]


@pytest.mark.parametrize("M,K,N", [(2048, 64, 128), (2049, 96, 130), (5000, 192, 384), (4097, 160, 97), (2560, 736, 200),
                                   (3000, 120, 360), (2300, 384, 120), (70000, 48, 96)])
def test_split_gemm_tails_match_fp64(M, K, N):
    """The split-fp16 GEMM kernels (LDS-DMA pipeline for K % 32 == 0 and N > 96, register-staged tiles otherwise) on
    ragged shapes: partial M / N tiles, K of 2..23 K tiles and K not a multiple of 32, against an fp64 product."""
    import ctypes as C
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_time_gemm.restype = C.c_float
    lib.rd_debug_time_gemm.argtypes = [C.c_int] * 5 + [C.c_void_p] * 6
    g = torch.Generator(device="cuda").manual_seed(M + K + N)
    x = torch.rand((M, K), device="cuda", generator=g) - 0.5
    w = (torch.rand((N, K), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand(N, device="cuda", generator=g) - 0.5
    Kp = (K + 31) // 32 * 32
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    wh = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wh[:, :K] = hi
    wl = torch.zeros((N, Kp), dtype=torch.float16, device="cuda"); wl[:, :K] = lo
    y = torch.full((M + 8, N), 777.0, device="cuda")          # guard rows: nothing may be written past M
    lib.rd_debug_time_gemm(M, K, N, 1, 1, x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), wh.data_ptr(), wl.data_ptr())
    torch.cuda.synchronize()
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert float((y[:M].double() - ref).abs().max()) < 2e-6
    assert float((y[M:] - 777.0).abs().max()) == 0.0


_MIXER_SHAPES = [(48, 1000), (96, 4097), (192, 333), (192, 12800 + 5), (192, 70001), (96, 40000 + 3)]


def _mixer_variant_covers(variant, C_):
    """Which (variant, C) pairs rd_debug_time_mixer instantiates: 0 = fp32 debug entry (C = 192), 100 = round-1 split-fp16 (any C),
    200.. = weight-streaming (C = 96 / 192), 300 = resident weights (C = 96), 400.. = prefetching form (C = 192)."""
    if variant == 0 or variant >= 400:
        return C_ == 192
    if variant >= 300:
        return C_ == 96
    if variant >= 200:
        return C_ in (96, 192)
    return True


@pytest.mark.parametrize("C_,M,variant", [(c, m, v) for v in (0, 100, 200, 201, 202, 204, 208, 300, 400, 401, 404, 408)
                                          for c, m in _MIXER_SHAPES if _mixer_variant_covers(v, c)])
def test_fused_mixer_kernels_match_fp64(C_, M, variant):
    """Fused channel mixer x + W2 gelu(W1 x + b1) + b2 (rec_lcnetv4.py:226-236): fp32-MFMA kernel (variant 0, C = 192
    only in the debug entry), round-1 split-fp16 kernel (variant 100) and the weight-streaming kernel (200; +1 lock step, +4 per-wavefront
    phases, +8 per-workgroup phases, +2 flipped residual policy), its prefetching form (400 + the same phase bits: next tile and residual
    re-fetched into the X registers, residual folded into the accumulator - what the engine launches at C = 192) and the
    resident-weights kernel of the narrow blocks (300) on ragged pixel counts incl. several persistent rounds, against fp64."""
    import ctypes as C
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_time_mixer.restype = C.c_float
    lib.rd_debug_time_mixer.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6
    g = torch.Generator(device="cuda").manual_seed(C_ + M)
    x = torch.rand((M, C_), device="cuda", generator=g) - 0.5
    w1 = (torch.rand((2 * C_, C_), device="cuda", generator=g) - 0.5) * 0.2
    w2 = (torch.rand((C_, 2 * C_), device="cuda", generator=g) - 0.5) * 0.2
    b1 = torch.rand(2 * C_, device="cuda", generator=g) - 0.5
    b2 = torch.rand(C_, device="cuda", generator=g) - 0.5
    y = torch.full((M + 4, C_), 555.0, device="cuda")
    lib.rd_debug_time_mixer(C_, M, variant, 1, x.data_ptr(), y.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr())
    torch.cuda.synchronize()
    xd = x.double()
    ref = xd + torch.nn.functional.gelu(xd @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    assert float((y[:M].double() - ref).abs().max()) < (2e-6 if variant == 0 else 1e-6)
    assert float((y[M:] - 555.0).abs().max()) == 0.0


@pytest.mark.parametrize("C_,variant", [(96, 300), (192, 400), (96, 200), (192, 200)])
@pytest.mark.parametrize("xs", [1.0, 1e-3, 1e-5])
def test_mixer_residual_from_split_fragments_small_inputs(C_, variant, xs):
    """The resident-weights (300) and weight-streaming (200 / 400) mixers re-form the residual x from the (hi, lo) fragments they already
    hold instead of reading the tile again (round 5).  hi + lo differs from x by lo's rounding: <= 2^-22 |x| where lo is a normal fp16,
    an ABSOLUTE 2^-25 (res kernel; ws: 2^-25 / the pixel's scale) where it is subnormal - so the output's error against fp64 stays
    bounded by that plus the products' own 2^-21-relative error, for activations of ordinary and of small magnitude."""
    import ctypes as C
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_time_mixer.restype = C.c_float
    lib.rd_debug_time_mixer.argtypes = [C.c_int] * 4 + [C.c_void_p] * 6
    M = 4099
    g = torch.Generator(device="cuda").manual_seed(C_ + variant)
    x = (torch.rand((M, C_), device="cuda", generator=g) - 0.5) * 2 * xs
    x[:, ::7] *= 1e-3                                        # channels far below their pixel's largest
    w1 = (torch.rand((2 * C_, C_), device="cuda", generator=g) - 0.5) * 0.2
    w2 = (torch.rand((C_, 2 * C_), device="cuda", generator=g) - 0.5) * 0.2
    b1 = (torch.rand(2 * C_, device="cuda", generator=g) - 0.5) * xs
    b2 = (torch.rand(C_, device="cuda", generator=g) - 0.5) * xs
    y = torch.zeros((M, C_), device="cuda")
    lib.rd_debug_time_mixer(C_, M, variant, 1, x.data_ptr(), y.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr())
    torch.cuda.synchronize()
    xd = x.double()
    ref = xd + torch.nn.functional.gelu(xd @ w1.double().t() + b1.double()) @ w2.double().t() + b2.double()
    err = (y.double() - ref).abs()
    # per element: 2^-22 |x| (residual) + 2^-25 (subnormal low plane of the residual) + the mixer's own product error, 1e-6 of the
    # output scale + what the SAME subnormal low planes do to the two products: every operand element whose low plane is an fp16
    # subnormal (|x| < 2^-3 in the unscaled res kernel; the small channels of a pixel in ws) enters its product with an absolute
    # rounding error uniform in +-2^-25, which reaches the output through one row of W1, GELU' (<= 1.13) and one row of W2, and the
    # hidden activation's own low plane through one row of W2: 6 sigma of that sum (an absolute 1e-7..2e-7 at these weights - fp32's
    # own eps at |y| ~ 1, whatever the tensor's scale: the guard's range flag covers overflow, underflow costs this much)
    sig = 2.0 ** -25 / 3 ** 0.5
    r1 = float(w1.double().norm(dim=1).max()); r2 = float(w2.double().norm(dim=1).max())
    bound = 2.0 ** -22 * xd.abs() + 2.0 ** -25 + 1e-6 * float(ref.abs().max()) + 6 * sig * r2 * (1.0 + 1.13 * r1)
    assert bool((err <= bound).all()), (float(err.max()), float((err - bound).max()), float(bound.min()))
    print(f"mixer small-input error C={C_} variant={variant} xs={xs}: max err {float(err.max()):.3e}, min bound {float(bound.min()):.3e}")


def test_forward_is_deterministic(engines, golden_dir):
    """Same input, same handle, twice: bit-identical outputs (no atomics or run-to-run scheduling in the arithmetic; the SE
    pooling is a fixed two-stage reduction, the split-fp16 kernels only use an atomic for the range flag)."""
    eng, _ = engines["ppocrv6_rec"]
    x = torch.from_numpy(np.load(golden_dir / "rec_seed0_b3_w640.npz")["x"]).cuda()
    a = [t.clone() for t in eng.rec_forward(x)[:2]]
    b = eng.rec_forward(x)[:2]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    eng, _ = engines["ppocrv6_det"]
    x = torch.from_numpy(np.load(golden_dir / "det_seed0_b2_96x160.npz")["x"]).cuda()
    assert torch.equal(eng.det_forward(x).clone(), eng.det_forward(x))


def test_region_ocr_flow_on_given_layout(golden_dir):
    """The OCR branch of BatchAnalyze for given layout detections (rapiddoc_amd/analyze.py): crops with a 50-px white
    margin, 64-px size groups, det + DB post-process, box sort / merge / formula cut, page-coordinate spans, rec.  Random
    weights give meaningless det maps, so the maps are rendered from the generator's line boxes; checked: one span per text
    line, polys back in page coordinates, formula boxes cut out, schema and score conventions of the reference."""
    from rapiddoc_amd.analyze import LOW_SCORE_TEXT, OCR_TEXT, RegionOcr
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline
    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, rec_batch_num=32)
    pages_np, boxes = synth_batch(0, 2)
    pages = torch.from_numpy(pages_np).cuda()
    dets_per_page, expect = [], []
    for p in range(2):
        b = np.asarray(boxes[p], dtype=np.float64).reshape(-1, 4)
        thirds = np.array_split(np.arange(len(b)), 3)
        dets, n_lines = [], 0
        for t, idx in enumerate(thirds):                      # three text regions, each the hull of its lines
            x0, y0, x1, y1 = b[idx, 0].min() - 6, b[idx, 1].min() - 6, b[idx, 2].max() + 6, b[idx, 3].max() + 6
            dets.append({"category_id": 1, "original_label": "text", "original_order": t, "score": 0.9,
                         "poly": [x0, y0, x1, y0, x1, y1, x0, y1]})
            n_lines += len(idx)
        l0 = b[thirds[0][0]]                                  # an inline formula in the middle of the first line
        fx0, fx1 = l0[0] + 0.4 * (l0[2] - l0[0]), l0[0] + 0.6 * (l0[2] - l0[0])
        dets.append({"category_id": 13, "original_label": "inline_formula", "original_order": 9, "score": 0.9,
                     "poly": [fx0, l0[1] - 1, fx1, l0[1] - 1, fx1, l0[3] + 1, fx0, l0[3] + 1]})
        dets_per_page.append(dets)
        expect.append(n_lines + 1)                            # the cut line becomes two spans

    def maps_fn(regions, ghw, dhw):
        (gh, gw), (dh, dw) = ghw, dhw
        m = torch.zeros((len(regions), 1, dh, dw), dtype=torch.float32)
        for k, (p, r, useful) in enumerate(regions):
            px, py, x0, y0 = useful[:4]
            for lb in np.asarray(boxes[p], dtype=np.float64).reshape(-1, 4):
                if lb[0] >= r["poly"][0] and lb[2] <= r["poly"][4] and lb[1] >= r["poly"][1] and lb[3] <= r["poly"][5]:
                    cx0, cy0, cx1, cy1 = lb[0] - x0 + px, lb[1] - y0 + py, lb[2] - x0 + px, lb[3] - y0 + py
                    sy, sx = dh / gh, dw / gw
                    # the DB map marks the SHRUNK text kernel (pipeline.render_text_maps: 0.32 x height per side)
                    d = 0.32 * min(cx1 - cx0, cy1 - cy0)
                    m[k, 0, int(round((cy0 + d) * sy)):int(round((cy1 - d) * sy)), int(round((cx0 + d) * sx)):int(round((cx1 - d) * sx))] = 0.95
        return m.cuda()

    out = RegionOcr(pipe)(pages, dets_per_page, det_maps_fn=maps_fn)
    for p in range(2):
        spans = [d for d in out[p] if d["category_id"] in (OCR_TEXT, LOW_SCORE_TEXT)]
        assert out[p][: len(dets_per_page[p])] == dets_per_page[p]        # layout detections are passed through
        assert len(spans) == expect[p], (len(spans), expect[p])
        lines = np.asarray(boxes[p], dtype=np.float64).reshape(-1, 4)
        for s in spans:
            assert set(s) == {"category_id", "original_label", "original_order", "poly", "score", "text"}
            assert isinstance(s["text"], str) and 0.0 <= s["score"] <= 1.0 and s["score"] == float(f"{s['score']:.3f}")
            assert (s["category_id"] == LOW_SCORE_TEXT) == (s["score"] < 0.5)
            x0, y0, x1, y1 = s["poly"][0], s["poly"][1], s["poly"][4], s["poly"][5]
            # every span sits on a generator line (page coordinates), within the DB unclip slack
            hit = (np.abs(lines[:, 1] - y0) < 8) & (np.abs(lines[:, 3] - y1) < 8) & (lines[:, 0] - 8 <= x0) & (x1 <= lines[:, 2] + 8)
            assert hit.any(), s["poly"]
        # the formula's x-range is not covered by any span of its line
        f = dets_per_page[p][3]["poly"]
        on_line = [s for s in spans if abs(s["poly"][1] - (f[1] + 1)) < 8]
        assert len(on_line) == 2 and all(s["poly"][4] <= f[0] + 1 or s["poly"][0] >= f[4] - 1 for s in on_line)


def test_formula_branch_on_given_layout(golden_dir):
    """batch_analyze.py:258-283: formula regions are cropped (grown by 2 px unless a neighbour is in the way), recognised and
    get a `latex` field; other detections are untouched.  Token ids stand in for the string (the tokenizer is download-only)."""
    from rapiddoc_amd.analyze import recognise_formulas
    from rapiddoc_amd.formula_host import FormulaRecognizer
    man = W.load_manifest(golden_dir / "manifest_ppformulanet_plus_m_m8.json")
    rec = FormulaRecognizer(W.synth_state_dict(man, 0), max_new_tokens=6)
    rng = np.random.default_rng(5)
    pages = torch.from_numpy(rng.integers(0, 255, (2, 400, 600, 3), dtype=np.uint8)).cuda()
    pages[:, 100:160, 200:420] = 255
    pages[:, 120:140, 230:390] = 30                                        # a dark "formula" on white
    mk = lambda cid, x0, y0, x1, y1: {"category_id": cid, "poly": [x0, y0, x1, y0, x1, y1, x0, y1], "score": 0.9}
    dets = [[mk(14, 200, 100, 420, 160), mk(1, 20, 20, 580, 90), mk(13, 230, 300, 300, 330)], [mk(8, 200, 100, 420, 160)]]
    before = [[dict(d) for d in page] for page in dets]
    n = recognise_formulas(pages, dets, rec)
    assert n == 3
    for page, page0 in zip(dets, before):
        for d, d0 in zip(page, page0):
            if d["category_id"] in (8, 13, 14):
                assert {k: v for k, v in d.items() if k != "latex"} == d0
                if "latex" in d:
                    assert isinstance(d["latex"], list) and len(d["latex"]) <= 6 and all(isinstance(t, int) for t in d["latex"])
            else:
                assert d == d0 and "latex" not in d
    assert "latex" in dets[0][0] and "latex" in dets[1][0]                # same crop content on both pages
    assert dets[0][0]["latex"] == dets[1][0]["latex"]


def test_region_text_model_is_a_custom_ocr_model(golden_dir):
    """Seam S1: `batch_predict(list of BGR region crops) -> list[str]`, one (multi-line) string per region, order kept,
    ragged sizes grouped by 64-px buckets; empty regions give an empty string."""
    from rapiddoc_amd.analyze import RegionTextModel
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline
    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0) for k in ("ppocrv6_det", "ppocrv6_rec")}
    model = RegionTextModel(PagePipeline(states, rec_batch_num=32))
    pages_np, boxes = synth_batch(0, 1)
    b = np.asarray(boxes[0], dtype=np.float64).reshape(-1, 4)
    crops, want = [], []
    for lo, hi in ((0, 5), (5, 6), (6, 20)):                       # three regions of 5, 1 and 14 lines
        x0, y0, x1, y1 = int(b[lo:hi, 0].min()) - 8, int(b[lo:hi, 1].min()) - 8, int(b[lo:hi, 2].max()) + 8, int(b[lo:hi, 3].max()) + 8
        crops.append(np.ascontiguousarray(pages_np[0, y0:y1, x0:x1, ::-1]))       # BGR like cv2.cvtColor(.., RGB2BGR)
        want.append((hi - lo, (x0, y0)))
    crops.append(np.full((40, 300, 3), 255, np.uint8))             # a blank region

    def maps_fn(ids, ghw, dhw):
        (gh, gw), (dh, dw) = ghw, dhw
        m = torch.zeros((len(ids), 1, dh, dw), dtype=torch.float32)
        for k, i in enumerate(ids):
            if i >= len(want):
                continue
            n, (ox, oy) = want[i]
            lo = (0, 5, 6)[i]
            for lb in b[lo:lo + n]:
                cx0, cy0, cx1, cy1 = lb[0] - ox, lb[1] - oy, lb[2] - ox, lb[3] - oy
                d = 0.32 * min(cx1 - cx0, cy1 - cy0)
                m[k, 0, int(round((cy0 + d) * dh / gh)):int(round((cy1 - d) * dh / gh)),
                  int(round((cx0 + d) * dw / gw)):int(round((cx1 - d) * dw / gw))] = 0.95
        return m.cuda()

    out = model.batch_predict(crops, det_maps_fn=maps_fn, batch_size=4)
    assert isinstance(out, list) and len(out) == 4 and all(isinstance(t, str) for t in out)
    assert out[3] == ""
    for (n, _), text in zip(want, out[:3]):
        assert len(text.split("\n")) <= n                           # never more lines than text lines in the region


def test_rec_two_stage_equals_single_stage(engines):
    """rd_rec_backbone_forward per batch into one token buffer + ONE rd_rec_tail_forward over the lines of all batches
    (different widths => different line lengths) == rd_rec_forward batch by batch."""
    eng, _ = engines["ppocrv6_rec"]
    rng = np.random.default_rng(21)
    shapes = [(5, 3, 48, 328), (3, 3, 48, 96), (2, 3, 48, 1200), (1, 3, 48, 16)]
    xs = [torch.from_numpy(rng.uniform(-1, 1, s).astype(np.float32)).cuda() for s in shapes]
    singles = [eng.rec_forward(x) for x in xs]
    dim = eng.rec_token_dim
    lens = []
    for x, (idx, _p, _f) in zip(xs, singles):
        lens += [idx.shape[1]] * x.shape[0]
    tokens = torch.empty((sum(lens), dim), dtype=torch.float32, device="cuda")
    pos = 0
    for x, (idx, _p, _f) in zip(xs, singles):
        n = x.shape[0] * idx.shape[1]
        out = eng.rec_backbone_forward(x, tokens[pos: pos + n])
        assert out.data_ptr() == tokens[pos: pos + n].data_ptr()
        pos += n
    idx2, prob2 = eng.rec_tail_forward(tokens, lens)
    pos = 0
    for x, (idx, prob, _f) in zip(xs, singles):
        n = idx.numel()
        assert (idx2[pos: pos + n].cpu().numpy() == idx.reshape(-1).cpu().numpy()).all()
        # bit for bit (round 6): every kernel of the tail is picked by the layer / the line's own length, the class split of the CTC head by
        # the dictionary - nothing by the token count of the launch
        assert np.array_equal(prob2[pos: pos + n].cpu().numpy(), prob.reshape(-1).cpu().numpy())
        pos += n


def test_pipeline_two_stage_rec_gives_the_single_stage_results(engines, golden_dir):
    """PagePipeline with the recogniser in two stages (backbone per batch, neck + CTC head once per group of batches, the
    default) decodes the same strings as the whole network batch by batch, with the same confidences - bit for bit since round 6: the
    matrix kernels are picked by the layer and the class split of the fused CTC head by the dictionary, not by the row / token count
    (rounds 1-5: a 16-line batch had < 2048 tokens => native fp32 MFMA, the group of three batches got the split-fp16 kernels)."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0)
              for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, rec_batch_num=16, n_rec_streams=3)
    pages_np, boxes = synth_batch(11, 2)
    pages = torch.from_numpy(pages_np).cuda()
    det_hw = pipe.det_forward(pages[:1])[1]
    maps = render_text_maps(boxes, pages_np.shape[1:3], det_hw, pages.device)
    out = {}
    for two in (True, False):
        pipe.rec_two_stage = two
        pipe.keep_rec_inputs = True
        res = pipe.run_batch(pages, None, det_maps_override=maps)
        out[two] = ([[(t, s) for _q, t, s in r.lines] for r in res],
                    [(c.copy(), i.cpu().numpy().copy(), p.cpu().numpy().copy()) for c, _x, i, p in pipe.last_rec_batches])
    assert [len(p) for p in out[True][0]] == [45, 45]
    for (ca, ia, pa), (cb, ib, pb) in zip(out[True][1], out[False][1]):
        assert (ca == cb).all() and (ia == ib).all()
        assert np.array_equal(pa, pb)
    for pa, pb in zip(out[True][0], out[False][0]):
        assert pa == pb


def _debug_conv(x_nhwc, w_oihw, bias, stride, pads, act=0, res=None, split=True, iters=0, force_direct=False, force_stream=False):
    """One convolution through the library's dense-conv launcher on prepared operands (api.cpp rd_debug_conv)."""
    import ctypes as C
    from rapiddoc_amd import _lib
    lib = _lib.load()
    lib.rd_debug_conv.restype = C.c_float
    lib.rd_debug_conv.argtypes = [C.c_int] * 14 + [C.c_void_p] * 7 + [C.c_void_p]
    N, H, W_, Cin = x_nhwc.shape
    Cout, _, KH, KW = w_oihw.shape
    K = KH * KW * Cin
    wf = w_oihw.permute(0, 2, 3, 1).reshape(Cout, K).contiguous()          # k = (kh * KW + kw) * Cin + ci
    Kp = (K + 31) // 32 * 32
    wh = wl = None
    if split:
        hi = wf.half()
        lo = ((wf - hi.float()) * 2048.0).half()
        wh = torch.zeros((Cout, Kp), dtype=torch.float16, device="cuda"); wh[:, :K] = hi
        wl = torch.zeros((Cout, Kp), dtype=torch.float16, device="cuda"); wl[:, :K] = lo
    pt, pl, pb, pr = pads
    OH, OW = (H + pt + pb - KH) // stride + 1, (W_ + pl + pr - KW) // stride + 1
    y = torch.full((N, OH, OW, Cout), float("nan"), device="cuda")
    used = C.c_int(2 if force_stream else 1 if force_direct else 0)
    ms = lib.rd_debug_conv(N, H, W_, Cin, Cout, KH, KW, stride, pt, pl, pb, pr, act, iters, x_nhwc.data_ptr(), wf.data_ptr(),
                           wh.data_ptr() if split else None, wl.data_ptr() if split else None, bias.data_ptr(),
                           res.data_ptr() if res is not None else None, y.data_ptr(), C.byref(used))
    torch.cuda.synchronize()
    return y, used.value, ms


@pytest.mark.parametrize("N,H,W_,Cin,Cout,k,pads,act,with_res", [
    (2, 50, 67, 48, 48, 3, (1, 1, 1, 1), 1, False),      # B4 stages.0 3x3: two 32-blocks, the second half empty
    (2, 41, 90, 96, 96, 3, (1, 1, 1, 1), 1, True),       # B4 stages.1: two channel passes, three blocks, residual
    (3, 24, 130, 48, 24, 2, (0, 0, 1, 1), 1, False),     # rec stem2a: 2x2, pad right / bottom only
    (3, 24, 130, 24, 48, 2, (0, 0, 1, 1), 1, False),     # rec stem2b: Cin 24 padded to 32 in LDS
    (2, 12, 333, 64, 32, 3, (1, 1, 1, 1), 0, False),     # 4-row tiles (OH = 12), Cin 64 in one pass
    (1, 96, 96, 192, 24, 3, (1, 1, 1, 1), 3, False),     # Cin 192 = 3 passes of 64, SiLU
])
def test_direct_conv_matches_fp64(N, H, W_, Cin, Cout, k, pads, act, with_res):
    """kernels_conv_direct_h3.hip against torch conv2d in float64: fp32-class error (the split keeps 22 of 24 mantissa bits
    per operand, fp32 accumulate) at ragged sizes (tile edges in both directions, partial channel blocks)."""
    g = torch.Generator(device="cuda").manual_seed(N * 1000 + Cin)
    x = torch.rand((N, H, W_, Cin), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((Cout, Cin, k, k), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand((Cout,), device="cuda", generator=g) - 0.5
    pt, pl, pb, pr = pads
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2).double(), (pl, pr, pt, pb))
    ref = torch.nn.functional.conv2d(xp, w.double(), b.double())
    ref = {0: ref, 1: torch.relu(ref), 3: torch.nn.functional.silu(ref)}[act]
    res = None
    if with_res:
        res = torch.rand((N, ref.shape[2], ref.shape[3], Cout), device="cuda", generator=g)
        ref = ref + res.permute(0, 3, 1, 2).double()
    y, used, _ = _debug_conv(x, w, b, 1, pads, act, res, force_direct=True)
    assert used == 1, "the direct kernel did not take this geometry"
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err


@pytest.mark.parametrize("N,H,W_,Cin,Cout,k,stride,pads,act,with_res", [
    (3, 24, 130, 48, 24, 2, 1, (0, 0, 1, 1), 1, False),     # rec stem2a
    (3, 24, 130, 24, 48, 2, 1, (0, 0, 1, 1), 1, False),     # rec stem2b: Cin 24 -> two k-steps, the second half masked
    (2, 33, 71, 12, 24, 2, 1, (0, 0, 1, 1), 1, False),      # det stem2b: Cin 12
    (2, 40, 57, 48, 96, 1, 1, (0, 0, 0, 0), 1, True),       # stem4 1x1, three output blocks, residual
    (2, 37, 53, 16, 32, 3, 2, (1, 1, 1, 1), 3, False),      # 3x3 stride 2 (any geometry is address arithmetic here)
])
def test_stream_conv_matches_fp64(N, H, W_, Cin, Cout, k, stride, pads, act, with_res):
    """kernels_conv_stream_h3.hip (small-K layers, operands streamed from global memory) against torch conv2d in float64."""
    g = torch.Generator(device="cuda").manual_seed(N * 77 + Cin)
    x = torch.rand((N, H, W_, Cin), device="cuda", generator=g) * 2 - 1
    w = (torch.rand((Cout, Cin, k, k), device="cuda", generator=g) - 0.5) * 0.2
    b = torch.rand((Cout,), device="cuda", generator=g) - 0.5
    pt, pl, pb, pr = pads
    xp = torch.nn.functional.pad(x.permute(0, 3, 1, 2).double(), (pl, pr, pt, pb))
    ref = torch.nn.functional.conv2d(xp, w.double(), b.double(), stride=stride)
    ref = {0: ref, 1: torch.relu(ref), 3: torch.nn.functional.silu(ref)}[act]
    res = None
    if with_res:
        res = torch.rand((N, ref.shape[2], ref.shape[3], Cout), device="cuda", generator=g)
        ref = ref + res.permute(0, 3, 1, 2).double()
    y, used, _ = _debug_conv(x, w, b, stride, pads, act, res, force_stream=True)
    assert used == 2, "the streaming kernel did not take this geometry"
    err = (y.permute(0, 3, 1, 2).double() - ref).abs().max().item()
    assert err < 2e-5 * max(1.0, ref.abs().max().item()), err
