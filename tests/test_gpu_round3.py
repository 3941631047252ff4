"""GPU parity at the sizes and in the modes that are actually run ("Next round" #1):

 (a) PP-FormulaNet_plus-M at BASELINE config 3's sizes: B6 encoder on [2,1,384,384] (144 encoder states) in `auto` and
     `fp32`, decoder for 300 generated tokens against the REFERENCE's own ids (fixture minted by make_golden_r3.py);
 (b) det [32,3,960,704] and B4 [32,3,800,800] through the very engine objects / plans `bench.py` builds: 3 of the 32
     outputs against the oracle, and 3 of the step's rec batches (64 lines each) against the oracle on the tensors the
     crop kernels produced;
 (c) the strict reference rec batching (one global argsort, chunks of 6, width int(48 * max ratio)): the pipeline's
     batches equal an independent restatement of rapid_ocr.py:404-449, (idx, prob, strings) equal the oracle's on those
     reference-chunked tensors.
"""
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


def _states(golden_dir, kinds):
    return {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0) for k in kinds}


# ---------------------------------------------------------------------------------------------------------------------
# (a) formula path at full size
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("precision", ["auto", "fp32"])
def test_formula_encoder_fullsize_384(golden_dir, precision, monkeypatch):
    """PPHGNetV2_B6_Formula on the 384 x 384 grey formula image of UniMERNet's pre-process (rec_pphgnetv2.py:1587-1642):
    [2,1,384,384] -> [2,144,2048]."""
    monkeypatch.setenv("RD_PRECISION", "auto")
    eng, st = _engine(golden_dir, "pphgnetv2_b6_formula", precision)
    x = torch.from_numpy(np.random.default_rng(21).uniform(-1, 1, (2, 1, 384, 384)).astype(np.float32))
    with torch.no_grad():
        ref = O.formula_encoder_forward(st, x).numpy()
    enc = eng.formula_encoder_forward(x.cuda()).cpu().numpy()
    assert not eng.range_overflow()
    assert enc.shape == ref.shape == (2, 144, 2048)
    assert np.abs(enc - ref).max() < TOL * max(1.0, float(np.abs(ref).max()))


def test_formula_decoder_300_tokens_equals_reference(golden_dir):
    """MBart decoder with its KV cache grown to 300 positions, 144 cross-attention keys, one sequence ending early (EOS,
    then padding) and one running to max_new_tokens: token ids == the reference head's generate_export
    (rec_ppformulanet_head.py:1054-1176), and every step's choice is the argmax of the oracle's teacher-forced logits."""
    from oracle import formula as OF
    from rapiddoc_amd.engine import RdEngine
    from test_oracle_golden import formula_long_case, live_steps
    st, enc, g = formula_long_case(golden_dir)
    eng = RdEngine("ppformulanet_head").load_weights(st)
    assert eng.formula_max_new_tokens == 300
    ids = eng.formula_decode(torch.from_numpy(enc).cuda(), 300).cpu().numpy()
    ref = g["ids"]
    assert ids.shape == ref.shape == (2, 301)
    live = live_steps(ref)
    safe = live & (g["top2gap"] > 1e-2)
    for b in range(2):                                  # identical up to the first step whose top-2 gap is inside fp32 noise
        unsafe = np.nonzero(live[b] & ~safe[b])[0]
        n = (unsafe[0] if len(unsafe) else 300) + 1
        assert (ids[b, :n] == ref[b, :n]).all(), (b, np.nonzero(ids[b] != ref[b])[0][:5])
    assert safe[0].all() and (ids[0] == ref[0]).all()   # this fixture's long sequence has no unsafe step at all
    assert (ids[1] == ref[1]).all()                     # EOS at the same step, padded the same way
    # independent of the fixture's path: each token the GPU chose is (within noise) the oracle's argmax for the GPU's own prefix
    with torch.no_grad():
        lg = OF.teacher_forced_logits(O.as_torch_state(st), torch.from_numpy(enc), torch.from_numpy(ids)).numpy()
    chosen = np.take_along_axis(lg, ids[:, 1:, None], axis=2)[..., 0]
    live_gpu = live_steps(ids)
    assert ((lg.max(axis=2) - chosen)[live_gpu] < 2e-3).all()


# ---------------------------------------------------------------------------------------------------------------------
# (b) the plans the benchmark runs: 32 pages
# ---------------------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def bench_pipe(golden_dir):
    """The pipeline exactly as bench.py builds it (adaptive rec chunking, width multiple 32, 8 rec streams) + its 32 pages."""
    from rapiddoc_amd.pages import synth_pages
    from rapiddoc_amd.pipeline import PagePipeline
    states = _states(golden_dir, ("ppocrv6_det", "ppocrv6_rec", "pphgnetv2_b4"))
    pipe = PagePipeline(states, rec_batch_num=64, rec_width_multiple=32, n_rec_streams=8, rec_chunking="adaptive")
    pages_np, boxes = synth_pages(list(range(32)))
    return pipe, states, torch.from_numpy(pages_np).cuda(), pages_np, boxes


def test_bench_plan_det_b32_matches_oracle(bench_pipe):
    from rapiddoc_amd import ocr_host
    from rapiddoc_amd.engine import preproc_resize_norm
    pipe, states, pages, _np, _boxes = bench_pipe
    x, (dh, dw) = pipe.det_preprocess(pages)
    assert tuple(x.shape) == (32, 3, 960, 704)
    maps = pipe.det.det_forward(x)
    assert not pipe.det.check_range_and_fallback()
    st = O.as_torch_state(states["ppocrv6_det"])
    for i in (0, 13, 31):
        # the batched pre-process launch == the per-image one (bit for bit; that one is pinned to oracle/cv2_ops.py)
        one = preproc_resize_norm(pages[i], (dh, dw), mean=ocr_host.DET_MEAN, std=ocr_host.DET_STD, interp=1, swap_rb=True)
        assert torch.equal(one, x[i])
        with torch.no_grad():
            ref = O.det_forward(st, x[i:i + 1].cpu()).numpy()
        assert np.abs(maps[i:i + 1].cpu().numpy() - ref).max() < TOL


def test_bench_plan_b4_b32_matches_oracle(bench_pipe):
    from rapiddoc_amd.engine import preproc_resize_norm
    pipe, states, pages, _np, _boxes = bench_pipe
    x = pipe.layout_preprocess(pages)
    assert tuple(x.shape) == (32, 3, 800, 800)
    feats = pipe.layout.backbone_forward(x)
    assert not pipe.layout.check_range_and_fallback()
    st = O.as_torch_state(states["pphgnetv2_b4"])
    for i in (0, 13, 31):
        assert torch.equal(preproc_resize_norm(pages[i], (800, 800), interp=2), x[i])
        with torch.no_grad():
            ref = O.pphgnetv2_features(st, x[i:i + 1].cpu())
        for r, f in zip(ref, feats):
            assert np.abs(f[i:i + 1].cpu().numpy() - r.numpy()).max() < TOL


def _check_rec_batches_against_oracle(pipe, st_rec, batch_ids, flat_lines):
    from rapiddoc_amd import ocr_host
    for bi in batch_ids:
        chunk, x, idx, prob = pipe.last_rec_batches[bi]
        with torch.no_grad():
            lg = O.rec_forward(st_rec, x.cpu())
        ridx, rprob = O.ctc_greedy_stats(lg)
        top2 = torch.topk(lg, 2, dim=2).values
        safe = ((top2[..., 0] - top2[..., 1]) > 1e-2).numpy()
        assert (idx.cpu().numpy() == ridx.numpy())[safe].all()
        assert np.abs(prob.cpu().numpy() - rprob.numpy())[safe].max() < TOL
        dec = ocr_host.ctc_decode(idx.cpu().numpy(), prob.cpu().numpy(), pipe.characters)
        for j, i in enumerate(chunk.tolist()):
            assert flat_lines[i][1] == dec[j][0]
            assert flat_lines[i][2] == ocr_host.format_score(dec[j][1])


@pytest.mark.parametrize("chunking", ["adaptive", "fixed"])
def test_bench_step_rec_batches_match_oracle(bench_pipe, chunking):
    """One whole benchmark step (32 pages, 1440 lines on 8 streams, two-stage recogniser) in the chunking bench.py runs ("adaptive":
    chunk sizes chosen per width so that the persistent kernels fill whole rounds of the chip) and in fixed chunks of 64 (23
    batches): the first, a middle and the last rec batch against the oracle on the tensors the crop kernels produced, and the
    strings the step returned are the decode of those (idx, prob).  Two steps give the same result (what bench.py's result_crc32
    hashes)."""
    from rapiddoc_amd.pipeline import render_text_maps
    pipe, states, pages, pages_np, boxes = bench_pipe
    det_hw = pipe.det_preprocess(pages[:1])[1]
    maps = render_text_maps(boxes, pages_np.shape[1:3], det_hw, pages.device)
    pipe.keep_rec_inputs = True
    pipe.rec_chunking = chunking
    try:
        res = pipe.run_batch(pages, None, det_maps_override=maps)
        assert [len(r.lines) for r in res] == [45] * 32
        nb = len(pipe.last_rec_batches)
        sizes = [len(c) for c, *_ in pipe.last_rec_batches]
        assert sum(sizes) == 1440 and (nb == 23 if chunking == "fixed" else (all(16 <= n <= 160 for n in sizes) and nb < 23))
        flat = [ln for r in res for ln in r.lines]
        _check_rec_batches_against_oracle(pipe, O.as_torch_state(states["ppocrv6_rec"]), (0, nb // 2, nb - 1), flat)
        again = pipe.run_batch(pages, None, det_maps_override=maps)
        assert [[(t, s) for _q, t, s in r.lines] for r in again] == [[(t, s) for _q, t, s in r.lines] for r in res]
    finally:
        pipe.keep_rec_inputs = False
        pipe.last_rec_batches = []
        pipe.rec_chunking = "adaptive"


# ---------------------------------------------------------------------------------------------------------------------
# (c) strict reference rec batching
# ---------------------------------------------------------------------------------------------------------------------
def _reference_rec_chunks(crop_hw, rec_batch_num=6):
    """rapid_ocr.py:404-449 restated independently of rapiddoc_amd: [(indices, imgW)] for crops of the given (h, w)."""
    width_list = [w / float(h) for h, w in crop_hw]
    indices = np.argsort(np.array(width_list))
    out = []
    for beg in range(0, len(crop_hw), rec_batch_num):
        end = min(len(crop_hw), beg + rec_batch_num)
        max_wh_ratio = 320 / 48
        for ino in range(beg, end):
            h, w = crop_hw[indices[ino]]
            max_wh_ratio = max(max_wh_ratio, w * 1.0 / h)
        out.append(([int(indices[i]) for i in range(beg, end)], int(48 * max_wh_ratio)))
    return out


def test_strict_rec_mode_is_the_reference_batching(golden_dir):
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = _states(golden_dir, ("ppocrv6_det", "ppocrv6_rec"))
    pipe = PagePipeline(states, rec_mode="strict", n_rec_streams=4, rec_batch_num=64)     # rec_batch_num is ignored in strict mode
    assert pipe.rec_batch_num == 6 and pipe.rec_width_multiple == 1
    pipe.rec_two_stage = False           # the whole-network form: one launch per distinct padded width (tests/test_gpu_rec_lines.py: the default form)
    pipe.keep_rec_inputs = True
    pages_np, boxes = synth_batch(7, 3)
    pages = torch.from_numpy(pages_np).cuda()
    maps = render_text_maps(boxes, pages_np.shape[1:3], pipe.det_preprocess(pages[:1])[1], pages.device)
    res = pipe.run_batch(pages, None, det_maps_override=maps)
    flat = [ln for r in res for ln in r.lines]
    n = len(flat)
    assert n == 135
    cw, ch, rot, keep = pipe.last_rec_crop_sizes
    assert len(keep) == n
    crop_hw = [(int(cw[i]), int(ch[i])) if rot[i] else (int(ch[i]), int(cw[i])) for i in range(n)]   # (h, w) of the image rec sees
    expected = _reference_rec_chunks(crop_hw)
    assert len(expected) == 23
    want_w = {}
    for idxs, w in expected:
        for i in idxs:
            want_w[i] = w
    got_w, launched = {}, 0
    for chunk, x, _idx, _prob in pipe.last_rec_batches:
        launched += 1
        assert x.shape[1:3] == (3, 48)
        same_w = [i for idxs, w in expected if w == x.shape[3] for i in idxs]      # the reference chunks of this width, in order
        assert chunk.tolist() == same_w
        for i in chunk.tolist():
            got_w[int(i)] = int(x.shape[3])
    assert got_w == want_w                                    # every line sees exactly the padded width the reference gives it
    assert launched == len({w for _i, w in expected}) <= 23   # equal widths share a launch
    _check_rec_batches_against_oracle(pipe, O.as_torch_state(states["ppocrv6_rec"]), range(len(pipe.last_rec_batches)), flat)
    # zero right-padding starts at min(imgW, ceil(48 * w / h)) (resize_norm_img)
    chunk, x, _i, _p = pipe.last_rec_batches[-1]
    for j, i in enumerate(chunk.tolist()):
        h, w = crop_hw[i]
        rw = min(x.shape[3], int(np.ceil(48 * (w / float(h)))))
        assert float(x[j, :, :, rw:].abs().max()) == 0.0 if rw < x.shape[3] else True


def test_region_ocr_pools_the_lines_of_all_size_groups(golden_dir):
    """RegionOcr recognises the lines of every region of the page batch in ONE pooled call ordered page by page
    (analyze_utils.py:216-252), although the regions were detected in separate 64-px size groups."""
    from rapiddoc_amd.analyze import RegionOcr
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = _states(golden_dir, ("ppocrv6_det", "ppocrv6_rec"))
    pipe = PagePipeline(states, rec_mode="strict", n_rec_streams=2)
    pages_np, boxes = synth_batch(3, 2)
    pages = torch.from_numpy(pages_np).cuda()

    def region(x0, y0, x1, y1, order):
        return {"category_id": 1, "original_label": "text", "original_order": order,
                "poly": [x0, y0, x1, y0, x1, y1, x0, y1], "score": 0.9}
    # two regions of different sizes per page (the full-width lines above the figure, the short ones beside it) -> different
    # 64-px buckets -> separate det groups
    dets = [[region(60, 50, 1140, 985, 0), region(60, 990, 645, 1420, 1)] for _ in range(2)]

    def maps_fn(regs, ghw, dhw):
        per_img = []
        for p, r, useful in regs:
            px, py, x0, y0 = useful[0], useful[1], useful[2], useful[3]
            bb = np.asarray(boxes[p], dtype=np.float64).reshape(-1, 4)
            x1, y1 = r["poly"][4], r["poly"][5]
            inside = bb[(bb[:, 0] >= x0) & (bb[:, 2] <= x1) & (bb[:, 1] >= y0) & (bb[:, 3] <= y1)]
            per_img.append(inside - [x0 - px, y0 - py, x0 - px, y0 - py])
        return render_text_maps(per_img, ghw, dhw, pages.device)
    calls = []
    orig = pipe.rec_forward_sources
    pipe.rec_forward_sources = lambda sources, image_keys=None: (calls.append((len(sources), image_keys)), orig(sources, image_keys))[1]
    out = RegionOcr(pipe)(pages, dets, det_maps_fn=maps_fn)
    assert len(calls) == 1 and calls[0][0] == 2 and calls[0][1] == [[0, 1], [0, 1]]      # one pooled call, two size groups
    spans = [[d for d in page if d["category_id"] in (15, 16)] for page in out]
    assert all(len(s) > 5 for s in spans)
    assert all(isinstance(d["text"], str) and 0.0 <= d["score"] <= 1.0 for s in spans for d in s)


# ---------------------------------------------------------------------------------------------------------------------
# (d) the reference driver's traces on the GPU (rows a1 / a7): same replay as tests/test_analyze_trace.py, but the canvases are
#     built on the device and the lines go through the real crop kernels and recogniser
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [0, 2])
def test_page_analyzer_replays_the_reference_trace_on_the_gpu(golden_dir, seed):
    import copy
    import json
    import zlib
    from rapiddoc_amd import analyze
    from rapiddoc_amd.pages import synth_page
    from rapiddoc_amd.pipeline import PagePipeline
    from test_analyze_trace import ReplayFormula, ReplayLayout
    fx = json.loads((golden_dir / f"analyze_trace_seed{seed}.json").read_text())
    tr = fx["trace"]
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]])).cuda()
    pipe = PagePipeline(_states(golden_dir, ("ppocrv6_det", "ppocrv6_rec")), rec_mode="strict", n_rec_streams=2)
    pipe.keep_rec_inputs = True
    log = {"layout": [], "formula": [], "det": []}
    det_calls = iter(tr["det_calls"])

    def det_raw_fn(canvases, batch_size):
        assert canvases.is_cuda
        call = next(det_calls)
        bgr = np.ascontiguousarray(canvases.cpu().numpy()[..., ::-1])
        log["det"].append({"batch_size": batch_size, "shapes": [list(c.shape) for c in bgr],
                           "crc32": [zlib.crc32(np.ascontiguousarray(c).tobytes()) for c in bgr]})
        return [np.asarray(b, dtype=np.float32).reshape(-1, 4, 2) for b in call["boxes"]]
    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], log["layout"]), pipe,
                              formula_model=ReplayFormula(log["formula"]) if fx["formula_enable"] else None,
                              layout_batch_size=fx["layout_batch_num"], formula_level=fx["formula_level"],
                              formula_batch_size=fx["formula_batch_num"], det_batch_num=fx["ocr_config"]["Det.rec_batch_num"],
                              det_raw_fn=det_raw_fn)
    out = pa(pages)
    assert log["formula"] == tr["formula_calls"]
    assert [{k: c[k] for k in ("batch_size", "shapes", "crc32")} for c in tr["det_calls"]] == log["det"]
    # the recogniser saw the reference's crops (sizes, pooled order) ...
    cw, ch, rot, keep = pipe.last_rec_crop_sizes
    shapes = [[int(cw[i]), int(ch[i])] if rot[i] else [int(ch[i]), int(cw[i])] for i in range(len(cw))]
    assert shapes == tr["rec_calls"][0]["shapes"]
    # ... in the reference's chunks of 6 (one global argsort over the pooled list)
    expected = _reference_rec_chunks([tuple(s) for s in shapes])
    want_w = {i: w for idxs, w in expected for i in idxs}
    got_w = {int(i): int(line_w[j]) for chunk, _x, line_w, _i, _p in pipe.last_rec_batches for j, i in enumerate(chunk.tolist())}
    assert got_w == want_w
    # ... and every page's output is the reference's, except what the (random-weight) recogniser read
    for mine, theirs in zip(out, fx["output"]):
        assert len(mine) == len(theirs)
        for a, b in zip(mine, theirs):
            skip = ("text", "score", "category_id") if b["category_id"] in (15, 16) else ()
            assert {k: v for k, v in a.items() if k not in skip} == {k: v for k, v in b.items() if k not in skip}
            if skip:
                assert a["category_id"] == (16 if a["score"] < 0.5 else 15) and isinstance(a["text"], str)


# ---------------------------------------------------------------------------------------------------------------------
# (e) fused stem front (stem1 + stem2a + stem2b + max-pool in one kernel) against the four separate kernels
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("kind,shape", [("ppocrv6_rec", (3, 3, 48, 1000)), ("ppocrv6_rec", (2, 3, 48, 17)), ("ppocrv6_det", (2, 3, 96, 160)),
                                        ("ppocrv6_det", (1, 3, 224, 352)), ("pphgnetv2_b4", (2, 3, 192, 320)),
                                        ("pphgnetv2_b6_formula", (2, 1, 96, 160))])
def test_fused_stem_equals_separate_kernels(golden_dir, kind, shape, monkeypatch):
    """Same engine, same weights, same input: plans built with the fused stem kernel vs plans built with stem_conv3x3s2 +
    conv_stream (2x2) x 2 + maxpool (RD_STEM_FUSED=0).  Ragged sizes: tiles cut by the right / bottom edge, odd widths, a map
    narrower than one tile, the 1-channel formula input.  Both forms split their operands (22 significant bits): the outputs of
    the whole network agree to fp32 round-off, and the per-op profile says which form ran."""
    from rapiddoc_amd.engine import RdEngine
    monkeypatch.setenv("RD_PRECISION", "auto")
    st = W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{kind}.json"), 0)
    x = torch.from_numpy(np.random.default_rng(sum(shape)).uniform(-1, 1, shape).astype(np.float32)).cuda()

    def run(fused):
        monkeypatch.setenv("RD_STEM_FUSED", "1" if fused else "0")
        eng = RdEngine(kind, guard="off").load_weights(st)
        eng.set_profiling(True)
        if kind == "ppocrv6_rec":
            from rapiddoc_amd.engine import REC_WANT_LOGITS
            out = [eng.rec_forward(x, REC_WANT_LOGITS)[2]]
        elif kind == "ppocrv6_det":
            out = [eng.det_forward(x)]
        elif kind == "pphgnetv2_b4":
            out = eng.backbone_forward(x)
        else:
            out = [eng.formula_encoder_forward(x)]
        kinds = [o["kind"] for o in eng.profile_log]
        assert not eng.range_overflow()
        return [o.clone() for o in out], kinds
    a, ka = run(True)
    b, kb = run(False)
    assert "stem_fused" in ka and "stem3x3s2" not in ka
    assert "stem_fused" not in kb and "stem3x3s2" in kb
    for u, v in zip(a, b):
        scale = max(1.0, float(v.abs().max()))
        assert float((u - v).abs().max()) < 2e-5 * scale


_DECODE_AB = r'''
import sys, numpy as np, torch
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
from pathlib import Path
from rapiddoc_amd.engine import RdEngine
from test_oracle_golden import formula_long_case
st, enc, g = formula_long_case(Path(sys.argv[1]) / "tests" / "golden")
eng = RdEngine("ppformulanet_head").load_weights(st)
e = torch.from_numpy(enc).cuda()
e8 = torch.cat([e, e.flip(0) * 0.5, e * 1.5, e.flip(1)], 0)          # B = 8: the batch the bench quotes
ids2 = eng.formula_decode(e, 300).cpu().numpy()
ids8 = eng.formula_decode(e8, 120).cpu().numpy()
np.savez(sys.argv[2], ids2=ids2, ids8=ids8)
'''


def test_formula_decode_round6_launches_are_bit_identical_to_round5(tmp_path):
    """Round 6 replaced the decode step's skinny GEMMs by weight-streaming GEMVs (dec_gemv_kernel), folded the attention launches' latency
    chain (dec_attn_fused2_kernel) and moved the next step's embedding into the select launch.  Each keeps the arithmetic of the launch
    it replaces (same lane -> k assignment, same product expression, same reduction trees), so the ids of a 300-token decode (B = 2) and
    of a 120-token decode at the bench's B = 8 must equal the round-5 launches' ids exactly (env switches are read once per process:
    two child processes)."""
    import os
    import subprocess
    import sys
    from pathlib import Path
    root = str(Path(__file__).resolve().parents[1])
    script = tmp_path / "ab.py"
    script.write_text(_DECODE_AB)
    outs = []
    for tag, extra in (("new", {}), ("old", {"RD_DEC_GEMV": "0", "RD_DEC_ATTN2": "0", "RD_DEC_EMBED_IN_SELECT": "0"})):
        out = tmp_path / f"{tag}.npz"
        env = dict(os.environ, **extra)
        r = subprocess.run([sys.executable, str(script), root, str(out)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    assert (outs[0]["ids2"] == outs[1]["ids2"]).all()
    assert (outs[0]["ids8"] == outs[1]["ids8"]).all()
    assert outs[0]["ids8"].shape[0] == 8 and outs[0]["ids8"].shape[1] > 20
