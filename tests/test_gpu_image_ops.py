"""GPU: the image kernels (rapiddoc_amd/csrc/kernels_image.hip) against the numpy restatement of OpenCV's 8-bit arithmetic
(oracle/cv2_ops.py): uint8 pixels must be IDENTICAL (compared through the normalised float outputs, which are exact
functions of the uint8 value), plus the layout wrapper plumbing of BASELINE.json configs[0]."""
import ctypes as C

import numpy as np
import pytest
import torch

from oracle import cv2_ops as CV

pytestmark = pytest.mark.gpu


def _to_u8(x, scale=255.0):
    return np.rint(x * scale).astype(np.int64)


@pytest.mark.parametrize("hw,out_hw", [((1684, 1191), (800, 800)), ((300, 200), (480, 480)), ((97, 131), (64, 64)), ((50, 70), (50, 70))])
def test_cubic_preprocess_is_pixel_exact(hw, out_hw):
    """PPPreProcess resize (cv2 INTER_CUBIC, pp_doclayout/pre_process.py:35) -> /255, mean 0 / std 1."""
    from rapiddoc_amd.engine import preproc_resize_norm
    img = np.random.default_rng(hw[0]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    img[: hw[0] // 3, : hw[1] // 2] = 255                                # a hard edge: overshoot / saturation
    got = preproc_resize_norm(torch.from_numpy(img).cuda(), out_hw, interp=2).cpu().numpy()
    ref = CV.resize_cubic_u8(img, out_hw).transpose(2, 0, 1)
    assert np.array_equal(_to_u8(got), ref.astype(np.int64))
    assert np.abs(got - CV.layout_preprocess(img, out_hw[0])[0]).max() < 1e-6 if out_hw[0] == out_hw[1] else True


@pytest.mark.parametrize("norm", ["rapidocr-default", "rapiddoc"])
@pytest.mark.parametrize("hw,out_hw", [((1784, 1291), (960, 704)), ((128, 448), (128, 448)), ((70, 333), (64, 320))])
def test_linear_preprocess_is_pixel_exact(hw, out_hw, norm):
    """rapidocr DetPreProcess resize (cv2 INTER_LINEAR) -> BGR, (x / 255 - mean) / std: with rapidocr's own default (0.5 / 0.5) and
    with the constants RapidDoc configures it with (Det.mean / Det.std, rapid_ocr.py:61-62 - what the product path uses)."""
    from rapiddoc_amd import ocr_host
    from rapiddoc_amd.engine import preproc_resize_norm
    mean, std = ((0.5, 0.5, 0.5), (0.5, 0.5, 0.5)) if norm == "rapidocr-default" else (ocr_host.DET_MEAN, ocr_host.DET_STD)
    img = np.random.default_rng(hw[1]).integers(0, 256, (*hw, 3), dtype=np.uint8)
    got = preproc_resize_norm(torch.from_numpy(img).cuda(), out_hw, mean=mean, std=std, interp=1, swap_rb=True).cpu().numpy()
    ref = CV.resize_linear_u8(img, out_hw)[:, :, ::-1].transpose(2, 0, 1)
    m, s_ = np.float32(mean).reshape(3, 1, 1), np.float32(std).reshape(3, 1, 1)
    assert np.array_equal(_to_u8(got * s_ + m), ref.astype(np.int64))
    # the reference normalises in float64 and rounds to float32 at the end (DetPreProcess.normalize)
    want = ((ref.astype("float32") * (1 / 255.0) - np.array(mean).reshape(3, 1, 1)) / np.array(std).reshape(3, 1, 1)).astype(np.float32)
    assert np.abs(got - want).max() < 2e-6


def test_line_crops_match_get_rotate_crop_image_and_resize_norm_img():
    """rd_line_crops_batch == get_rotate_crop_image (cubic warp, BORDER_REPLICATE, rot90 of tall crops, ocr_utils.py:494-536)
    followed by rapidocr resize_norm_img, on axis-aligned, tilted, border-crossing and tall quads."""
    from rapiddoc_amd import _lib
    from rapiddoc_amd.pipeline import LINE_DTYPE, quads_to_crop_matrices
    lib = _lib.load()
    rng = np.random.default_rng(9)
    pages = rng.integers(0, 256, (2, 400, 600, 3), dtype=np.uint8)
    quads = np.array([
        [[40, 50], [440, 50], [440, 82], [40, 82]],                      # axis-aligned line
        [[30, 120], [520, 150], [518, 182], [28, 152]],                  # tilted
        [[-6, 300], [200, 296], [201, 330], [-5, 334]],                  # crosses the left page border
        [[500, 20], [530, 20], [530, 200], [500, 200]],                  # tall: rotated by 90 degrees
        [[100, 350], [595, 352], [596.5, 391], [100.5, 389]],            # long
    ], np.float64)
    page_of = np.array([0, 1, 0, 1, 0])
    mats, cws, chs, _ok = quads_to_crop_matrices(quads)
    rot = (chs / cws >= 2.0).astype(np.int32)
    eff_w, eff_h = np.where(rot == 1, chs, cws), np.where(rot == 1, cws, chs)
    wpad = 608
    n = len(quads)
    d = np.zeros(n, dtype=LINE_DTYPE)
    d["page"], d["crop_w"], d["crop_h"], d["rot90"], d["m"] = page_of, cws.astype(np.int32), chs.astype(np.int32), rot, mats
    d["out_w"] = np.minimum(wpad, np.ceil(48 * eff_w / eff_h)).astype(np.int32)
    nbytes = (d["crop_w"].astype(np.int64) * d["crop_h"] * 3 + 15) // 16 * 16
    d["scratch_off"] = (np.cumsum(nbytes) - nbytes).astype(np.int32)
    dd = torch.from_numpy(d.view(np.uint8)).cuda()
    pg = torch.from_numpy(pages).cuda()
    scratch = torch.zeros(int(nbytes.sum()) + 64, dtype=torch.uint8, device="cuda")
    out = torch.full((n, 3, 48, wpad), 7.0, device="cuda")
    rc = lib.rd_line_crops_batch(0, pg.data_ptr(), 2, 400, 600, dd.data_ptr(), n, int((d["crop_w"].astype(np.int64) * d["crop_h"]).max()),
                                 scratch.data_ptr(), 48, wpad, 0, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    got = out.cpu().numpy()
    sc = scratch.cpu().numpy()
    for i in range(n):
        crop = CV.get_rotate_crop_image(pages[page_of[i]], quads[i].astype(np.float32))
        M, cw, ch = CV.perspective_dst_to_src(quads[i])
        raw = CV.warp_perspective_cubic_u8(pages[page_of[i]], M, (cw, ch))
        mine = sc[d["scratch_off"][i]: d["scratch_off"][i] + cw * ch * 3].reshape(ch, cw, 3)
        assert (cw, ch) == (d["crop_w"][i], d["crop_h"][i])
        assert np.array_equal(mine, raw), f"warp {i}: {np.abs(mine.astype(int) - raw.astype(int)).max()}"
        ref = CV.resize_norm_img(crop, wpad / 48.0)
        assert np.array_equal(got[i], ref), f"line {i}: {np.abs(got[i] - ref).max()}"


def test_layout_model_plumbing_config1():
    """BASELINE.json configs[0]: PP-DocLayout-S on 4 x 1684 x 1191 pages through the wrapper with a synthetic-box session:
    chunks of `batch_size`, [b,3,480,480] ImageNet-normalised inputs, scale_factor = [S/H, S/W], post-process, layout_dets
    schema (rapid_layout.py:55-108, pp_doclayout/main.py:38-80)."""
    import json
    from pathlib import Path
    from rapiddoc_amd.layout_host import LayoutPostProcess, to_layout_dets
    from rapiddoc_amd.layout_model import IMAGENET_MEAN, IMAGENET_STD, LayoutModel, SyntheticBoxSession, split_session_output
    from rapiddoc_amd.pages import synth_batch
    maps = json.loads((Path(__file__).resolve().parent / "golden" / "layout_category_maps.json").read_text())
    labels = list(maps["label_to_category"]["pp_doclayout"])
    sess = SyntheticBoxSession(labels, boxes_per_page=60, ncol=6, size=480)
    model = LayoutModel(sess, "pp_doclayout_s")
    pages_np, _ = synth_batch(0, 4)
    out = model.batch_predict([p for p in pages_np], batch_size=3)
    assert [c[0] for c in sess.calls] == [(3, 3, 480, 480), (1, 3, 480, 480)]
    assert np.allclose(sess.calls[0][1], [[480 / 1684, 480 / 1191]] * 3)
    assert len(out) == 4 and all(len(o) > 0 for o in out)
    for dets in out:
        for d in dets:
            assert set(d) == {"category_id", "original_label", "original_order", "poly", "polygon_points", "score"}
            assert d["original_order"] == -1 and d["polygon_points"] is None and d["score"] == round(d["score"], 3)
            x0, y0, x1, y1 = d["poly"][0], d["poly"][1], d["poly"][4], d["poly"][5]
            assert d["poly"] == [x0, y0, x1, y0, x1, y1, x0, y1] and 0 <= x0 < x1 <= 1191 and 0 <= y0 < y1 <= 1684
            want_cat = 2 if d["original_label"] in model.ignore else maps["label_to_category"]["pp_doclayout"][d["original_label"]]
            assert d["category_id"] in (want_cat, 13)
    # the wrapper is exactly: session boxes -> LayoutPostProcess(conf 0.2 for S) -> to_layout_dets
    x, sf = model.preprocess([pages_np[0]])
    ref_in = CV.layout_preprocess(pages_np[0], 480, IMAGENET_MEAN, IMAGENET_STD)
    assert np.abs(x.cpu().numpy() - ref_in).max() < 1e-5
    boxes = split_session_output(SyntheticBoxSession(labels, 60, 6, size=480)(x.cpu().numpy()[:1].repeat(3, 0), np.repeat(sf, 3, 0)))[0]["boxes"]
    want = to_layout_dets(LayoutPostProcess(labels, 0.2, 0.5)(boxes, [1191, 1684], None, "rect"), "pp_doclayout", False, model.ignore)
    got0 = [dict(d, category_id=d["category_id"]) for d in out[0]]
    assert [d["poly"] for d in got0] == [d["poly"] for d in want] and [d["original_label"] for d in got0] == [d["original_label"] for d in want]
    # V3: ordered output, 7-column boxes, 800 x 800, mean 0 / std 1, per-class merge table
    labels_v2 = list(maps["label_to_category"]["pp_doclayoutv2"])
    m3 = LayoutModel(SyntheticBoxSession(labels_v2, 50, 7), "pp_doclayoutv3")
    o3 = m3.batch_predict([pages_np[1]], 1)[0]
    assert m3.session.calls[0][0] == (1, 3, 800, 800) and [d["original_order"] for d in o3] == list(range(len(o3)))


@pytest.mark.parametrize("B,T", [(1, 2), (7, 40), (64, 136), (3, 400), (2, 700)])
def test_device_ctc_collapse_equals_host_decode(B, T):
    """rd_ctc_collapse (collapse repeats, drop blank, dictionary bytes, float32 np.mean of the kept probabilities) must give
    exactly what the host decode of the same (idx, prob) gives - strings AND confidences, bit for bit."""
    from rapiddoc_amd import _lib, ocr_host
    lib = _lib.load()
    rng = np.random.default_rng(B * 1000 + T)
    chars = ["blank"] + [chr(0x4E00 + i) for i in range(300)] + ["a", "é", "\U0001f600", " "]
    idx = rng.integers(0, len(chars), (B, T)).astype(np.int32)
    idx[rng.random((B, T)) < 0.45] = 0                            # blanks
    rep = rng.random((B, T)) < 0.3
    idx[:, 1:][rep[:, 1:]] = idx[:, :-1][rep[:, 1:]]              # repeats
    if B > 1:
        idx[1] = 0                                               # an empty line
    prob = rng.uniform(0.05, 1.0, (B, T)).astype(np.float32)
    tab, max_len = ocr_host.char_table(chars)
    row_bytes = (16 + T * max_len + 15) // 16 * 16
    out = torch.zeros((B, row_bytes), dtype=torch.uint8, device="cuda")
    ti, tp, tt = torch.from_numpy(idx).cuda(), torch.from_numpy(prob).cuda(), torch.from_numpy(tab).cuda()
    assert lib.rd_ctc_collapse(0, ti.data_ptr(), tp.data_ptr(), B, T, tt.data_ptr(), max_len, len(chars), out.data_ptr(), row_bytes,
                               torch.cuda.current_stream().cuda_stream) == 0
    got = ocr_host.parse_ctc_rows(out.cpu().numpy())
    ref = ocr_host.ctc_decode(idx, prob, chars)
    assert [g[0] for g in got] == [r[0] for r in ref]
    assert [np.float32(g[1]).tobytes() for g in got] == [np.float32(r[1]).tobytes() for r in ref]


def test_page_analyzer_runs_the_reference_stage_order(golden_dir):
    """BatchAnalyze.__call__ sequencing (batch_analyze.py:78-164) on 2 synthetic pages: layout (synthetic-box session through
    the real wrapper) -> overlap filter -> formulas -> OCR det/rec spans -> table seam; pages independent; schema kept."""
    import json
    from rapiddoc_amd import weights as W
    from rapiddoc_amd.analyze import LOW_SCORE_TEXT, OCR_TEXT, PageAnalyzer
    from rapiddoc_amd.layout_model import LayoutModel
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline
    maps = json.loads((golden_dir / "layout_category_maps.json").read_text())
    labels = list(maps["label_to_category"]["pp_doclayoutv2"])

    class FixedSession:          # page-pixel boxes: two text regions, one table, one display formula (7-column V3 rows)
        characters = labels
        accepts_device_tensors = False
        calls = 0

        def __call__(self, x, sf):
            FixedSession.calls += 1
            rows = []
            for b in range(x.shape[0]):
                def row(label, score, x0, y0, x1, y1, order):
                    return [labels.index(label), score, x0, y0, x1, y1, order]
                rows += [row("text", 0.9, 80, 50, 1150, 500, 0), row("text", 0.8, 80, 520, 640, 980, 1),
                         row("table", 0.9, 650, 1000, 1150, 1400, 2), row("display_formula" if "display_formula" in labels else "formula", 0.9, 100, 1450, 600, 1520, 3)]
            return [np.asarray(rows, np.float32), np.full(x.shape[0], 4, np.int32)]

    class Formula:
        def batch_predict(self, imgs, batch_size=16):
            return ["x^{%d}" % im.shape[1] for im in imgs]

    class Table:
        def batch_predict(self, imgs, **kw):
            return ["<table><tr><td>%dx%d</td></tr></table>" % im.shape[:2] for im in imgs]

    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, rec_batch_num=32, n_rec_streams=2)
    an = PageAnalyzer(LayoutModel(FixedSession(), "pp_doclayoutv3"), pipe, formula_model=Formula(), table_model=Table(), layout_batch_size=2)
    pages_np, boxes = synth_batch(0, 2)

    def maps_fn(regions, ghw, dhw):
        (gh, gw), (dh, dw) = ghw, dhw
        m = torch.zeros((len(regions), 1, dh, dw), dtype=torch.float32)
        for k, (p, r, useful) in enumerate(regions):
            px, py, x0, y0 = useful[:4]
            for lb in np.asarray(boxes[p], dtype=np.float64).reshape(-1, 4):
                if lb[0] >= r["poly"][0] and lb[2] <= r["poly"][4] and lb[1] >= r["poly"][1] and lb[3] <= r["poly"][5]:
                    cx0, cy0, cx1, cy1 = lb[0] - x0 + px, lb[1] - y0 + py, lb[2] - x0 + px, lb[3] - y0 + py
                    d = 0.32 * min(cx1 - cx0, cy1 - cy0)
                    m[k, 0, int(round((cy0 + d) * dh / gh)):int(round((cy1 - d) * dh / gh)), int(round((cx0 + d) * dw / gw)):int(round((cx1 - d) * dw / gw))] = 0.95
        return m.cuda()

    out = an(torch.from_numpy(pages_np).cuda(), det_maps_fn=maps_fn)
    assert FixedSession.calls == 1 and len(out) == 2
    for p in range(2):
        layout = [d for d in out[p] if d["category_id"] not in (OCR_TEXT, LOW_SCORE_TEXT)]
        spans = [d for d in out[p] if d["category_id"] in (OCR_TEXT, LOW_SCORE_TEXT)]
        assert [d["original_label"] for d in layout][:2] == ["text", "text"] and [d["original_order"] for d in layout] == list(range(len(layout)))
        table = [d for d in layout if d["category_id"] == 5]
        formula = [d for d in layout if d["category_id"] in (8, 14)]
        assert len(table) == 1 and table[0]["html"] == "<table><tr><td>400x500</td></tr></table>"
        assert len(formula) == 1 and formula[0]["latex"].startswith("x^{")
        lines = np.asarray(boxes[p]).reshape(-1, 4)
        inside = sum(1 for lb in lines if (lb[1] >= 50 and lb[3] <= 500 and lb[2] <= 1150) or (lb[1] >= 520 and lb[3] <= 980 and lb[2] <= 640))
        assert len(spans) == inside and inside > 10
        assert all(isinstance(s["text"], str) and s["score"] == float(f"{s['score']:.3f}") for s in spans)


def _db_test_map(kind):
    """The DB probability maps of test_device_db_postprocess_equals_host_path (also read by tests/test_db_postprocess.py on the CPU)."""
    rng = np.random.default_rng(5)
    B, H, W = 3, 320, 448
    m = np.full((B, H, W), 0.02, np.float32)
    box_thresh = 0.3
    if kind == "lines":
        for b in range(B):
            y = 10
            while y + 30 < H:
                w = int(rng.integers(60, W - 20))
                m[b, y + 6: y + 20, 10: 10 + w] = rng.uniform(0.5, 0.99, (14, w))
                y += int(rng.integers(30, 44))
    elif kind == "blobs":
        f = rng.standard_normal((B, H // 8 + 2, W // 8 + 2)).astype(np.float32)
        f = np.kron(f, np.ones((8, 8), np.float32))[:, :H, :W]
        f = (f + np.roll(f, 3, 1) + np.roll(f, 5, 2)) / 3
        m = (1 / (1 + np.exp(-3 * f))).astype(np.float32)
        for k in range(40):                                   # diagonal staircases: 8- but not 4-connected pixels
            m[0, 100 + k, 200 + k] = 0.9
        m[1, 50:90, 300:380] = 0.9
        m[1, 60:80, 320:360] = 0.01                            # a hole
    elif kind == "holes":
        box_thresh = 0.02
        m[0, 20:120, 30:300] = 0.9
        m[0, 40:100, 60:260] = 0.01                            # a ring ...
        m[0, 55:85, 100:220] = 0.9                             # ... an island in its hole ...
        m[0, 65:75, 120:160] = 0.01                            # ... with two holes of its own
        m[0, 62:70, 180:200] = 0.01
        m[0, 150:250, 0:200] = 0.9
        m[0, 170:200, 0:50] = 0.01                             # reaches the left frame: not a hole
        m[0, 210:230, 80:120] = 0.01
        m[0, 260:319, 250:447] = 0.9                           # region in the bottom right corner
        m[0, 300:320, 300:330] = 0.01                          # reaches the bottom frame
        m[0, 280:290, 400:448] = 0.01                          # reaches the right frame
        m[0, 270:296, 260:290] = 0.01
        m[1, 30:200, 30:400] = 0.9
        for k in range(12):                                    # a diagonal chain of holes that touch at their corners (after the
            m[1, 40 + 4 * k: 45 + 4 * k, 50 + 4 * k: 55 + 4 * k] = 0.01     # dilation too): each is its own 4-connected hole
        m[1, 150:180, 100:103] = 0.01                          # thin holes: two columns / two rows survive the dilation's bite
        m[1, 150:153, 200:300] = 0.01
        m[1, 100:140, 300:340] = 0.01                          # an L-shaped hole
        m[1, 120:140, 340:380] = 0.01
        m[2, 10:310, 10:440] = 0.9                             # page 2: a region with hundreds of ragged holes
        for _ in range(400):
            y, x, h, w = int(rng.integers(12, 300)), int(rng.integers(12, 430)), int(rng.integers(3, 9)), int(rng.integers(3, 9))
            m[2, y:y + h, x:x + w] = 0.01
    elif kind == "full":
        m[:] = 0.8
    elif kind == "speckle":
        dots = rng.random((B, H, W)) < 0.02                    # ~2800 isolated specks per page: more regions than max_candidates
        dots[:, 250:, :] = False
        m[dots] = 0.9
        m[:, 280:300, 40:400] = 0.85                           # a text line BEHIND the first 1000 regions of page 0 / 1 ...
        m[2, :250] = 0.02                                      # ... and in front of them on page 2
        m[2, 20:40, 40:400] = 0.85
    return m, box_thresh, B


@pytest.mark.parametrize("kind", ["lines", "blobs", "holes", "empty", "full", "speckle"])
def test_device_db_postprocess_equals_host_path(kind):
    """rd_db_boxes_device (everything on the GPU: raster-ordered runs, union-find regions, hulls of the row extremes, min-area
    rectangles, scores, unclip, filter) == rd_db_postprocess (flood fill on the host), box for box and in the same order:
    text-line maps, irregular blobs (touching the borders, diagonal 8-connections, holes), a map built around hole borders
    (cv2.findContours RETR_LIST returns them as contours: rings, a hole with an island with a hole, one-pixel holes, holes that
    touch only diagonally = two holes, notches that reach the image frame = no hole; scored with a box_thresh low enough to keep
    them), an empty and a full map, and a speckled map with more than max_candidates regions (only the first 1000 contours in
    raster order count)."""
    from rapiddoc_amd import ocr_host
    m, box_thresh, B = _db_test_map(kind)
    hw = [(640, 896)] * B
    host = ocr_host.db_postprocess(m, hw, thresh=0.3, box_thresh=box_thresh, unclip_ratio=1.8)
    dev = ocr_host.db_postprocess_device(torch.from_numpy(m).cuda(), hw, thresh=0.3, box_thresh=box_thresh, unclip_ratio=1.8)
    assert len(host) == len(dev) == B
    for (hb, hs), (db, ds) in zip(host, dev):
        assert hb.shape == db.shape and np.array_equal(hb, db)
        assert np.allclose(hs, ds, rtol=0, atol=1e-6)
    # ... and against the independent restatement (oracle/dbpost.py: scipy labelling / hulls, float64), every kind:
    # the same boxes in the same order.  Corner coordinates are equal except where the two arithmetic classes part: the C++ / device
    # chain carries the minimum-area rectangle in float32, the oracle in float64, so (a) a scaled corner within float32 rounding of
    # k + 0.5 rounds to the other integer (1 px) and (b) when two edge orientations of the truncated box give minimum-area rectangles of
    # equal area to float32 precision, either may be picked (<= 3 px).  Measured on these maps: 12 of 5992 coordinates.
    from oracle import dbpost as OD
    if kind == "holes":
        assert len(host[0][0]) == 9 and len(host[1][0]) == 1 + 12 + 3 and len(host[2][0]) > 100
    n_coord = n_diff = 0
    for b in range(B):
        ob, osc = OD.db_postprocess(m[b], hw[b], thresh=0.3, box_thresh=box_thresh, unclip_ratio=1.8)
        assert len(ob) == len(dev[b][0]), (kind, b)
        for x, y, so, sd in zip(ob, dev[b][0], osc, dev[b][1]):
            d = np.abs(np.asarray(x) - y)
            assert d.max() <= 3 and abs(so - sd) < 1e-5
            n_coord += d.size
            n_diff += int((d > 0).sum())
    assert n_diff <= max(2, 0.005 * n_coord), (kind, n_diff, n_coord)
    if kind in ("lines", "full", "speckle"):
        assert n_diff == 0
    if kind == "speckle":
        assert [len(b) for b, _ in dev] == [0, 0, 1] or [len(b) for b, _ in dev][2] >= 1
    if kind == "lines":
        assert all(len(b) >= 5 for b, _ in dev)
        # the benchmark's size: 960 x 704 maps rendered from the page generator's line boxes, 8 pages
        from rapiddoc_amd.pages import synth_batch
        from rapiddoc_amd.pipeline import render_text_maps
        pages_np, boxes = synth_batch(40, 8)
        maps = render_text_maps(boxes, pages_np.shape[1:3], (960, 704), "cuda")
        hw2 = [tuple(pages_np.shape[1:3])] * 8
        h2 = ocr_host.db_postprocess(maps.cpu().numpy(), hw2, thresh=0.3, box_thresh=0.3, unclip_ratio=1.8)
        d2 = ocr_host.db_postprocess_device(maps, hw2, thresh=0.3, box_thresh=0.3, unclip_ratio=1.8)
        for (hb, hs), (db, ds) in zip(h2, d2):
            assert len(hb) == 45 and np.array_equal(hb, db) and np.allclose(hs, ds, rtol=0, atol=1e-6)
    if kind == "empty":
        assert all(len(b) == 0 for b, _ in dev)
    # overflow of the run buffer falls back to the host path with identical results
    small = ocr_host.db_postprocess_device(torch.from_numpy(m).cuda(), hw, thresh=0.3, box_thresh=box_thresh, unclip_ratio=1.8, max_runs=4)
    for (hb, _), (sb, _) in zip(host, small):
        assert np.array_equal(hb, sb)


def test_full_pipeline_composition_config3(golden_dir, tmp_path):
    """BASELINE.json configs[2] as far as it can be composed offline: layout wrapper -> OCR det / rec -> PP-FormulaNet_plus-M
    (the real B6 encoder + MBart decoder engines with synthetic weights, strings through a BPE tokenizer JSON + LaTeX fix-ups)
    -> table seam (SLANet_plus is ONNX-only in the reference).  Checks that every stage ran on the GPU engines and that the
    formula string equals the decode of the oracle's greedy token ids for that crop."""
    import json
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from oracle import formula as OF
    from oracle import nets as O
    from rapiddoc_amd import formula_host as FH
    from rapiddoc_amd import weights as W
    from rapiddoc_amd.analyze import PageAnalyzer
    from rapiddoc_amd.layout_model import LayoutModel
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline
    # a byte-level BPE tokenizer whose vocabulary covers every id the 50 000-way head can emit is not needed: ids the tokenizer
    # does not know decode to nothing, exactly like `tokenizers` does for the reference
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for i in range(4, 300):
        vocab["t%d" % i] = i
    tok = Tokenizer(models.BPE(vocab=vocab, merges=[], unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.add_special_tokens(["<s>", "<pad>", "</s>", "<unk>"])
    tj = tmp_path / "tok.json"
    tok.save(str(tj))
    st_f = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppformulanet_plus_m_m8.json"), 0)
    formula = FH.FormulaRecognizer(st_f, max_new_tokens=6, tokenizer_json=str(tj), fix_text=None)
    maps = json.loads((golden_dir / "layout_category_maps.json").read_text())
    labels = list(maps["label_to_category"]["pp_doclayoutv2"])

    class Session:
        characters = labels

        def __call__(self, x, sf):
            rows = [[labels.index("text"), 0.9, 80, 50, 1150, 300, 0], [labels.index("display_formula"), 0.9, 100, 1000, 600, 1100, 1],
                    [labels.index("table"), 0.9, 650, 1000, 1150, 1400, 2]] * x.shape[0]
            return [np.asarray(rows, np.float32), np.full(x.shape[0], 3, np.int32)]

    class Table:
        def batch_predict(self, imgs, **kw):
            return ["<table></table>" for _ in imgs]

    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, rec_batch_num=16, n_rec_streams=2)
    an = PageAnalyzer(LayoutModel(Session(), "pp_doclayoutv3"), pipe, formula_model=formula, table_model=Table())
    pages_np, _ = synth_batch(3, 1)
    pages_np[0, 1000:1100, 100:600] = 255
    pages_np[0, 1030:1070, 150:550] = 20                       # a dark bar as the "formula"
    out = an(torch.from_numpy(pages_np).cuda())[0]
    f = [d for d in out if d["category_id"] == 14]
    assert len(f) == 1 and isinstance(f[0].get("latex", ""), str)
    t = [d for d in out if d["category_id"] == 5]
    assert len(t) == 1 and t[0]["html"] == "<table></table>"
    # the formula string is the decode of the oracle's greedy ids on the same crop tensor
    crop = pages_np[0, 1000 - 0:1100 + 0, 100:600]
    from rapiddoc_amd import layout_host
    c = layout_host.expand_formula_crop(f[0], out, pages_np.shape[1:3], 2)
    x0, y0, x1, y1 = int(c["poly"][0]), int(c["poly"][1]), int(c["poly"][4]), int(c["poly"][5])
    xin = FH.preprocess([pages_np[0, y0:y1, x0:x1]])[0]
    tst = O.as_torch_state(st_f)
    ids, lgs = OF.formula_decode(tst, O.formula_encoder_forward(tst, torch.from_numpy(xin)), 6, return_logits=True)
    gaps = torch.stack([torch.topk(l, 2, dim=-1).values for l in lgs], 1)
    if float((gaps[..., 0] - gaps[..., 1]).min()) > 1e-2:
        toks = ids[0, 1:].tolist()
        toks = toks[: toks.index(2)] if 2 in toks else toks
        assert f[0].get("latex", "") == FH.make_token_decoder(str(tj), None)(toks)
