"""CPU: `PageAnalyzer` / `RegionOcr` replay traces of the REFERENCE's own page-batch driver.

tests/golden/analyze_trace_seed*.json were recorded by running rapid_doc's `BatchAnalyze.__call__`
(backend/pipeline/batch_analyze.py:78-164 -> analyze_utils.py:105-292 -> utils/ocr_utils.py:361-431) with recording stand-ins
for the three models (tests/golden/make_golden_analyze.py).  Here the same stand-ins are plugged into this repo's driver through
its model seams and everything the reference did is demanded back:

  * the call sequence: one layout call (batch size, image shapes), one formula call with every formula crop of the batch
    (shapes = `_expand_formula_crop_res` + `crop_img`), one detector call per (language, 64-px size bucket) in first-appearance
    order with the reference's batch size, ONE recogniser call with every text line of the page batch pooled page by page;
  * the detector's input canvases BYTE FOR BYTE (crc32 of the BGR image: 50-px white margin, 255 padding to the bucket, formula
    boxes whited out of the det copy only);
  * the formula crops BYTE FOR BYTE (crc32 of the RGB crop);
  * seed 3: every layout box carries `polygon_points` - region and formula crops are whited out outside their polygon (crop_img,
    utils/model_utils.py:109-118) and two nearly coincident text boxes survive the overlap filter because their POLYGONS barely
    overlap (backend/utils/utils.py:150-155); cv2.fillPoly / shapely are this repo's primitives on both sides (make_golden_polygon.py);
  * the recogniser's crop SIZES in pooled order (`get_rotate_crop_image`'s float32 edge norms, truncation, the h/w >= 2 rotation);
  * the output `layout_dets` of every page, dict for dict: pass-through layout boxes (+ the 'bbox' / 'latex' fields written
    into formulas), overlap filter, inline-formula drop, spans sorted / merged / cut around formulas / tilt-corrected / mapped
    to page coordinates, scores formatted to 3 decimals, LowScoreText demotion.
No GPU: the networks are exactly what the stand-ins replace."""
import copy
import json
import zlib

import numpy as np
import pytest
import torch

from rapiddoc_amd import analyze, ocr_host
from rapiddoc_amd.pages import synth_page
from rapiddoc_amd.pipeline import quads_to_crop_matrices


def rec_text_and_score(h, w, k):        # == make_golden_analyze.rec_text_and_score
    score = ((h * 131 + w * 17 + k * 29) % 1000) / 1000.0
    return f"T{k}:{h}x{w}", score


class ReplayLayout:
    def __init__(self, dets, log):
        self.dets, self.log = dets, log

    def batch_predict(self, images, batch_size):
        self.log.append({"batch_size": int(batch_size), "shapes": [list(i.shape) for i in images]})
        return copy.deepcopy(self.dets)


class ReplayFormula:
    def __init__(self, log):
        self.log = log

    def batch_predict(self, images, batch_size=1, **kw):
        self.log.append({"batch_size": int(batch_size), "shapes": [list(np.asarray(i).shape[:2]) for i in images],
                         "crc32": [zlib.crc32(np.ascontiguousarray(np.asarray(i)).tobytes()) for i in images]})
        return [f"\\\\frac{{{np.asarray(i).shape[0]}}}{{{np.asarray(i).shape[1]}}}" for i in images]


class ReplayPipe:
    """Stands where a PagePipeline stands: `rec_forward_sources` is the recogniser seam (`ocr_model.ocr(crops, det=False)` in
    the reference, analyze_utils.py:252)."""

    def __init__(self, log):
        self.log = log

    def rec_forward_sources(self, sources, image_keys=None):
        # the pooled line list exactly as PagePipeline builds it: (source, image, line), then a stable sort by the image key
        quads, owner = [], []
        for si, (_imgs, per_img) in enumerate(sources):
            for pi, q in enumerate(per_img):
                q = np.asarray(q, dtype=np.float64).reshape(-1, 4, 2)
                for j in range(len(q)):
                    quads.append(q[j])
                    owner.append((si, pi))
        keys = np.array([image_keys[si][pi] for si, pi in owner], dtype=np.int64)
        perm = np.argsort(keys, kind="stable")
        _m, cw, ch, ok = quads_to_crop_matrices(np.asarray(quads)[perm])
        assert ok.all()
        shapes = []
        for w_, h_ in zip(cw.astype(int).tolist(), ch.astype(int).tolist()):
            shapes.append([w_, h_] if h_ / w_ >= 2 else [h_, w_])          # np.rot90 of tall crops (ocr_utils.py:531-535)
        self.log.append({"shapes": shapes})
        out = [[[] for _ in per_img] for _imgs, per_img in sources]
        res = [None] * len(quads)
        for k, src_index in enumerate(perm.tolist()):
            t, s = rec_text_and_score(shapes[k][0], shapes[k][1], k)
            res[src_index] = (t, ocr_host.format_score(s))
        for (si, pi), r in zip(owner, res):
            out[si][pi].append(r)
        return out


@pytest.mark.parametrize("seed", [0, 1, 2, 3, 5, 6, 7, 8, 9])
def test_page_analyzer_replays_the_reference_trace(golden_dir, seed):
    """Seeds 0-3: traces cut with ocr_config["seal_enable"] = False.  5 and 6: the reference's DEFAULT (seal OCR on, batch_analyze.py:62) -
    5 = seed 0's pages (no seal region: same output), 6 = three seal regions, one with a polygon, one nothing is read in: the crops
    handed to the seal OCR model byte for byte, the `text` lists written into the regions.  7: checkbox detection on (batch_analyze.py:
    207-219) - the BGR pages handed to checkbox_predict, its hits as CheckBox detections in front of the OCR spans and whited out of the
    det canvases next to the formulas.  8 / 9: USE_DOC_ORIENTATION_CLASSIFY (batch_analyze.py:113-125,153-161) - the pages handed to
    the orientation model; 8: both pages sideways ("90", "270") -> turned upright before the layout model (its logged input shapes),
    every `poly` mapped back, `rotate_label` on every detection; 9: "0" and "180" -> pages untouched, labels only."""
    fx = json.loads((golden_dir / f"analyze_trace_seed{seed}.json").read_text())
    tr = fx["trace"]
    seal_on = fx["ocr_config"].get("seal_enable", True)
    assert seal_on == (seed >= 5)
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]]))      # CPU tensor: the networks are stood in for
    assert list(pages.shape[1:3]) == fx["page_hw"]
    if fx.get("input_rot90"):          # seed 8: the pages come in lying on their side
        pages = torch.from_numpy(np.stack([np.ascontiguousarray(np.rot90(p_, fx["input_rot90"])) for p_ in pages.numpy()]))
    log = {"layout": [], "formula": [], "det": [], "rec": []}
    det_calls = iter(tr["det_calls"])

    def det_raw_fn(canvases, batch_size):
        call = next(det_calls)
        bgr = np.ascontiguousarray(canvases.numpy()[..., ::-1])
        log["det"].append({"batch_size": batch_size, "shapes": [list(c.shape) for c in bgr],
                           "crc32": [zlib.crc32(np.ascontiguousarray(c).tobytes()) for c in bgr]})
        return [np.asarray(b, dtype=np.float32).reshape(-1, 4, 2) for b in call["boxes"]]

    class ReplaySeal:                 # == make_golden_analyze.SealOcr
        def ocr(self, img, det=True, rec=True, **kw):
            assert det is True and rec is True and not kw
            k = len(log.setdefault("seal", []))
            log["seal"].append({"shape": list(img.shape), "crc32": zlib.crc32(np.ascontiguousarray(img).tobytes())})
            if k == 2:
                return [None]
            box = [[1.0, 2.0], [30.0, 2.0], [30.0, 12.0], [1.0, 12.0]]
            return [[[box, (f"seal {k} line a {img.shape[0]}x{img.shape[1]}", 0.91)], None, [box], [box, ()], [box, ("", 0.4)], [box, ("line b", 0.2)]]]

    class ReplayOrientation:          # == make_golden_analyze.OrientationCls
        def predict(self, img):
            k = len(log.setdefault("orientation", []))
            log["orientation"].append({"shape": list(img.shape), "crc32": zlib.crc32(np.ascontiguousarray(img).tobytes())})
            return fx["rotate_labels"][k]

    def checkbox_fn(bgr):             # == make_golden_analyze.checkbox_predict
        k = len(log.setdefault("checkbox", []))
        log["checkbox"].append({"shape": list(bgr.shape), "crc32": zlib.crc32(np.ascontiguousarray(bgr).tobytes())})
        first = next(d for d in fx["layout_dets"][k] if d["category_id"] in (0, 1, 2, 4, 6, 7, 9))
        x0, y0 = int(first["poly"][0]) + 12, int(first["poly"][1]) + 6
        return [{"bbox": [x0, y0, x0 + 14, y0 + 14], "text": "checked" if k == 0 else "unchecked", "score": 0.97},
                {"bbox": [3, 3, 15, 15], "text": "unchecked", "score": 0.5}]

    formula_model = ReplayFormula(log["formula"]) if fx["formula_enable"] else None
    extra = {}
    if fx.get("checkbox_enable"):
        extra.update(checkbox_fn=checkbox_fn, checkbox_enable=True)
    if fx.get("rotate_labels"):
        extra.update(orientation_model=ReplayOrientation(), use_doc_orientation_classify=True)
    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], log["layout"]), ReplayPipe(log["rec"]), formula_model=formula_model,
                              layout_batch_size=fx["layout_batch_num"], formula_level=fx["formula_level"],
                              formula_batch_size=fx["formula_batch_num"], det_batch_num=fx["ocr_config"]["Det.rec_batch_num"],
                              det_raw_fn=det_raw_fn, seal_model=ReplaySeal() if seed == 6 else None, seal_enable=seal_on, **extra)
    out = pa(pages)
    assert log.get("seal", []) == tr.get("seal_calls", [])
    assert log.get("checkbox", []) == tr.get("checkbox_calls", []) and log.get("orientation", []) == tr.get("orientation_calls", [])
    if seed == 7:
        assert len(log["checkbox"]) == 2 and sum(1 for page in out for d in page if d["category_id"] == 200) == 4
    if seed in (8, 9):
        assert pa.last_rotate_labels == fx["rotate_labels"] and all(d["rotate_label"] == fx["rotate_labels"][p] for p, page in enumerate(out) for d in page)
    if seed == 6:
        assert len(log["seal"]) == 3 and sum(1 for page in out for d in page if d.get("original_label") == "seal" and "text" in d) == 2
    if seed == 5:
        assert out == json.loads((golden_dir / "analyze_trace_seed0.json").read_text())["output"]

    # ---- calls
    assert log["layout"] == tr["layout_calls"]
    assert log["formula"] == tr["formula_calls"]
    assert [{k: c[k] for k in ("batch_size", "shapes", "crc32")} for c in tr["det_calls"]] == log["det"]
    assert next(det_calls, None) is None
    assert log["rec"] == tr["rec_calls"] and len(log["rec"]) == 1
    # ---- output, dict for dict
    ref = fx["output"]
    assert len(out) == len(ref)
    for p, (mine, theirs) in enumerate(zip(out, ref)):
        assert len(mine) == len(theirs), (p, len(mine), len(theirs))
        for a, b in zip(mine, theirs):
            assert a == b and list(a) == list(b), (p, a, b)          # same fields in the same order


# ---------------------------------------------------------------------------------------------------------------------
# the table stage (SURVEY row a17's host side): traces of BatchAnalyze with table_enable, tests/golden/make_golden_table_trace.py
# ---------------------------------------------------------------------------------------------------------------------
def fill_summary(fill_image_res):        # == make_golden_table_trace.fill_summary
    out = []
    for f in fill_image_res:
        d = {k: json.loads(json.dumps(v)) for k, v in f.items() if k not in ("uuid", "pil_image")}
        d["keys"] = list(f)
        d["pil_size"] = list(f["pil_image"].size)
        d["pil_crc32"] = zlib.crc32(np.ascontiguousarray(np.asarray(f["pil_image"])).tobytes())
        out.append(d)
    return out


def table_text_and_score(h, w):          # == make_golden_table_trace.table_text_and_score
    return ("香" if (h + w) % 7 == 0 else f"<{h}x{w}>"), ((h * 37 + w * 11) % 1000) / 1000.0


def table_words_for(h, w, dt_box):       # == make_golden_table_trace.table_words_for
    q = np.asarray(dt_box, dtype=np.float64).reshape(4, 2)
    words = []
    for j in range((h + w) % 4):
        if (h * 3 + w + j) % 11 == 0:
            words.append((f"gone{j}", 0.5, None))
            continue
        a, b = j / 4.0, (j + 1) / 4.0
        tl, tr, br, bl = q[0] + a * (q[1] - q[0]), q[0] + b * (q[1] - q[0]), q[3] + b * (q[2] - q[3]), q[3] + a * (q[2] - q[3])
        box = [[float(tl[0]) - 0.6 - (700.0 if (h + j) % 13 == 0 else 0.0), float(tl[1]) + 0.3], [float(tr[0]) + 0.6, float(tr[1]) + 0.3],
               [float(br[0]) + 0.6, float(br[1]) + 0.7 + (900.0 if (w + j) % 17 == 0 else 0.0)], [float(bl[0]) - 0.6, float(bl[1]) + 0.7]]
        words.append((("香<" if (h + w + j) % 5 == 0 else "w") + f"{j}:{h}x{w}", round(((h * 7 + w * 3 + j) % 1000) / 1000.0, 5), box))
    return words


@pytest.mark.parametrize("kind", ["traditional", "custom", "traditional_words", "traditional_checkbox"])
def test_table_stage_replays_the_reference_trace(golden_dir, kind):
    """`traditional`: a `predict`-shaped table model - per table the reference crops (box snapped outwards to multiples of 5 px), whites the
    page's formulas out of the detector's copy, detects (0.5 / 1.6, sorted, cut around formulas, NOT merged), recognises every line
    from the unmasked crop, normalises / HTML-escapes the texts and calls predict(image, [boxes, texts, scores], [], formula boxes
    with latex, True, False, skip_table_orientation=True); the answer's <table> part and the page's formula boxes / scale land on the
    region; a picture lying inside a table (>= 0.8 of its area) reaches the model as `fill_image_res` (page box, corners in crop
    coordinates, the PIL crop) and its box / scale lands in `img_boxes`.  Demanded back: the detector canvases and the table images
    byte for byte, every argument of every predict call, the output dicts.  `custom`: ONE batch_predict over the tables of all pages with the same crops and fill_image_res_list.
    `traditional_checkbox`: checkbox detection on - a hit inside a table is whited out of the table detector's copy and handed to the
    table model next to the formulas (analyze_utils.py:318-321), every hit of the page is listed in `formula_boxes` (:411-418)."""
    fx = json.loads((golden_dir / f"analyze_trace_table_{kind}.json").read_text())
    tr = fx["trace"]
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]]))
    log = {"layout": [], "formula": [], "det": [], "rec": [], "table_det": [], "table": []}
    det_calls, tdet_calls = iter(tr["det_calls"]), iter(tr["table_det_calls"])

    def det_raw_fn(canvases, batch_size):
        call = next(det_calls)
        bgr = np.ascontiguousarray(canvases.numpy()[..., ::-1])
        log["det"].append({"batch_size": batch_size, "shapes": [list(c.shape) for c in bgr],
                           "crc32": [zlib.crc32(np.ascontiguousarray(c).tobytes()) for c in bgr]})
        return [np.asarray(b, dtype=np.float32).reshape(-1, 4, 2) for b in call["boxes"]]

    def table_det_raw_fn(canvas, batch_size):
        call = next(tdet_calls)
        bgr = np.ascontiguousarray(canvas.numpy()[0][..., ::-1])
        log["table_det"].append({"shape": list(bgr.shape), "crc32": zlib.crc32(bgr.tobytes())})
        return [np.asarray(call["boxes"], dtype=np.float32).reshape(-1, 4, 2)]

    word_box = fx["table_config"].get("use_word_box", True)      # the reference's default: True (analyze_utils.py:308)
    assert word_box == (kind == "traditional_words")

    def table_rec_fn(canvas, quads):
        _m, cw, ch, ok = quads_to_crop_matrices(np.asarray(quads, dtype=np.float64))
        assert ok.all()
        out = []
        for q, w_, h_ in zip(quads, cw.astype(int).tolist(), ch.astype(int).tolist()):
            hh, ww = (w_, h_) if h_ / w_ >= 2 else (h_, w_)
            if word_box:        # + what rapidocr's cal_rec_boxes leaves per line (the generator's stand-in: a function of crop shape and box)
                out.append(table_text_and_score(hh, ww) + (table_words_for(hh, ww, q),))
            else:
                out.append(table_text_and_score(hh, ww))
        if word_box:
            log.setdefault("table_words", []).append({"shapes": [[int(ch_), int(cw_), 3] if not (ch_ / cw_ >= 2) else [int(cw_), int(ch_), 3]
                                                                 for cw_, ch_ in zip(cw.astype(int).tolist(), ch.astype(int).tolist())],
                                                      "dt_boxes": [np.asarray(q, dtype=np.float32).tolist() for q in quads]})
        return out

    class ReplayTable:
        def predict(self, image, ocr_result, fill_image_res, mfd_res, skip_text_in_image, use_img2table, skip_table_orientation=False):
            boxes, texts, scores = ocr_result if ocr_result else ([], [], [])
            log["table"].append({"shape": list(image.shape), "crc32": zlib.crc32(np.ascontiguousarray(image).tobytes()),
                                 "boxes": [np.asarray(b, dtype=np.float64).tolist() for b in boxes], "texts": list(texts),
                                 "scores": [float(s) for s in scores], "fill_image_res": fill_summary(fill_image_res), "mfd_res": mfd_res,
                                 "flags": [bool(skip_text_in_image), bool(use_img2table), bool(skip_table_orientation)]})
            if len(texts) % 2 == 0 and kind != "traditional_checkbox":
                return "<html><body>nothing found</body></html>"
            return f"<html><body><table><tr><td>{len(texts)} lines</td></tr></table></body></html>"

    def checkbox_fn(bgr):             # == make_golden_table_trace.checkbox_predict
        k = len(log.setdefault("checkbox", []))
        log["checkbox"].append({"shape": list(bgr.shape), "crc32": zlib.crc32(np.ascontiguousarray(bgr).tobytes())})
        tb = next(d for d in fx["layout_dets"][k] if d["category_id"] == 5)
        x0, y0 = int(tb["poly"][0]) + 30, int(tb["poly"][1]) + 20
        return [{"bbox": [x0, y0, x0 + 16, y0 + 16], "text": "checked" if k == 0 else "unchecked", "score": 0.9},
                {"bbox": [5, 5, 17, 17], "text": "unchecked", "score": 0.6}]

    class ReplayCustomTable:
        def batch_predict(self, image_list, **kwargs):
            log["table"].append({"shapes": [list(i.shape) for i in image_list],
                                 "crc32": [zlib.crc32(np.ascontiguousarray(i).tobytes()) for i in image_list],
                                 "kwargs": {k: [fill_summary(f) for f in v] for k, v in kwargs.items()}})
            return [f"<table><tr><td>{i.shape[0]}x{i.shape[1]}</td></tr></table>" if i.shape[0] > 150 else "" for i in image_list]

    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], log["layout"]), ReplayPipe(log["rec"]), formula_model=ReplayFormula(log["formula"]),
                              table_model=ReplayCustomTable() if kind == "custom" else ReplayTable(),
                              layout_batch_size=fx["layout_batch_num"], formula_level=fx["formula_level"],
                              formula_batch_size=fx["formula_batch_num"], det_batch_num=fx["ocr_config"]["Det.rec_batch_num"],
                              det_raw_fn=det_raw_fn, table_det_raw_fn=table_det_raw_fn, table_rec_fn=table_rec_fn, table_use_word_box=word_box,
                              seal_enable=fx["ocr_config"].get("seal_enable", True),
                              **(dict(checkbox_fn=checkbox_fn, checkbox_enable=True) if fx.get("checkbox_enable") else {}))
    out = pa(pages, page_scales=[fx["page_scale"]] * len(pages))
    assert log.get("checkbox", []) == tr.get("checkbox_calls", [])
    if kind == "traditional_checkbox":
        assert len(log["checkbox"]) == 2 and any({"bbox": [31, 22, 47, 38]} in c["mfd_res"] for c in tr["table_calls"])
        assert all([2, 2, 8, 8] in d["formula_boxes"] for page in out for d in page if d["category_id"] == 5)
    if word_box:       # the crops and boxes rapidocr's cal_rec_boxes was called with, and the reference's zip quirk was exercised
        assert log["table_words"] == tr["table_word_calls"]
        n_lines = sum(len(c["shapes"]) for c in tr["table_word_calls"])
        assert any(len(table_words_for(s_[0], s_[1], b)) == 0 for c in tr["table_word_calls"] for s_, b in zip(c["shapes"], c["dt_boxes"])) and n_lines > 10

    assert log["layout"] == tr["layout_calls"] and log["formula"] == tr["formula_calls"]
    assert [{k: c[k] for k in ("batch_size", "shapes", "crc32")} for c in tr["det_calls"]] == log["det"]
    assert log["rec"] == tr["rec_calls"]
    assert [{k: c[k] for k in ("shape", "crc32")} for c in tr["table_det_calls"]] == log["table_det"]
    assert len(log["table"]) == len(tr["table_calls"]) > 0
    for mine, theirs in zip(log["table"], tr["table_calls"]):
        assert mine == theirs
    if kind == "traditional":
        assert any("否" in c["texts"] for c in tr["table_calls"]) and any(c["mfd_res"] for c in tr["table_calls"])
        assert any(c["fill_image_res"] for c in tr["table_calls"]) and any("img_boxes" in d for page in fx["output"] for d in page)
    for p, (mine, theirs) in enumerate(zip(out, fx["output"])):
        assert len(mine) == len(theirs), (p, len(mine), len(theirs))
        for a, b in zip(mine, theirs):
            assert a == b and list(a) == list(b), (p, a, b)          # same fields in the same order
    assert sum(1 for page in out for d in page if "html" in d) == sum(1 for page in fx["output"] for d in page if "html" in d) > 0


def test_custom_ocr_seam_replays_the_reference_trace(golden_dir):
    """`_run_custom_ocr` (batch_analyze.py:286-333): one `batch_predict(bgr region crops, batch_size=Det.rec_batch_num)` over the text
    regions of ALL pages (crops byte for byte - on the first page every region carries a polygon, so crop_img's mask is in them), one
    result dict per region: score 0.95, stripped text, `vl_ocr`, the region's poly / polygon_points; no detector, no recogniser."""
    fx = json.loads((golden_dir / "analyze_trace_table_custom_ocr.json").read_text())
    tr = fx["trace"]
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]]))
    log = {"layout": [], "formula": [], "ocr": []}

    class ReplayCustomOcr:
        def batch_predict(self, image_list, **kwargs):
            log["ocr"].append({"shapes": [list(i.shape) for i in image_list],
                               "crc32": [zlib.crc32(np.ascontiguousarray(i).tobytes()) for i in image_list], "kwargs": dict(kwargs)})
            return [None if k % 4 == 3 else f"  region {k}: {i.shape[0]}x{i.shape[1]}\nsecond line " for k, i in enumerate(image_list)]

    def no_detector(canvases, batch_size):
        raise AssertionError("the custom-OCR seam must not reach the detector")

    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], log["layout"]), ReplayPipe([]), formula_model=ReplayFormula(log["formula"]),
                              custom_ocr=ReplayCustomOcr(), layout_batch_size=fx["layout_batch_num"], formula_level=fx["formula_level"],
                              formula_batch_size=fx["formula_batch_num"], det_batch_num=fx["ocr_config"]["Det.rec_batch_num"],
                              det_raw_fn=no_detector)
    out = pa(pages)
    assert log["layout"] == tr["layout_calls"] and log["formula"] == tr["formula_calls"]
    assert log["ocr"] == tr["custom_ocr_calls"] and len(log["ocr"]) == 1
    assert not tr["det_calls"] and not tr["rec_calls"]
    for p, (mine, theirs) in enumerate(zip(out, fx["output"])):
        assert len(mine) == len(theirs), (p, len(mine), len(theirs))
        for a, b in zip(mine, theirs):
            assert a == b and list(a) == list(b), (p, a, b)          # same keys in the same order, too


def test_two_languages_in_one_page_batch_replay_the_reference_trace(golden_dir):
    """Pages of different languages in one batch (analyze_utils.py:150-166, :222-250): detector groups by language first (in order of
    first appearance), then by 64-px size bucket; one pooled recogniser call PER LANGUAGE, each through that language's model."""
    fx = json.loads((golden_dir / "analyze_trace_seed4.json").read_text())
    tr, langs = fx["trace"], fx["page_langs"]
    assert sorted(set(langs)) == ["ch", "en"]
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]]))
    log = {"layout": [], "formula": [], "det": [], "rec": {lg: [] for lg in dict.fromkeys(langs)}}
    det_calls = iter(tr["det_calls"])

    def det_for(lang):
        def det_raw_fn(canvases, batch_size):
            call = next(det_calls)
            bgr = np.ascontiguousarray(canvases.numpy()[..., ::-1])
            log["det"].append({"batch_size": batch_size, "shapes": [list(c.shape) for c in bgr], "lang": lang,
                               "crc32": [zlib.crc32(np.ascontiguousarray(c).tobytes()) for c in bgr]})
            return [np.asarray(b, dtype=np.float32).reshape(-1, 4, 2) for b in call["boxes"]]
        return det_raw_fn

    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], log["layout"]), {lg: ReplayPipe(log["rec"][lg]) for lg in log["rec"]},
                              formula_model=ReplayFormula(log["formula"]), layout_batch_size=fx["layout_batch_num"],
                              formula_level=fx["formula_level"], formula_batch_size=fx["formula_batch_num"],
                              det_batch_num=fx["ocr_config"]["Det.rec_batch_num"], det_raw_fn={lg: det_for(lg) for lg in log["rec"]})
    out = pa(pages, page_langs=langs)
    assert log["layout"] == tr["layout_calls"] and log["formula"] == tr["formula_calls"]
    assert [{k: c[k] for k in ("batch_size", "shapes", "lang", "crc32")} for c in tr["det_calls"]] == log["det"]
    assert next(det_calls, None) is None
    assert [c["lang"] for c in tr["rec_calls"]] == list(log["rec"])                      # one call per language, first appearance first
    for c in tr["rec_calls"]:
        assert log["rec"][c["lang"]] == [{"shapes": c["shapes"]}]
    for p, (mine, theirs) in enumerate(zip(out, fx["output"])):
        assert len(mine) == len(theirs), (p, len(mine), len(theirs))
        for a, b in zip(mine, theirs):
            assert a == b and list(a) == list(b), (p, a, b)


def test_a_seal_region_without_a_seal_model_fails_loudly(golden_dir):
    """seal OCR is on by default (like the reference); its networks are not part of this build: a page with a seal region says so."""
    fx = json.loads((golden_dir / "analyze_trace_seed6.json").read_text())
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]]))
    calls = iter(fx["trace"]["det_calls"])
    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], []), ReplayPipe([]), formula_model=ReplayFormula([]),
                              layout_batch_size=2, formula_batch_size=4, det_batch_num=3,
                              det_raw_fn=lambda c, b: [np.asarray(x, dtype=np.float32).reshape(-1, 4, 2) for x in next(calls)["boxes"]])
    with pytest.raises(RuntimeError, match="seal"):
        pa(pages)

    class CustomWithSeal:             # the custom-OCR seam with an `is_seal` parameter reads the seals itself (batch_analyze.py:433-437)
        def batch_predict(self, image_list, is_seal=False, **kw):
            return ["first\nsecond" if is_seal else "region"] * len(image_list)
    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], []), ReplayPipe([]), custom_ocr=CustomWithSeal(), layout_batch_size=2,
                              det_raw_fn=lambda c, b: [])          # (CPU pages: the detector is never reached on the custom-OCR path)
    out = pa(pages)
    assert [d["text"] for page in out for d in page if d.get("original_label") == "seal"] == [["first", "second"]] * 3


def test_a_language_without_a_pipeline_fails_loudly():
    ocr = analyze.RegionOcr({"ch": object()})
    with pytest.raises(KeyError):
        ocr._pipe_for("en")


def test_optional_stages_without_their_models_fail_loudly(golden_dir):
    """checkbox detection / page orientation are seams for caller-supplied models: switching one on without its model is an error at
    construction, and a batch that mixes upright and sideways pages (one [P,H,W,3] tensor cannot hold both) says so."""
    fx = json.loads((golden_dir / "analyze_trace_seed9.json").read_text())
    with pytest.raises(ValueError, match="checkbox_fn"):
        analyze.PageAnalyzer(ReplayLayout([], []), ReplayPipe([]), checkbox_enable=True)
    with pytest.raises(ValueError, match="orientation_model"):
        analyze.PageAnalyzer(ReplayLayout([], []), ReplayPipe([]), use_doc_orientation_classify=True)

    class Mixed:
        def __init__(self):
            self.k = 0

        def predict(self, img):
            self.k += 1
            return ["0", "90"][self.k - 1]
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]]))
    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], []), ReplayPipe([]), det_raw_fn=lambda c, b: [], seal_enable=False,
                              orientation_model=Mixed(), use_doc_orientation_classify=True)
    with pytest.raises(NotImplementedError, match="mixes upright and sideways"):
        pa(pages)


def test_restore_poly_matches_the_quarter_turns():
    """restore_poly (utils/boxbase.py:328-363) against the geometry it stands for: a pixel's box on the turned page maps back onto the
    pixel's box on the page as it came in (get_rotate_image: "270" = clockwise quarter turn, "90" = counter-clockwise)."""
    h, w = 7, 11
    img = np.arange(h * w).reshape(h, w)
    for label, k in (("90", 1), ("270", -1)):
        turned = np.rot90(img, k)
        for y in range(turned.shape[0]):
            for x in range(turned.shape[1]):
                x0, y0, x1, _, _, y1, _, _ = analyze.restore_poly([x, y, x, y, x, y, x, y], label, w, h)
                assert (x0, y0) == (x1, y1) and img[y0, x0] == turned[y, x]
    assert analyze.restore_poly([1, 2, 3, 2, 3, 4, 1, 4], "180", w, h) == [w - 1 - 3, h - 1 - 4, w - 1 - 1, h - 1 - 4, w - 1 - 1, h - 1 - 2, w - 1 - 3, h - 1 - 2]


def test_page_sharded_analyzer_hands_global_page_keys_to_one_pooled_call_per_agreed_language(golden_dir):
    """Page-sharded runs (`rec_width_sync` set on the pipelines, rapiddoc_amd.dist.GlobalLineWidths): the OCR stage's pooled recogniser
    call - the one step of `BatchAnalyze.__call__` whose result depends on the OTHER pages of the batch (analyze_utils.py:216-252) - gets
    the pages' GLOBAL positions as pooling keys and is made once per language of the agreed list, in that order, on every rank (also a
    rank that holds no region of a language, or no region at all: the width exchange inside must pair up across ranks).  The output is
    the fixture's: keys and call order do not change what is read."""
    fx = json.loads((golden_dir / "analyze_trace_seed4.json").read_text())
    tr, langs = fx["trace"], fx["page_langs"]
    pages = torch.from_numpy(np.stack([synth_page(i)[0] for i in fx["page_ids"]]))
    order = []

    class ShardPipe(ReplayPipe):
        rec_width_sync = staticmethod(lambda keys, ratios: None)        # (only its presence matters to the analyzer)

        def __init__(self, lang):
            super().__init__([])
            self.lang, self.keys = lang, []

        def rec_forward_sources(self, sources, image_keys=None):
            order.append(self.lang)
            self.keys.append(image_keys)
            return super().rec_forward_sources(sources, image_keys) if sources else []
    det_calls = iter(tr["det_calls"])

    def det_raw_fn(canvases, batch_size):
        return [np.asarray(b, dtype=np.float32).reshape(-1, 4, 2) for b in next(det_calls)["boxes"]]
    pipes = {lg: ShardPipe(lg) for lg in ("ch", "en")}
    pa = analyze.PageAnalyzer(ReplayLayout(fx["layout_dets"], []), pipes, formula_model=ReplayFormula([]),
                              layout_batch_size=fx["layout_batch_num"], formula_level=fx["formula_level"], formula_batch_size=fx["formula_batch_num"],
                              det_batch_num=fx["ocr_config"]["Det.rec_batch_num"], det_raw_fn=det_raw_fn)
    with pytest.raises(ValueError, match="page_keys"):
        pa(pages, page_langs=langs)
    det_calls = iter(tr["det_calls"])
    agreed = sorted(set(langs), reverse=True)                            # an order of its own: not the order of first appearance
    keys = [40 + 3 * i for i in range(len(langs))]
    out = pa(pages, page_langs=langs, page_keys=keys, all_langs=agreed)
    assert order == agreed
    for lg in agreed:
        seen = sorted({k for per_src in pipes[lg].keys[0] for k in per_src})
        assert seen == sorted(k for k, l in zip(keys, langs) if l == lg)              # the pages of that language, by their GLOBAL key
    for mine, theirs in zip(out, fx["output"]):
        assert mine == theirs
    # a language outside the agreed list cannot be exchanged
    det_calls = iter(tr["det_calls"])
    with pytest.raises(ValueError, match="all_langs"):
        pa(pages, page_langs=langs, page_keys=keys, all_langs=[agreed[0]])
    # a rank without a single region: one (empty) pooled call per agreed language all the same
    order.clear()
    pa_empty = analyze.PageAnalyzer(ReplayLayout([[] for _ in langs], []), pipes, formula_model=ReplayFormula([]), layout_batch_size=2,
                                    formula_batch_size=4, det_batch_num=3, det_raw_fn=lambda c, b: [])
    assert pa_empty(pages, page_langs=langs, page_keys=keys, all_langs=agreed) == [[] for _ in langs]
    assert order == agreed
