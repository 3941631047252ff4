#!/usr/bin/env python3
"""Traces of the REFERENCE's table stage, run with recording stand-ins for the models (build container only).

    python tests/golden/make_golden_table_trace.py        # writes tests/golden/analyze_trace_table_{traditional,custom,custom_ocr,traditional_words}.json
                                                          # (the third: the custom-OCR seam of the same driver, tables off)

What runs is the reference's code, unmodified (same import machinery as make_golden_analyze.py):
    rapid_doc/backend/pipeline/batch_analyze.py:78-164,230-256,351-410   BatchAnalyze.__call__ with table_enable: the table crop (the box
                                                                          snapped outwards to multiples of 5 px), both table seams
    rapid_doc/backend/pipeline/analyze_utils.py:295-427,478-540          _process_single_table, _run_table_ocr
    rapid_doc/model/ocr/rapid_ocr.py:225-299,404-471                     RapidOcrModel.ocr (det only: sort / cut around formulas, no merge;
                                                                          rec only: the recogniser loop)
    rapid_doc/utils/ocr_utils.py, rapid_doc/model/table/utils.py         get_adjusted_mfdetrec_res(return_text), normalize_table_ocr_text
Stood in for: the table model (`predict(...)` of the traditional seam / `batch_predict(...)` of a CustomBaseModel) - records what it is
handed (image crc32, OCR boxes / texts / scores, formula boxes, flags) and answers with an HTML string; rapidocr's detector and
recogniser inside a REAL RapidOcrModel object (`text_detector`: seeded boxes, records the canvas; `text_recognizer`: the recording
stand-in of make_golden_recbatch.py whose text is a function of the crop's shape); the orientation classifier (answers "0": the
orientation sub-stage is outside SURVEY s8); cv2 and the absent wheels as in make_golden_analyze.py.
table_config = {use_word_box: False} for the first three traces; `traditional_words` runs the reference's DEFAULT ({}: use_word_box True,
analyze_utils.py:308): `ocr(det=False, return_word_box=True)` -> rapidocr's cal_rec_boxes (absent: a recording stand-in whose words are a
function of the line crop's shape and box, `table_words_for`) -> calc_word_boxes / map_boxes_to_original -> one OCR entry per word.
Images inside a table travel to the table model (`layout_image_list` -> extract_table_fill_image's layout branch; the page carries no PDF images)."""
import importlib
import json
import sys
import types
import zlib
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden_analyze as MGA  # noqa: E402


def table_text_and_score(h, w):
    """What the stand-in recogniser 'reads' from a table line crop: a function of its shape only (the recogniser loop sorts crops
    by aspect ratio, so a position in the call is not something both sides share)."""
    return ("香" if (h + w) % 7 == 0 else f"<{h}x{w}>"), ((h * 37 + w * 11) % 1000) / 1000.0


def table_words_for(h, w, dt_box):
    """What the stand-in `cal_rec_boxes` answers for a table line (word, confidence, box) - a function of the line crop's shape and of its
    detection box only: (h + w) % 4 words (NONE for some lines: the reference then drops the line from the word results and its zip with
    the texts shifts, rapid_ocr.py:295,325-326), one of them without a box now and then (skipped, :318-319), boxes partly outside the
    table image (clipped by map_boxes_to_original) with fractional corners (truncated by the int32 cast)."""
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


def fill_summary(fill_image_res):
    """What a table model is handed for the images inside a table, minus the random uuid; the PIL crop as size + crc32."""
    out = []
    for f in fill_image_res:
        d = {k: json.loads(json.dumps(v)) for k, v in f.items() if k not in ("uuid", "pil_image")}
        d["keys"] = list(f)
        d["pil_size"] = list(f["pil_image"].size)
        d["pil_crc32"] = zlib.crc32(np.ascontiguousarray(np.asarray(f["pil_image"])).tobytes())
        out.append(d)
    return out


class TableRecognizer:
    """`self.text_recognizer` of the RapidOcrModel that serves tables (cf. make_golden_recbatch.RecordingRecognizer)."""
    rec_batch_num, rec_image_shape = 6, [3, 48, 320]

    def __init__(self):
        self.chunk = []

    def resize_norm_img(self, img, max_wh_ratio):
        self.chunk.append((int(img.shape[0]), int(img.shape[1])))
        return np.zeros((3, 48, int(48 * max_wh_ratio)), np.float32)

    def session(self, batch):
        return batch

    def postprocess_op(self, preds, return_word_box, wh_ratio_list=None, max_wh_ratio=None):
        out = [table_text_and_score(h, w) for h, w in self.chunk]
        words = [("word-info", h, w) if return_word_box else None for h, w in self.chunk]     # rapidocr's WordInfo: opaque to the reference
        self.chunk = []
        return out, words


def layout_with_tables(rng, H, W):
    dets = MGA.layout_for_page(rng, H, W)
    y = min(max(d["poly"][5] for d in dets if d["original_label"] != "abandon") + 30.3, H - 260.0)      # keep the table on the page

    def add(cat, label, x0, y0, x1, y1, score):
        dets.append({"category_id": cat, "original_label": label, "original_order": len(dets),
                     "poly": [x0, y0, x1, y0, x1, y1, x0, y1], "polygon_points": None, "score": score})
    add(5, "table", 101.7, y, 1083.2, y + 236.6, 0.93)                       # a table with an inline formula inside it
    add(13, "inline_formula", 420.4, y + 40.2, 560.9, y + 71.8, 0.81)
    add(3, "image", 700.3, y + 100.6, 880.9, y + 200.2, 0.77)               # a picture inside that table: handed to the table model
    add(3, "image", 1000.0, y + 150.0, 1150.0, y + 230.0, 0.6)               # one that only half lies in it (< 0.8 of its area): not
    add(5, "table", 641.0, 133.0, 1103.6, 248.2, 0.4)                        # a second, smaller table without formulas
    return dets


def main():
    ba, reg = MGA.import_reference()
    from rapid_doc.backend.pipeline.model_list import AtomicModel
    from rapid_doc.model.custom import CustomBaseModel
    from rapiddoc_amd.pages import synth_page
    # the reference's RapidOcrModel class: its module-level reach into the absent rapidocr package is stubbed as in make_golden_recbatch.py
    for name, attrs in (("rapid_doc.model.ocr.ocr_patch", {"apply_ocr_patch": lambda: None}),
                        ("rapid_doc.model.ocr.seal_crop", {"SortPolyBoxes": type("U", (), {}), "CropByPolys": type("U", (), {})})):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
    import importlib.metadata as md
    md.version = lambda name: "0.0.0"
    finder = next(f for f in sys.meta_path if isinstance(f, MGA._RefFinder))
    ro = None
    for _ in range(40):                               # further absent wheels (rapidocr, ...) become inert mocks, one by one
        try:
            ro = importlib.import_module("rapid_doc.model.ocr.rapid_ocr")
            break
        except ModuleNotFoundError as e:
            top = (e.name or "").split(".")[0]
            if not top or top == "rapid_doc" or top in finder.mocked:
                raise
            finder.mocked.add(top)
            sys.modules.pop("rapid_doc.model.ocr.rapid_ocr", None)
    ro.TextRecInput = lambda img, return_word_box=False: types.SimpleNamespace(img=img, return_word_box=return_word_box)   # rapidocr dataclasses
    ro.TextRecOutput = lambda imgs, txts, scores, words, elapse: types.SimpleNamespace(txts=list(txts), scores=list(scores), word_results=words)

    # "traditional_checkbox": checkbox detection on (batch_analyze.py:207-219) - a hit inside the first table is whited out of the table
    # detector's input and travels to the table model with its `checkbox` text, every hit is listed in `formula_boxes`
    kinds = ("traditional", "custom", "custom_ocr", "traditional_words", "traditional_checkbox")
    for kind in kinds:
        rng = np.random.default_rng(8000 + kinds.index(kind))
        trace = {"det_calls": [], "rec_calls": [], "formula_calls": [], "layout_calls": [], "table_det_calls": [], "table_calls": []}
        page_ids = [int(rng.integers(0, 1000)) for _ in range(2)]
        pages = [synth_page(i)[0] for i in page_ids]
        H, W = pages[0].shape[:2]
        dets = [layout_with_tables(rng, H, W) for _ in pages]
        if kind == "custom_ocr":              # one page with polygons on every box (crop_img's mask), one without
            dets[0] = MGA.layout_for_page(rng, H, W, polygons=True)
        ocr = MGA.RecordingOcr(trace)

        def checkbox_predict(bgr):           # utils/checkbox_det_cls.py checkbox_predict(bgr page) -> [{'bbox', 'text', ...}]
            bgr = np.asarray(bgr)
            k = len(trace.setdefault("checkbox_calls", []))
            trace["checkbox_calls"].append({"shape": list(bgr.shape), "crc32": zlib.crc32(np.ascontiguousarray(bgr).tobytes())})
            tb = next(d for d in dets[k] if d["category_id"] == 5)
            x0, y0 = int(tb["poly"][0]) + 30, int(tb["poly"][1]) + 20
            return [{"bbox": [x0, y0, x0 + 16, y0 + 16], "text": "checked" if k == 0 else "unchecked", "score": 0.9},
                    {"bbox": [5, 5, 17, 17], "text": "unchecked", "score": 0.6}]
        ba.checkbox_predict = checkbox_predict

        def text_detector(img):
            img = np.asarray(img)
            crc = zlib.crc32(np.ascontiguousarray(img).tobytes())
            b = MGA.det_boxes_for(img.shape[0] + 100, img.shape[1] + 100, crc & 0xffff) - 50.0      # (that helper leaves a 50-px margin)
            b = b[(b[:, :, 0].min(1) >= 0) & (b[:, :, 1].min(1) >= 0)]
            trace["table_det_calls"].append({"shape": list(img.shape), "crc32": crc, "boxes": b.tolist()})
            return types.SimpleNamespace(boxes=(b if len(b) else None), elapse=0.0)

        table_ocr = object.__new__(ro.RapidOcrModel)
        table_ocr.text_detector, table_ocr.text_recognizer = text_detector, TableRecognizer()
        table_ocr.is_seal, table_ocr.enable_merge_det_boxes, table_ocr.drop_score = False, False, 0.5

        def cal_rec_boxes(imgs, dt_boxes, rec_res, return_single_char_box):      # rapidocr's CalRecBoxes.__call__ (absent): recording stand-in
            assert len(imgs) == len(dt_boxes) == len(rec_res.txts) and return_single_char_box is False
            assert all(wi == ("word-info", i.shape[0], i.shape[1]) for wi, i in zip(rec_res.word_results, imgs))
            trace["table_word_calls"] = trace.get("table_word_calls", []) + [{"shapes": [list(i.shape) for i in imgs],
                                                                              "dt_boxes": [np.asarray(b).tolist() for b in dt_boxes]}]
            rec_res.word_results = tuple(table_words_for(i.shape[0], i.shape[1], b) for i, b in zip(imgs, dt_boxes))
            return rec_res
        table_ocr.ocr_engine = types.SimpleNamespace(cal_rec_boxes=cal_rec_boxes, return_single_char_box=False)

        class TableModel:                     # the traditional seam: RapidTableModel.predict (rapid_table.py:120)
            def predict(self, image, ocr_result, fill_image_res, mfd_res, skip_text_in_image, use_img2table, skip_table_orientation=False):
                boxes, texts, scores = ocr_result if ocr_result else ([], [], [])
                trace["table_calls"].append({
                    "shape": list(image.shape), "crc32": zlib.crc32(np.ascontiguousarray(image).tobytes()),
                    "boxes": [np.asarray(b, dtype=np.float64).tolist() for b in boxes], "texts": list(texts), "scores": [float(s) for s in scores],
                    "fill_image_res": fill_summary(fill_image_res), "mfd_res": json.loads(json.dumps(mfd_res)),
                    "flags": [bool(skip_text_in_image), bool(use_img2table), bool(skip_table_orientation)]})
                if len(texts) % 2 == 0 and kind != "traditional_checkbox":       # an answer without a table in it: the reference leaves the region without `html`
                    return "<html><body>nothing found</body></html>"
                return f"<html><body><table><tr><td>{len(texts)} lines</td></tr></table></body></html>"

        class CustomTable(CustomBaseModel):   # seam S1 (model/custom/__init__.py)
            def batch_predict(self, image_list, **kwargs):
                trace["table_calls"].append({"shapes": [list(i.shape) for i in image_list],
                                             "crc32": [zlib.crc32(np.ascontiguousarray(i).tobytes()) for i in image_list],
                                             "kwargs": {"fill_image_res_list": [fill_summary(f) for f in kwargs["fill_image_res_list"]]}
                                             if set(kwargs) == {"fill_image_res_list"} else {"unexpected": sorted(kwargs)}})
                return [f"<table><tr><td>{i.shape[0]}x{i.shape[1]}</td></tr></table>" if i.shape[0] > 150 else "" for i in image_list]

        class CustomOcr(CustomBaseModel):     # seam S1 for OCR: one multi-line string per layout region (batch_analyze.py:286-333)
            def batch_predict(self, image_list, **kwargs):
                trace["custom_ocr_calls"] = trace.get("custom_ocr_calls", []) + [{
                    "shapes": [list(i.shape) for i in image_list], "crc32": [zlib.crc32(np.ascontiguousarray(i).tobytes()) for i in image_list],
                    "kwargs": dict(kwargs)}]
                return [None if k % 4 == 3 else f"  region {k}: {i.shape[0]}x{i.shape[1]}\nsecond line " for k, i in enumerate(image_list)]

        class Registry:
            def get_atom_model(self, atom_model_name, **kw):
                if atom_model_name == AtomicModel.ImgOrientationCls:
                    return types.SimpleNamespace(predict=lambda img, det_res=None: "0")
                if atom_model_name == "table":
                    return TableModel()
                assert atom_model_name == AtomicModel.OCR, atom_model_name
                if kw.get("det_db_box_thresh") == 0.5:
                    assert kw.get("det_db_unclip_ratio") == 1.6 and kw.get("enable_merge_det_boxes") is False
                    return table_ocr
                return ocr
        reg.AtomModelSingleton = Registry
        ba.AtomModelSingleton = Registry
        sys.modules["rapid_doc.backend.pipeline.analyze_utils"].AtomModelSingleton = Registry

        class Model:
            device = "cpu"
            layout_model = MGA.RecordingLayout(trace, dets)
            ocr_model = CustomOcr() if kind == "custom_ocr" else ocr
            formula_model = MGA.RecordingFormula(trace)
            table_model = CustomTable() if kind == "custom" else TableModel()

        class Manager:
            def get_model(self, **kw):
                return Model()
        ocr_cfg = {"use_det_mode": "ocr", "Det.rec_batch_num": 3, "seal_enable": False}
        # table_image_enable stays at its default, True; "traditional_words" runs the reference's DEFAULT table config (use_word_box True)
        table_cfg = {} if kind == "traditional_words" else {"use_word_box": False}
        fixture_extra = {"checkbox_enable": kind == "traditional_checkbox"}
        analyzer = ba.BatchAnalyze(Manager(), batch_ratio=1, formula_enable=True, table_enable=(kind != "custom_ocr"), layout_config={"batch_num": 2},
                                   ocr_config=ocr_cfg, table_config=table_cfg,
                                   formula_config={"formula_level": 0, "batch_num": 4, "bbox_expand_px": 2},
                                   checkbox_config={"checkbox_enable": True} if kind == "traditional_checkbox" else None)
        ba.clean_vram = lambda *a, **k: None
        from PIL import Image
        out = analyzer([(Image.fromarray(p), 2.0, True, "ch", {"blocks": [], "ori_image_list": []}) for p in pages])

        def clean(o):
            if isinstance(o, dict):
                return {k: clean(v) for k, v in o.items()}
            if isinstance(o, (list, tuple)):
                return [clean(v) for v in o]
            if isinstance(o, (np.floating, np.integer)):
                return o.item()
            return o
        fixture = {"kind": kind, "page_ids": page_ids, "page_hw": [H, W], "page_scale": 2.0, "formula_enable": True, "formula_level": 0,
                   "ocr_config": ocr_cfg, "table_config": table_cfg, "layout_batch_num": 2, "formula_batch_num": 4, "layout_dets": dets,
                   "trace": clean(trace), "output": clean(out), **fixture_extra}
        (HERE / f"analyze_trace_table_{kind}.json").write_text(json.dumps(fixture))
        print(f"{kind}: table det calls {[c['shape'][:2] for c in trace['table_det_calls']]}, table calls {len(trace['table_calls'])}, "
              f"html fields {sum(1 for p in fixture['output'] for d in p if 'html' in d)}")


if __name__ == "__main__":
    main()
