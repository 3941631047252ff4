#!/usr/bin/env python3
"""Traces of the REFERENCE's own layout wrapper stack, run with a stand-in detector session (build container only).

    python tests/golden/make_golden_layout_trace.py        # writes tests/golden/layout_trace.json

What runs is the reference's code, unmodified, imported from /root/reference:
    rapid_doc/model/layout/rapid_layout.py:8-122                       RapidLayoutModel.__init__ (threshold defaults per model),
                                                                        batch_predict (label -> CategoryId, poly, score rounding),
                                                                        check_inline_formula
    rapid_doc/model/layout/rapid_layout_self/main.py:16-56             RapidLayout.__init__ (conf_thresh defaults), __call__ (chunks)
    .../model_handler/pp_doclayout/main.py:14-139                      input size per model, scale_factor, session call, output split
    .../model_handler/pp_doclayout/pre_process.py, post_process.py     PPPreProcess (shape / dtype only here), PPPostProcess
What is stood in for: the detector session (rapiddoc_amd.layout_model.SyntheticBoxSession: deterministic boxes, a pure function of
the batch size, the page's position in the call and its scale factors - the same class the replay test plugs into this repo's
LayoutModel), `cv2.resize` (a blank image of the requested size: pre-process PIXELS are pinned elsewhere), the engine factory /
model download / image loader of RapidLayout.__init__, and the absent wheels (mocks, as in make_golden_analyze.py).
The cases whose session returns instance MASKS go through the reference's polygon branch; there `cv2` / `shapely` are the thin
modules of make_golden_polygon.py (this repo's C primitives behind OpenCV's / shapely's names - see that file for what this pins).
The committed JSON holds page sizes, the recorded session calls (shapes, scale factors) and the reference's output dicts."""
import importlib
import json
import sys
import types
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden_analyze as MGA  # noqa: E402
import make_golden_polygon as MGP  # noqa: E402

sys.path.insert(0, str(HERE.parents[1]))
from rapiddoc_amd.layout_model import SyntheticBoxSession, SyntheticYoloSession, _YOLO_LABELS  # noqa: E402


def import_layout():
    finder = MGA._RefFinder()
    sys.meta_path.insert(0, finder)
    cv2, poly_cv2 = MGA._fake_cv2(), MGP.fake_cv2()
    for name in ("RETR_EXTERNAL", "CHAIN_APPROX_SIMPLE", "INTER_NEAREST", "findContours", "contourArea", "arcLength", "approxPolyDP",
                 "minAreaRect", "boxPoints"):
        setattr(cv2, name, getattr(poly_cv2, name))
    # pages (3 channels) -> a blank image of the requested size; instance-mask crops (2-D) -> the nearest-neighbour primitive
    cv2.resize = lambda img, size, interpolation=None: (np.zeros((int(size[1]), int(size[0]), 3), np.uint8) if np.ndim(img) == 3
                                                        else poly_cv2.resize(img, size, interpolation))
    # DocLayout-YOLO's letterbox (doc_layout/utils.py:56-66): constant border = np.pad
    cv2.INTER_LINEAR, cv2.BORDER_CONSTANT = 1, 0
    cv2.copyMakeBorder = lambda img, top, bottom, left, right, kind, value=None: np.stack(
        [np.pad(img[..., c], ((top, bottom), (left, right)), constant_values=value[c]) for c in range(img.shape[2])], axis=-1)
    sys.modules["cv2"] = cv2
    sys.modules["shapely"], sys.modules["shapely.geometry"] = MGP.fake_shapely()
    for _ in range(60):
        try:
            return importlib.import_module("rapid_doc.model.layout.rapid_layout")
        except ModuleNotFoundError as e:
            top = (e.name or "").split(".")[0]
            if not top or top == "rapid_doc" or top in finder.mocked:
                raise
            finder.mocked.add(top)
            for k in [k for k in sys.modules if k.startswith("rapid_doc.")]:
                del sys.modules[k]
    raise RuntimeError("could not import rapid_doc.model.layout.rapid_layout")


def main():
    RL = import_layout()
    main_mod = importlib.import_module("rapid_doc.model.layout.rapid_layout_self.main")
    ModelType = RL.ModelType
    RL.get_device = lambda: "cpu"
    main_mod.ModelProcessor.get_model_path = staticmethod(lambda *a, **k: "unused.onnx")
    main_mod.LoadImage = lambda: (lambda x: x)                       # pages arrive as arrays
    ignore = ["number", "footnote", "header", "header_image", "footer", "footer_image", "aside_text"]
    d_pp, d_plus, d_v2 = RL.get_cls_dicts(ignore)
    sizes = [(1684, 1191), (1000, 800), (842, 595), (1191, 1684), (640, 480)]
    cases = []
    for name, mt, labels, ncol, n_boxes, batch, cfg, twins in (
            ("pp_doclayoutv3", ModelType.PP_DOCLAYOUTV3, list(d_v2), 7, 60, 2, {"masks": True}, None),
            ("pp_doclayoutv3", ModelType.PP_DOCLAYOUTV3, list(d_v2), 7, 30, 5, {"masks": True, "layout_shape_mode": "rect"}, None),
            ("pp_doclayoutv3", ModelType.PP_DOCLAYOUTV3, list(d_v2), 7, 30, 3, {"masks": True, "layout_shape_mode": "poly"}, None),
            ("pp_doclayoutv3", ModelType.PP_DOCLAYOUTV3, list(d_v2), 7, 60, 2, {}, None),
            ("pp_doclayout_plus_l", ModelType.PP_DOCLAYOUT_PLUS_L, list(d_plus), 6, 60, 3, {}, ["formula", "text", 3]),
            ("pp_doclayout_s", ModelType.PP_DOCLAYOUT_S, list(d_pp), 6, 80, 2, {}, ["formula", "text", 2]),
            ("pp_doclayout_l", ModelType.PP_DOCLAYOUT_L, list(d_pp), 6, 80, 5, {}, None),
            ("pp_doclayoutv3", ModelType.PP_DOCLAYOUTV3, list(d_v2), 7, 40, 1, {"conf_thresh": 0.6}, ["display_formula", "text", 2])):
        seed = len(cases)
        cfg = dict(cfg)
        with_masks = cfg.pop("masks", False)
        session = SyntheticBoxSession(labels, n_boxes, ncol, seed=seed, size={"pp_doclayout_s": 480, "pp_doclayout_l": 640}.get(name, 800),
                                      twins=twins, masks=with_masks)
        main_mod.get_engine = lambda engine_type: (lambda cfg_: session)
        model = RL.RapidLayoutModel(dict(model_type=mt, **cfg))
        pages = [np.zeros((h, w, 3), np.uint8) for h, w in sizes]
        out = model.batch_predict(pages, batch)
        dets = [[{"category_id": int(d["category_id"]), "original_label": d["original_label"],
                  "original_order": (None if d["original_order"] is None else int(d["original_order"])),
                  "poly": [float(v) for v in d["poly"]], "polygon_points": d["polygon_points"], "score": float(d["score"])} for d in page]
                for page in out]
        cases.append({"model_type": name, "labels": labels, "ncol": ncol, "boxes_per_page": n_boxes, "seed": seed, "batch_size": batch, "twins": twins,
                      "masks": with_masks, "layout_shape_mode": cfg.get("layout_shape_mode"), "conf_thresh": cfg.get("conf_thresh"), "page_hw": [list(s) for s in sizes],
                      "session_calls": [{"shape": list(shape), "scale_factor": sf.tolist()} for shape, sf in session.calls],
                      "layout_dets": dets})
        print(f"{name}: {len(session.calls)} session calls, dets per page {[len(p) for p in dets]}, polygon sizes "
              f"{sorted({len(d['polygon_points']) for p in dets for d in p if d['polygon_points'] is not None})}")
    # ---- DocLayout-YOLO (model_handler/doc_layout/): letterbox geometry, one page per session call, box rescaling, category ids
    yolo_cases = []
    for seed, conf in ((0, None), (1, 0.45)):
        session = SyntheticYoloSession(_YOLO_LABELS[:10] + ["inline_formula", "isolated_formula"], 50, seed=seed)
        main_mod.get_engine = lambda engine_type: (lambda cfg_: session)
        cfg = {"model_type": ModelType.DOCLAYOUT_DOCSTRUCTBENCH}
        if conf:
            cfg["conf_thresh"] = conf
        model = RL.RapidLayoutModel(cfg)
        ysizes = sizes + [(1024, 1024), (2048, 1000)]
        out = model.batch_predict([np.zeros((h, w, 3), np.uint8) for h, w in ysizes], 3)
        dets = [[{"category_id": int(d["category_id"]), "original_label": d["original_label"], "original_order": int(d["original_order"]),
                  "poly": [float(v) for v in d["poly"]], "polygon_points": d["polygon_points"], "score": float(d["score"])} for d in page]
                for page in out]
        yolo_cases.append({"labels": session.characters, "n": 50, "seed": seed, "conf_thresh": conf, "page_hw": [list(s) for s in ysizes],
                           "session_calls": [{"shape": list(sh), "crc32": crc} for sh, crc in session.calls], "layout_dets": dets})
        print(f"doclayout_docstructbench: {len(session.calls)} session calls, dets per page {[len(p) for p in dets]}")
    (HERE / "layout_trace_yolo.json").write_text(json.dumps({"cases": yolo_cases}))
    (HERE / "layout_trace.json").write_text(json.dumps({"source": "rapid_doc/model/layout/rapid_layout.py + rapid_layout_self", "cases": cases}))


if __name__ == "__main__":
    main()
