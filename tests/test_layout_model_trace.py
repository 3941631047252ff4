"""Row a2 pinned to the reference's own wrapper stack: LayoutModel.batch_predict (rapiddoc_amd/layout_model.py) replayed against
traces of RapidLayoutModel.batch_predict -> RapidLayout.__call__ -> PPDocLayoutModelHandler.__call__ -> PPPostProcess recorded by
tests/golden/make_golden_layout_trace.py (reference code imported unmodified, detector session stood in by SyntheticBoxSession).
Compared: how pages are chunked into session calls (input shape per model, scale factors), and the output dicts - category ids,
labels, reading order, polys, rounded scores, the inline-formula re-labelling - value for value.
The pre-process pixels are NOT what this test is about (the stand-in ignores them; GPU parity of the resize lives in
test_gpu_image_ops.py), so `preprocess` is replaced by its shape / scale-factor half and the test runs without a GPU."""
import json
from pathlib import Path

import numpy as np
import pytest
import torch

from rapiddoc_amd import _lib
from rapiddoc_amd.layout_model import LayoutModel, SyntheticBoxSession

TRACE = json.loads((Path(__file__).parent / "golden" / "layout_trace.json").read_text())
pytestmark = pytest.mark.skipif(not _lib.LIB_PATH.exists(), reason="librapiddoc_mi355.so not built (rd_layout_postprocess is host C++)")


def _shape_only_preprocess(self, pages):
    S = self.m["S"]
    sf = np.asarray([(S / p.shape[0], S / p.shape[1]) for p in pages], np.float32)
    return torch.zeros((len(pages), 3, S, S), dtype=torch.float32), sf


@pytest.mark.parametrize("case", TRACE["cases"], ids=[f"{i}-{c['model_type']}" for i, c in enumerate(TRACE["cases"])])
def test_layout_wrapper_replays_the_reference_trace(case, monkeypatch):
    monkeypatch.setattr(LayoutModel, "preprocess", _shape_only_preprocess)
    size = {"pp_doclayout_s": 480, "pp_doclayout_l": 640}.get(case["model_type"], 800)
    session = SyntheticBoxSession(case["labels"], case["boxes_per_page"], case["ncol"], seed=case["seed"], size=size, twins=case["twins"],
                                  masks=case["masks"])
    model = LayoutModel(session, case["model_type"], conf_thresh=case["conf_thresh"], layout_shape_mode=case["layout_shape_mode"] or "auto")
    pages = [np.zeros((h, w, 3), np.uint8) for h, w in case["page_hw"]]
    out = model.batch_predict(pages, case["batch_size"])

    assert len(session.calls) == len(case["session_calls"])
    for (shape, sf), ref in zip(session.calls, case["session_calls"]):
        assert list(shape) == ref["shape"]
        assert np.array_equal(sf, np.asarray(ref["scale_factor"], np.float32))

    assert len(out) == len(case["layout_dets"])
    for pi, (mine, ref) in enumerate(zip(out, case["layout_dets"])):
        assert len(mine) == len(ref), f"page {pi}"
        for bi, (a, b) in enumerate(zip(mine, ref)):
            where = f"page {pi} box {bi}"
            assert set(a) == set(b), where
            assert a["category_id"] == b["category_id"] and a["original_label"] == b["original_label"], where
            assert a["original_order"] == b["original_order"] and a["polygon_points"] == b["polygon_points"], where
            assert [float(v) for v in a["poly"]] == b["poly"], where
            assert float(a["score"]) == b["score"], where


def test_traces_cover_the_inline_formula_relabelling_and_reading_order():
    by_type = {}
    for c in TRACE["cases"]:
        by_type.setdefault(c["model_type"], []).append(c)
    cats = {d["category_id"] for c in by_type["pp_doclayout_s"] for p in c["layout_dets"] for d in p}
    assert 13 in cats, "no InlineEquation produced by check_inline_formula in the pp_doclayout families"
    for page in by_type["pp_doclayoutv3"][0]["layout_dets"]:               # V3: a reading order per box, 0 .. n-1 after the sort
        assert [d["original_order"] for d in page] == list(range(len(page)))
    assert all(d["original_order"] == -1 for p in by_type["pp_doclayout_l"][0]["layout_dets"] for d in p)


def test_traces_cover_the_mask_branch_in_three_shape_modes():
    masked = [c for c in TRACE["cases"] if c["masks"]]
    assert {c["layout_shape_mode"] for c in masked} == {None, "rect", "poly"}          # None = the default, "auto"
    for c in masked:
        polys = [d["polygon_points"] for p in c["layout_dets"] for d in p]
        if c["layout_shape_mode"] == "rect":
            assert all(p is None for p in polys)
        else:
            assert all(p is not None for p in polys) and max(len(p) for p in polys) > 8 and min(len(p) for p in polys) == 4


# ---- DocLayout-YOLO family (model_handler/doc_layout/): traces in tests/golden/layout_trace_yolo.json ------------------------------
YOLO = json.loads((Path(__file__).parent / "golden" / "layout_trace_yolo.json").read_text())


@pytest.mark.parametrize("case", YOLO["cases"], ids=lambda c: f"yolo-seed{c['seed']}")
def test_doclayout_yolo_wrapper_replays_the_reference_trace(case, monkeypatch):
    """LetterBox geometry (new size, left / top padding with 114), channel flip, float64 division by 255 -> the session input tensor
    BYTE FOR BYTE (pages are blank and the stand-in for the linear resize returns a blank image on both sides: resize pixels are a GPU
    test's business); one session call per page; rows -> page pixels (padding, gain, clip) in float32; labels -> category ids
    ('isolate_formula' -> 14); the inline-formula rule; scores rounded to 3 decimals."""
    from rapiddoc_amd.layout_model import SyntheticYoloSession
    monkeypatch.setattr(LayoutModel, "_resize_linear_u8", lambda self, page, new_h, new_w: torch.zeros((3, new_h, new_w), dtype=torch.float32))
    monkeypatch.setattr(torch.cuda, "is_available", lambda: False)
    session = SyntheticYoloSession(case["labels"], case["n"], seed=case["seed"])
    model = LayoutModel(session, "doclayout_docstructbench", conf_thresh=case["conf_thresh"])
    out = model.batch_predict([np.zeros((h, w, 3), np.uint8) for h, w in case["page_hw"]], 3)
    assert [{"shape": list(sh), "crc32": crc} for sh, crc in session.calls] == case["session_calls"]
    assert len(out) == len(case["layout_dets"])
    for pi, (mine, ref) in enumerate(zip(out, case["layout_dets"])):
        assert len(mine) == len(ref), pi
        for a, b in zip(mine, ref):
            assert list(a) == list(b)
            assert (a["category_id"], a["original_label"], a["original_order"], a["polygon_points"], float(a["score"])) == \
                   (b["category_id"], b["original_label"], b["original_order"], b["polygon_points"], b["score"]), pi
            assert [float(v) for v in a["poly"]] == b["poly"], pi
    cats = {d["category_id"] for p in case["layout_dets"] for d in p}
    assert 14 in cats and 1 in cats


def test_letterbox_geometry_known_answers():
    g = LayoutModel.letterbox_geometry
    assert g(1684, 1191, 1024) == (724, 1024, 150, 0)            # portrait A4: 150 px of padding left and right
    assert g(1024, 1024, 1024) == (1024, 1024, 0, 0)
    assert g(1000, 2048, 1024) == (1024, 500, 0, 262)
    assert g(480, 640, 1024) == (1024, 768, 0, 128)              # scaled UP (scaleup=True)
