"""CPU: layout post-process (C++ behind the C-ABI, rapiddoc_amd/layout_host.py) vs results of the reference's own
PPPostProcess in rect mode, captured by tests/golden/make_golden.py.  Kept set, order and coordinates bit-identical."""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module", autouse=True)
def _built():
    from rapiddoc_amd import build as rd_build
    rd_build.build(verbose=False)


def _cfg(case):
    def undict(v, conv):
        return {int(k): conv(x) for k, x in v.items()} if isinstance(v, dict) else v
    thr = undict(case["thr"], float)
    merge = undict(case["merge"], str)
    unclip = undict(case["unclip"], tuple)
    return thr, merge, unclip


@pytest.mark.parametrize("ci", range(6))
def test_layout_post_matches_reference(ci):
    from rapiddoc_amd.layout_host import LayoutPostProcess
    g = json.loads((GOLD / f"layout_post_seed{ci}.json").read_text())
    thr, merge, unclip = _cfg(g["case"])
    pp = LayoutPostProcess(g["labels"], thr, 0.5, layout_merge_bboxes_mode=merge, layout_unclip_ratio=unclip, scale_size=(800, 800))
    res = pp(np.array(g["boxes"], dtype=np.float32), g["img_size"], None, "rect")
    res = [] if isinstance(res, np.ndarray) else res
    assert len(res) == len(g["result"])
    for a, b in zip(res, g["result"]):
        assert a["cls_id"] == b["cls_id"] and a["label"] == b["label"] and a["order"] == b["order"]
        assert a["score"] == b["score"]
        assert a["coordinate"] == b["coordinate"]


def test_layout_post_empty_and_bad_shape():
    from rapiddoc_amd.layout_host import LayoutPostProcess
    pp = LayoutPostProcess(["a", "image"], 0.5)
    assert len(pp(np.zeros((0, 6), np.float32), (100, 100))) == 0
    assert len(pp(np.array([[0, 0.1, 1, 1, 5, 5]], np.float32), (100, 100))) == 0   # below threshold
    with pytest.raises(ValueError):
        pp(np.zeros((2, 5), np.float32), (100, 100))


def test_shipped_tables_are_consistent():
    t = json.loads((GOLD / "layout_tables.json").read_text())
    assert len(t["PP_DOCLAYOUTV2_layout_merge_bboxes_mode"]) == 25
    assert set(t["PP_DOCLAYOUTV2_layout_merge_bboxes_mode"].values()) <= {"union", "large", "small"}


@pytest.mark.parametrize("seed", range(4))
def test_filter_overlap_boxes_matches_reference(seed):
    from rapiddoc_amd.layout_host import filter_overlap_boxes
    g = json.loads((GOLD / f"layout_overlap_seed{seed}.json").read_text())
    for flag in (False, True):
        kept = filter_overlap_boxes(g["dets"], flag)
        assert [d["uid"] for d in kept] == g[f"kept_custom_ocr_{flag}"]


def test_region_split_and_crop_geometry():
    from rapiddoc_amd.layout_host import crop_geometry, split_regions
    mk = lambda cid, x0, y0, x1, y1: {"category_id": cid, "poly": [x0, y0, x1, y0, x1, y1, x0, y1]}
    dets = [mk(1, 0, 0, 10, 10), mk(5, 0, 0, 50, 50), mk(8, 3.7, 4.2, 20.9, 9.1), mk(3, 1, 1, 2, 2), mk(13, 0, 0, 1, 1)]
    ocr, tables, formulas = split_regions(dets)
    assert len(ocr) == 1 and len(tables) == 1 and len(formulas) == 2
    assert formulas[0]["bbox"] == [3, 4, 20, 9]
    assert crop_geometry(mk(1, 10.5, 20.5, 110.2, 60.9), 50, 50) == [50, 50, 10, 20, 110, 60, 200, 140]


def test_category_tables_match_reference_capture():
    from rapiddoc_amd.layout_host import CATEGORY_ID, category_map, to_layout_dets
    g = json.loads((GOLD / "layout_category_maps.json").read_text())
    assert CATEGORY_ID == g["category_id"]
    for fam in ("pp_doclayout", "pp_doclayout_plus", "pp_doclayoutv2"):
        assert category_map(fam) == g["label_to_category"][fam]
        assert category_map(fam, ["header", "footer", "number"]) == g["label_to_category_ignoring_header_footer_number"][fam]
    assert len(category_map("pp_doclayoutv2")) == 25
    dets = to_layout_dets([{"label": "table", "score": 0.87654, "coordinate": [1.5, 2.5, 30.0, 40.0]}], "pp_doclayoutv2", True)
    assert dets[0]["category_id"] == 5 and dets[0]["poly"] == [1.5, 2.5, 30.0, 2.5, 30.0, 40.0, 1.5, 40.0] and dets[0]["score"] == 0.877


def test_expand_formula_crop_matches_reference(golden_dir):
    """backend/utils/utils.py:189-243 on 40 seeded layouts (page clipping, neighbours on each side, expand_px 0..5)."""
    from rapiddoc_amd.layout_host import expand_formula_crop
    cases = json.loads((golden_dir / "formula_expand.json").read_text())
    assert len(cases) == 40
    for c in cases:
        dets = c["dets"]
        res = expand_formula_crop(dets[0], dets, tuple(c["image_hw"]), c["expand_px"])
        assert [float(v) for v in res["poly"]] == c["poly"]
        assert ("polygon_points" in res) == c["has_polygon_points"]



def test_analyze_crop_helpers():
    """Pure-host pieces of rapiddoc_amd.analyze: formula boxes moved into a region crop (get_adjusted_mfdetrec_res,
    utils/ocr_utils.py:320-342) and the integer box used for the white-out (normalize_to_int_bbox, utils/bbox_utils.py)."""
    from rapiddoc_amd.analyze import _formula_boxes_in_crop, _int_box
    from rapiddoc_amd.layout_host import crop_geometry
    region = {"poly": [100, 200, 400, 200, 400, 260, 100, 260]}
    useful = crop_geometry(region, 50, 50)
    assert useful == [50, 50, 100, 200, 400, 260, 400, 160]
    formulas = [{"bbox": [150, 210, 200, 240]},      # inside
                {"bbox": [0, 0, 40, 40]},            # left of / above the crop: x1 = -10 < 0 -> dropped
                {"bbox": [380, 250, 520, 300]},      # sticks out on the right: kept, clipped later by _int_box
                {"bbox": [460, 210, 500, 240]}]      # x0 = 410 > new_width 400 -> dropped
    got = _formula_boxes_in_crop(formulas, useful)
    assert got == [[100, 60, 150, 90], [330, 100, 470, 150]]
    assert _int_box(got[1], 160, 400) == [330, 100, 400, 150]
    assert _int_box([10.2, 5.7, 10.9, 5.9], 100, 100) == [10, 5, 11, 6]
    assert _int_box([120, 5, 130, 9], 100, 100) is None
