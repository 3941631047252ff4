"""Polygon branch of the layout post-process against vectors minted by the reference's own post_process.py
(tests/golden/make_golden_polygon.py: the reference code unmodified, with this repo's C primitives standing in for cv2 / shapely).
Pinned here: the masks following their boxes through threshold / NMS / filters / merge / sort, the box -> mask-grid arithmetic,
mask -> polygon, the vertex selection, the quad conversion, the `auto` decisions (quad vs polygon vs rectangle, the look at the
previous polygon), dropped boxes, the output dicts.  Not pinned here: the primitives' own arithmetic (tests/test_polygon_ops.py)."""
import json
import sys
from pathlib import Path

import numpy as np
import pytest

from rapiddoc_amd import _lib

pytestmark = pytest.mark.skipif(not _lib.LIB_PATH.exists(), reason="librapiddoc_mi355.so not built (host C++ primitives)")
GOLD = Path(__file__).parent / "golden"
sys.path.insert(0, str(GOLD))
import polygon_masks as PM  # noqa: E402

G = json.loads((GOLD / "layout_polygon.json").read_text())


def _same_points(mine, ref):
    if ref is None:
        return mine is None
    return mine is not None and [[float(x), float(y)] for x, y in mine] == ref


@pytest.mark.parametrize("case", G["custom_vertices"], ids=lambda c: f"{c['seed']}-{int(c['max_allowed_dist'])}")
def test_vertex_selection_matches_the_reference(case):
    from rapiddoc_amd.layout_polygon import custom_vertices
    poly = PM.random_polygon(case["seed"])
    assert len(poly) == case["n_in"]
    assert _same_points(custom_vertices(poly, case["max_allowed_dist"]), case["out"])


@pytest.mark.parametrize("case", G["post"], ids=lambda c: f"{c['seed']}-{c['mode']}-{c['ncol']}col")
def test_postprocess_with_masks_matches_the_reference(case):
    from rapiddoc_amd.layout_host import LayoutPostProcess
    thr = {int(k): v for k, v in case["thresh"].items()} if isinstance(case["thresh"], dict) else case["thresh"]
    merge = {int(k): v for k, v in case["merge"].items()} if isinstance(case["merge"], dict) else case["merge"]
    boxes, masks, _kinds = PM.make_case(case["seed"], case["n"], case["ncol"], len(case["labels"]))
    pp = LayoutPostProcess(case["labels"], thr, 0.5, layout_merge_bboxes_mode=merge, layout_unclip_ratio=case["unclip"],
                           scale_size=(PM.INPUT, PM.INPUT))
    out = pp(boxes, [PM.PAGE_W, PM.PAGE_H], masks, case["mode"])
    out = [] if isinstance(out, np.ndarray) else out
    assert len(out) == len(case["out"])
    for i, (a, b) in enumerate(zip(out, case["out"])):
        assert (a["cls_id"], a["label"], a["score"], a["coordinate"], a["order"]) == \
               (b["cls_id"], b["label"], b["score"], b["coordinate"], b["order"]), i
        assert ("polygon_points" in a) == b["has_polygon"], i
        assert _same_points(a.get("polygon_points"), b["polygon_points"]), i


def test_fixture_covers_every_outcome_of_the_auto_mode():
    auto = [c for c in G["post"] if c["mode"] == "auto"]
    sizes = {len(r["polygon_points"]) for c in auto for r in c["out"]}
    assert 4 in sizes and max(sizes) > 8                       # rectangles / quads and real polygons
    n_in = sum(c["n"] for c in auto)
    assert sum(len(c["out"]) for c in auto) < n_in              # boxes were dropped on the way
    rect = [c for c in G["post"] if c["mode"] == "rect"][0]
    assert all(not r["has_polygon"] for r in rect["out"])       # rect mode: masks ignored, no key at all


def test_masks_without_scale_size_fail_loudly():
    from rapiddoc_amd.layout_host import LayoutPostProcess
    boxes, masks, _ = PM.make_case(1, 4, 6, 3)
    with pytest.raises(ValueError):
        LayoutPostProcess(["a", "b", "c"], 0.1)(boxes, [PM.PAGE_W, PM.PAGE_H], masks, "auto")
