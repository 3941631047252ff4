"""GPU: the table stage (analyze.TableOcr) on the real det / rec engines.  In a file of its own that sorts after the other GPU tests."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("word_box", [False, True])
def test_table_stage_on_the_gpu_engines(golden_dir, word_box):
    """analyze.TableOcr through the real det / rec engines: a `predict`-shaped table model receives the table crop, one OCR line per
    synthetic text line inside the table (boxes in crop coordinates, strings, float scores) and the formula box inside it.  `word_box`
    (the reference's default, analyze_utils.py:308): one entry per word / CJK character instead - built from the device's kept time steps
    and probabilities (rd_ctc_collapse_lines) by rapiddoc_amd/word_boxes.py; every word box lies inside its line's box."""
    import json
    from rapiddoc_amd import weights as W
    from rapiddoc_amd.analyze import PageAnalyzer
    from rapiddoc_amd.layout_model import LayoutModel
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline
    maps = json.loads((golden_dir / "layout_category_maps.json").read_text())
    labels = list(maps["label_to_category"]["pp_doclayoutv2"])
    TAB = (80, 500, 1150, 1000)

    class Session:
        characters = labels

        def __call__(self, x, sf):
            rows = [[labels.index("table"), 0.9, *TAB, 0], [labels.index("inline_formula"), 0.9, 300, 600, 420, 640, 1]] * x.shape[0]
            return [np.asarray(rows, np.float32), np.full(x.shape[0], 2, np.int32)]

    seen = []

    class Table:
        def predict(self, image, ocr_result, fill_image_res, mfd_res, skip_text_in_image, use_img2table, skip_table_orientation=False):
            seen.append((image, ocr_result, fill_image_res, mfd_res, (skip_text_in_image, use_img2table, skip_table_orientation)))
            return "<html><table><tr><td>%d</td></tr></table></html>" % (len(ocr_result[0]) if ocr_result else 0)

    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, n_rec_streams=2)            # (strict rec mode, the default: word boxes need its per-line widths)
    class Formula:                       # (without a formula model the driver drops inline formulas, like the reference)
        def batch_predict(self, imgs, batch_size=16):
            return ["x" for _ in imgs]

    an = PageAnalyzer(LayoutModel(Session(), "pp_doclayoutv3"), pipe, formula_model=Formula(), table_model=Table(), table_use_word_box=word_box)
    pages_np, boxes = synth_batch(0, 1)
    lines = [lb for lb in np.asarray(boxes[0], dtype=np.float64).reshape(-1, 4)
             if lb[0] >= TAB[0] and lb[2] <= TAB[2] and lb[1] >= TAB[1] and lb[3] <= TAB[3]
             and not (lb[2] > 300 and lb[0] < 420 and lb[3] > 600 and lb[1] < 640)]          # lines clear of the formula box

    def table_maps(p, rect, dhw):
        (x0, y0, x1, y1), (dh, dw) = rect, dhw
        m = torch.zeros((1, 1, dh, dw), dtype=torch.float32)
        for lb in lines:
            d = 0.32 * min(lb[2] - lb[0], lb[3] - lb[1])
            m[0, 0, int(round((lb[1] - y0 + d) * dh / (y1 - y0))):int(round((lb[3] - y0 - d) * dh / (y1 - y0))),
              int(round((lb[0] - x0 + d) * dw / (x1 - x0))):int(round((lb[2] - x0 - d) * dw / (x1 - x0)))] = 0.95
        return m.cuda()

    out = an(torch.from_numpy(pages_np).cuda(), table_det_maps_fn=table_maps, page_scales=[2.0])[0]
    assert len(seen) == 1 and len(lines) > 3
    image, ocr_result, fill, mfd, flags = seen[0]
    assert image.shape == (TAB[3] - TAB[1], TAB[2] - TAB[0], 3) and fill == [] and flags == (True, False, True)
    assert mfd == [{"bbox": [300 - TAB[0], 600 - TAB[1], 420 - TAB[0], 640 - TAB[1]], "latex": "x"}]
    bxs, texts, scores = ocr_result
    assert len(bxs) == len(texts) == len(scores)
    assert all(np.asarray(b).shape == (4, 2) for b in bxs) and all(isinstance(t, str) for t in texts) and all(0.0 <= float(s) <= 1.0 for s in scores)
    if not word_box:
        assert len(bxs) == len(lines)
    else:
        # random weights read CJK strings: one box per character, the characters of a line share its height and stay inside the table image
        assert len(bxs) >= len(lines) and all(len(t) == 1 for t in texts)
        b = np.asarray(bxs, dtype=np.float64)
        assert b[..., 0].min() >= 0 and b[..., 0].max() <= image.shape[1] and b[..., 1].min() >= 0 and b[..., 1].max() <= image.shape[0]
        assert (b[:, 1, 0] >= b[:, 0, 0]).all() and (b[:, 2, 1] >= b[:, 1, 1]).all()
        rows = {}
        for q in b:
            rows.setdefault((int(q[0, 1]), int(q[2, 1])), []).append(q)
        assert len(rows) <= len(lines) + 2                      # the boxes group into (about) one row per text line
    table = [d for d in out if d["category_id"] == 5][0]
    assert table["html"] == "<table><tr><td>%d</td></tr></table>" % len(texts)
    assert table["formula_boxes"] == [[150, 300, 210, 320]]


def test_doclayout_yolo_letterbox_on_the_gpu():
    """LayoutModel.preprocess_letterbox (linear resize kernel + 114 padding + channel flip + float64 / 255) against
    oracle/cv2_ops.resize_linear_u8 and numpy, bit for bit; then a whole batch_predict with the stand-in session."""
    from oracle import cv2_ops as CV
    from rapiddoc_amd.layout_model import LayoutModel, SyntheticYoloSession, _YOLO_LABELS
    from rapiddoc_amd.pages import synth_batch
    pages_np, _ = synth_batch(5, 2)
    model = LayoutModel(SyntheticYoloSession(_YOLO_LABELS[:10], 30), "doclayout_docstructbench")
    for page in (pages_np[0], np.ascontiguousarray(pages_np[1][:600, :1100])):
        h, w = page.shape[:2]
        new_w, new_h, left, top = LayoutModel.letterbox_geometry(h, w)
        want = np.full((1024, 1024, 3), 114, np.uint8)
        want[top:top + new_h, left:left + new_w] = CV.resize_linear_u8(page, (new_h, new_w))
        want = (want[None][..., ::-1].transpose(0, 3, 1, 2) / 255).astype(np.float32)
        got = model.preprocess_letterbox(page).cpu().numpy()
        assert got.shape == (1, 3, 1024, 1024) and got.dtype == np.float32
        assert np.array_equal(got, want)
    out = model.batch_predict([pages_np[0], pages_np[1]], 2)
    assert len(out) == 2 and all(len(p) > 0 for p in out)
    assert all(0 <= d["poly"][0] <= d["poly"][4] <= 1191 and 0 <= d["poly"][1] <= d["poly"][5] <= 1684 for p in out for d in p)
