"""CPU: DB post-process (C++ host code behind the C-ABI) vs analytic known answers and vs the independent
oracle restatement (oracle/dbpost.py).  Parity with rapidocr/OpenCV itself is unpinned (see the file headers)."""
import numpy as np
import pytest

from oracle import dbpost as OD
from rapiddoc_amd import ocr_host as H


@pytest.fixture(scope="module", autouse=True)
def _built():
    from rapiddoc_amd import build as rd_build
    rd_build.build(verbose=False)


def test_axis_aligned_rectangle_known_answer():
    pred = np.full((320, 640), 0.05, np.float32)
    pred[50:80, 100:300] = 0.9                       # text blob: x 100..299, y 50..79
    (boxes, scores), = H.db_postprocess(pred[None], [(320, 640)], box_thresh=0.5, unclip_ratio=1.8)
    assert len(boxes) == 1
    # dilation adds one pixel to the right/bottom -> region points x 100..300, y 50..80 -> rect 200 x 30
    d = 200 * 30 * 1.8 / (2 * 230)
    exp = np.array([[100 - d, 50 - d], [300 + d, 50 - d], [300 + d, 80 + d], [100 - d, 80 + d]])
    assert np.abs(boxes[0] - np.round(exp)).max() <= 1
    inner = 0.9 * 200 * 30 + 0.05 * (201 * 31 - 200 * 30)
    assert abs(scores[0] - inner / (201 * 31)) < 1e-6


def test_scaling_to_source_and_threshold():
    pred = np.zeros((2, 160, 320), np.float32)
    pred[0, 40:60, 60:200] = 0.8
    pred[1, 40:60, 60:200] = 0.45                   # above thresh .3 but below box_thresh .5 -> dropped
    r0, r1 = H.db_postprocess(pred, [(320, 640), (320, 640)], box_thresh=0.5, unclip_ratio=1.6)
    assert len(r0[0]) == 1 and len(r1[0]) == 0
    b = r0[0][0]
    assert b[0, 0] < 120 and b[2, 0] > 400 and b[0, 1] < 80 and b[2, 1] > 120  # coordinates doubled


def _random_map(rng, h, w, n):
    pred = rng.uniform(0.0, 0.2, (h, w)).astype(np.float32)
    yy, xx = np.mgrid[0:h, 0:w]
    for _ in range(n):
        cx, cy = rng.uniform(40, w - 40), rng.uniform(20, h - 20)
        bw, bh = rng.uniform(30, 120), rng.uniform(8, 20)
        ang = rng.uniform(-0.5, 0.5)
        u = (xx - cx) * np.cos(ang) + (yy - cy) * np.sin(ang)
        v = -(xx - cx) * np.sin(ang) + (yy - cy) * np.cos(ang)
        pred[(np.abs(u) < bw / 2) & (np.abs(v) < bh / 2)] = rng.uniform(0.6, 0.95)
    return pred


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_matches_independent_oracle_on_rotated_blobs(seed):
    rng = np.random.default_rng(seed)
    pred = _random_map(rng, 192, 384, 7)
    (boxes, scores), = H.db_postprocess(pred[None], [(384, 768)], box_thresh=0.5, unclip_ratio=1.6)
    oboxes, oscores = OD.db_postprocess(pred, (384, 768), box_thresh=0.5, unclip_ratio=1.6)
    assert len(boxes) == len(oboxes) > 0
    for b, ob, s, os_ in zip(boxes, oboxes, scores, oscores):
        assert np.abs(b - ob).max() <= 1
        assert abs(s - os_) < 1e-5


def test_empty_and_full_maps():
    z = np.zeros((1, 64, 64), np.float32)
    assert H.db_postprocess(z, [(64, 64)])[0][0].shape == (0, 4, 2)
    o = np.ones((1, 64, 96), np.float32)
    (b, s), = H.db_postprocess(o, [(64, 96)])
    assert len(b) == 1 and abs(s[0] - 1.0) < 1e-6


def test_sorted_boxes_reading_order():
    mk = lambda x, y: np.array([[x, y], [x + 50, y], [x + 50, y + 20], [x, y + 20]])
    boxes = [mk(300, 12), mk(10, 100), mk(20, 8), mk(200, 104)]
    out = H.sorted_boxes(boxes)
    assert [int(b[0][0]) for b in out] == [20, 300, 10, 200]


def test_hole_borders_are_contours_like_retr_list():
    """cv2.findContours(RETR_LIST) hands the hole borders of a region back as contours of their own, so the
    reference scores them as candidates.  A 60 x 200 text blob with a 9 x 9 hole gives TWO candidates: the blob's outer border,
    and the ring of blob pixels around the hole - a 10 x 10 box whose mean PROBABILITY (81 pixels of the 9 x 9 probability hole at 0.05, 19 at
    0.9) is 0.2115: kept at box_thresh 0.2, dropped at 0.3 / 0.5 - which is why real pages rarely show them.  A hole that reaches the
    image frame is background, not a hole; a 2 x 2 hole's ring is too small (its min-area rectangle has a side < 3)."""
    pred = np.full((200, 400), 0.05, np.float32)
    pred[40:100, 100:300] = 0.9
    pred[60:69, 150:159] = 0.05                       # 9 x 9 hole (the 2 x 2 dilation eats one row / column of it: 8 x 8 remain)
    pred[80:82, 250:252] = 0.05                       # 2 x 2 hole: filled by the dilation
    pred[120:180, 0:120] = 0.9
    pred[140:160, 0:30] = 0.05                        # a notch that reaches the left image border: not a hole
    for box_thresh, expect in ((0.2, 3), (0.3, 2), (0.5, 2)):
        (boxes, scores), = H.db_postprocess(pred[None], [(200, 400)], box_thresh=box_thresh, unclip_ratio=1.8)
        oboxes, oscores = OD.db_postprocess(pred, (200, 400), box_thresh=box_thresh, unclip_ratio=1.8)
        assert len(boxes) == len(oboxes) == expect, (box_thresh, len(boxes), len(oboxes))
        for b, ob, s, os_ in zip(boxes, oboxes, scores, oscores):
            assert np.abs(b - ob).max() <= 1 and abs(s - os_) < 1e-6
    (boxes, scores), = H.db_postprocess(pred[None], [(200, 400)], box_thresh=0.2, unclip_ratio=1.8)
    # raster order of the contours' start pixels: blob (40, 100), hole ring (61, 150), second blob (120, 0)
    hole = boxes[1]
    assert 140 <= hole[:, 0].min() and hole[:, 0].max() <= 170 and 50 <= hole[:, 1].min() and hole[:, 1].max() <= 80
    ring_mean = (0.9 * 19 + 0.05 * 81) / 100.0
    assert abs(scores[1] - ring_mean) < 1e-6


@pytest.mark.parametrize("kind", ["lines", "blobs", "holes", "empty", "full", "speckle"])
def test_host_path_against_the_oracle_on_the_maps_of_the_device_test(kind):
    """The six map kinds of tests/test_gpu_image_ops.py::test_device_db_postprocess_equals_host_path (where the device chain must equal
    the host C++ box for box), here host C++ vs oracle/dbpost.py on the CPU: same boxes, same order, same count (also past
    max_candidates); coordinates equal except for the float32 / float64 effects that test's comment states (<= 3 px, <= 0.5 %)."""
    import importlib.util
    from pathlib import Path
    spec = importlib.util.spec_from_file_location("gpu_image_ops_maps", Path(__file__).with_name("test_gpu_image_ops.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    m, box_thresh, B = mod._db_test_map(kind)
    hw = [(640, 896)] * B
    host = H.db_postprocess(m, hw, thresh=0.3, box_thresh=box_thresh, unclip_ratio=1.8)
    n_coord = n_diff = 0
    for b in range(B):
        ob, osc = OD.db_postprocess(m[b], hw[b], thresh=0.3, box_thresh=box_thresh, unclip_ratio=1.8)
        assert len(ob) == len(host[b][0])
        for x, y, so, sh in zip(ob, host[b][0], osc, host[b][1]):
            d = np.abs(np.asarray(x) - y)
            assert d.max() <= 3 and abs(so - sh) < 1e-5
            n_coord += d.size
            n_diff += int((d > 0).sum())
    assert n_diff <= max(2, 0.005 * n_coord)
