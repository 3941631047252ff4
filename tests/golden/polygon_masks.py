"""Seeded inputs of the polygon-branch fixtures: detector boxes and instance masks drawn with numpy (no cv2).  Shared by
tests/golden/make_golden_polygon.py (which runs the reference on them) and tests/test_layout_polygon.py (which runs this repo on
them), so that the committed fixture holds only the parameters and the expected outputs, not megabytes of masks."""
import numpy as np

PAGE_W, PAGE_H, INPUT, GRID = 1191, 1684, 800, 200          # page pixels, detector input size, mask grid (input / 4)
KINDS = ("full", "quad", "ellipse", "ell", "empty", "speck", "two", "wedge")


def _shape(kind, gh, gw, rng):
    """u8 [gh, gw] drawing of one instance inside its box's patch of the mask grid."""
    yy, xx = np.mgrid[:gh, :gw]
    u, v = (xx + 0.5) / gw, (yy + 0.5) / gh                 # 0..1 inside the patch
    if kind == "full":
        m = np.ones((gh, gw), bool)
    elif kind == "quad":                                     # a slanted parallelogram
        t = rng.uniform(-0.25, 0.25)
        m = (u - t * (v - 0.5) > 0.12) & (u - t * (v - 0.5) < 0.88) & (v > 0.1) & (v < 0.9)
    elif kind == "ellipse":
        m = ((u - 0.5) / 0.45) ** 2 + ((v - 0.5) / 0.42) ** 2 <= 1
    elif kind == "ell":                                      # an L
        a, b = rng.uniform(0.3, 0.5), rng.uniform(0.3, 0.5)
        m = ((u < a) | (v > 1 - b)) & (u > 0.04) & (u < 0.96) & (v > 0.04) & (v < 0.96)
    elif kind == "empty":
        m = np.zeros((gh, gw), bool)
    elif kind == "speck":
        m = np.zeros((gh, gw), bool)
        m[gh // 2, gw // 2] = True
    elif kind == "two":                                      # two blobs of different area: the larger contour wins
        m = ((u < 0.3) & (v < 0.4)) | ((u > 0.45) & (v > 0.2))
    else:                                                    # "wedge": a triangle with a sharp tip
        m = (v > 0.1) & (v < 0.9) & (u > 0.05) & (u < 0.05 + 0.9 * (1 - np.abs(v - 0.5) / 0.4))
    return m.astype(np.uint8)


def make_case(seed: int, n: int, ncol: int, n_classes: int):
    """-> boxes float32 [n, ncol] (cls, score, x0, y0, x1, y1[, order[, 0]]), masks u8 [n, GRID, GRID], kinds."""
    rng = np.random.default_rng(seed)
    cls = rng.integers(0, n_classes, n).astype(np.float32)
    score = rng.uniform(0.05, 0.99, n)
    x0 = rng.uniform(0, PAGE_W - 240, n)
    y0 = rng.uniform(0, PAGE_H - 200, n)
    bw = rng.uniform(40, 600, n)
    bh = rng.uniform(24, 420, n)
    x1, y1 = np.minimum(x0 + bw, PAGE_W - 1), np.minimum(y0 + bh, PAGE_H - 1)
    cols = [cls, score, x0, y0, x1, y1]
    if ncol >= 7:
        cols.append(rng.permutation(n).astype(np.float32))
    if ncol == 8:
        cols.append(np.zeros(n))
    boxes = np.stack(cols, 1).astype(np.float32)
    sx, sy = INPUT / PAGE_W / 4, INPUT / PAGE_H / 4
    masks = np.zeros((n, GRID, GRID), np.uint8)
    kinds = []
    for i in range(n):
        kind = KINDS[int(rng.integers(0, len(KINDS)))]
        kinds.append(kind)
        gx0, gx1 = int(np.floor(boxes[i, 2] * sx)), int(np.ceil(boxes[i, 4] * sx))
        gy0, gy1 = int(np.floor(boxes[i, 3] * sy)), int(np.ceil(boxes[i, 5] * sy))
        gx1, gy1 = min(max(gx1, gx0 + 1), GRID), min(max(gy1, gy0 + 1), GRID)
        masks[i, gy0:gy1, gx0:gx1] = _shape(kind, gy1 - gy0, gx1 - gx0, rng)
    return boxes, masks, kinds


def random_polygon(seed: int):
    """An integer polygon as approxPolyDP would hand it to the vertex selection: a star-shaped ring with random radii, some
    vertices pulled inwards (concave runs), plus - for some seeds - a 45-degree tip."""
    rng = np.random.default_rng(seed)
    n = int(rng.integers(4, 18))
    ang = np.sort(rng.uniform(0, 2 * np.pi, n))
    rad = rng.uniform(60, 200, n)
    for k in range(0, n - 1, 5):                             # pairs of neighbouring dents
        rad[k:k + 2] *= rng.uniform(0.3, 0.7)
    pts = np.stack([300 + rad * np.cos(ang), 300 + rad * np.sin(ang)], 1)
    pts = np.round(pts).astype(np.int32)
    if seed % 3 == 0:
        pts = np.concatenate([pts, np.int32([[300, 700], [200, 600]])])[::-1].copy()
    keep = np.ones(len(pts), bool)                           # no repeated neighbours (a zero-length edge has no direction)
    for i in range(len(pts)):
        if (pts[i] == pts[(i + 1) % len(pts)]).all():
            keep[i] = False
    return pts[keep]
