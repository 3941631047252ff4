"""ORACLE (test infrastructure): independent numpy/scipy restatement of the DB post-process
(rapidocr DBPostProcess, box_type "quad", score_mode "fast", as configured in
rapid_doc/model/ocr/ocr_patch.py:141-154,223-241).  PARITY UNPINNED against the reference: rapidocr, cv2 and
pyclipper are not installed and the reference ships no vectors for this step; this file pins the C++
implementation (csrc/db_postprocess.cpp) to a second, differently-written statement of the same algorithm."""
from __future__ import annotations

import numpy as np
from scipy import ndimage
from scipy.spatial import ConvexHull, QhullError


def _min_area_rect(pts: np.ndarray):
    pts = np.unique(pts.astype(np.float64), axis=0)
    if len(pts) < 3:
        return None
    try:
        hull = pts[ConvexHull(pts).vertices]
    except QhullError:
        return None
    best = None
    for i in range(len(hull)):
        e = hull[(i + 1) % len(hull)] - hull[i]
        e = e / np.linalg.norm(e)
        rot = np.array([[e[0], e[1]], [-e[1], e[0]]])
        q = (hull - hull[i]) @ rot.T
        mn, mx = q.min(0), q.max(0)
        area = (mx[0] - mn[0]) * (mx[1] - mn[1])
        if best is None or area < best[0] - 1e-12:
            corners = np.array([[mn[0], mn[1]], [mx[0], mn[1]], [mx[0], mx[1]], [mn[0], mx[1]]]) @ rot + hull[i]
            best = (area, corners, mx[0] - mn[0], mx[1] - mn[1])
    return best[1], best[2], best[3]


def _order(box):
    p = sorted(box.tolist(), key=lambda t: t[0])
    (a, b), (c, d) = (p[0], p[1]), (p[2], p[3])
    tl, bl = (a, b) if b[1] > a[1] else (b, a)
    tr, br = (c, d) if d[1] > c[1] else (d, c)
    return np.array([tl, tr, br, bl])


def _score(pred, box):
    h, w = pred.shape
    x0 = int(np.clip(np.floor(box[:, 0].min()), 0, w - 1)); x1 = int(np.clip(np.ceil(box[:, 0].max()), 0, w - 1))
    y0 = int(np.clip(np.floor(box[:, 1].min()), 0, h - 1)); y1 = int(np.clip(np.ceil(box[:, 1].max()), 0, h - 1))
    q = (box - [x0, y0]).astype(np.int64)
    ys, xs = np.mgrid[y0:y1 + 1, x0:x1 + 1]
    px, py = xs - x0, ys - y0
    area2 = sum(q[i, 0] * q[(i + 1) % 4, 1] - q[(i + 1) % 4, 0] * q[i, 1] for i in range(4))
    sgn = 1 if area2 >= 0 else -1
    inside = np.ones_like(px, dtype=bool)
    for i in range(4):
        ex, ey = q[(i + 1) % 4] - q[i]
        inside &= (ex * (py - q[i, 1]) - ey * (px - q[i, 0])) * sgn >= 0
    return float(pred[y0:y1 + 1, x0:x1 + 1][inside].mean()) if inside.any() else 0.0


def db_postprocess(pred: np.ndarray, src_hw, thresh=0.3, box_thresh=0.5, unclip_ratio=1.6, use_dilation=True, min_size=3, max_candidates=1000):
    H, W = pred.shape
    src_h, src_w = src_hw
    bm = pred > thresh
    if use_dilation:
        d = bm.copy()
        d[:, 1:] |= bm[:, :-1]
        d[1:, :] |= bm[:-1, :]
        d[1:, 1:] |= bm[:-1, :-1]
        bm = d
    # contours (cv2.findContours RETR_LIST): one outer border per 8-connected region, one hole border per 4-connected
    # background component that does not reach the image frame (= the region pixels 4-adjacent to that hole)
    lab, n = ndimage.label(bm, structure=np.ones((3, 3)))
    point_sets = []
    for sl, k in zip(ndimage.find_objects(lab), range(1, n + 1)):
        ys, xs = np.nonzero(lab[sl] == k)
        pts = np.stack([xs + sl[1].start, ys + sl[0].start], 1)
        first = pts[np.lexsort((pts[:, 0], pts[:, 1]))][0]
        point_sets.append(((int(first[1]), int(first[0])), pts))
    cross = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]], bool)
    blab, nb = ndimage.label(~bm, structure=cross)
    frame = set(np.unique(np.concatenate([blab[0], blab[-1], blab[:, 0], blab[:, -1]]))) - {0}
    for k in range(1, nb + 1):
        if k in frame:
            continue
        hole = blab == k
        ring = ndimage.binary_dilation(hole, structure=cross) & bm
        ys, xs = np.nonzero(ring)
        hy, hx = np.nonzero(hole)
        o = np.lexsort((hx, hy))[0]
        point_sets.append(((int(hy[o]), int(hx[o]) - 1), np.stack([xs, ys], 1)))
    point_sets.sort(key=lambda t: t[0])            # raster order of the contours' start pixels
    point_sets = point_sets[:max_candidates]       # DBPostProcess looks at the first max_candidates contours only
    out = []
    for start, pts in point_sets:
        r = _min_area_rect(pts)
        if r is None:
            continue
        corners, w, h = r
        if min(w, h) < min_size:
            continue
        box = _order(corners)
        score = _score(pred, box)
        if box_thresh > score:
            continue
        dist = w * h * unclip_ratio / (2 * (w + h))
        ib = np.trunc(box)
        r2 = _min_area_rect(ib)
        if r2 is None:
            continue
        c2, w2, h2 = r2
        cen = c2.mean(0)
        u = (c2[1] - c2[0]) / np.linalg.norm(c2[1] - c2[0]); v = (c2[3] - c2[0]) / np.linalg.norm(c2[3] - c2[0])
        hu, hv = w2 / 2 + dist, h2 / 2 + dist
        if min(2 * hu, 2 * hv) < min_size + 2:
            continue
        ex = _order(np.array([cen - hu * u - hv * v, cen + hu * u - hv * v, cen + hu * u + hv * v, cen - hu * u + hv * v]))
        bx = np.clip(np.round(ex[:, 0] / W * src_w), 0, src_w); by = np.clip(np.round(ex[:, 1] / H * src_h), 0, src_h)
        q = np.stack([bx, by], 1)
        xs_ = q[np.argsort(q[:, 0], kind="stable")]
        l, rr = xs_[:2], xs_[2:]
        l = l[np.argsort(l[:, 1], kind="stable")]; rr = rr[np.argsort(rr[:, 1], kind="stable")]
        box4 = np.array([l[0], rr[0], rr[1], l[1]])
        box4[:, 0] = np.clip(box4[:, 0], 0, src_w - 1); box4[:, 1] = np.clip(box4[:, 1], 0, src_h - 1)
        if int(np.linalg.norm(box4[0] - box4[1])) <= 3 or int(np.linalg.norm(box4[0] - box4[3])) <= 3:
            continue
        out.append((box4.astype(np.int32), score))
    return [o[0] for o in out], [o[1] for o in out]
