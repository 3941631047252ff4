"""ORACLE (test infrastructure, not product code) - PARITY UNPINNED.

numpy restatement of the 8-bit OpenCV image operations on RapidDoc's page hot path.  `cv2`, `rapidocr` and their
wheels are absent from this container and the reference holds no pixel vectors for them, so this file
cannot be checked against the real library here: it restates the public OpenCV 4.x algorithms (modules/imgproc/src/
resize.cpp, imgwarp.cpp) from their published definitions, and the HIP kernels are tested for bit-equality against IT.
Call sites in the reference that these functions stand in for:

  resize_cubic_u8 ............ cv2.resize(img, (S, S), interpolation=2)      pp_doclayout/pre_process.py:35
  resize_linear_u8 ........... cv2.resize(img, (w, h))                        rapidocr DetPreProcess / resize_norm_img
                                                                              (called from rapid_ocr.py:517-518, 436-440)
  warp_perspective_cubic_u8 .. cv2.warpPerspective(img, M, (w, h), borderMode=BORDER_REPLICATE, flags=INTER_CUBIC)
                                                                              utils/ocr_utils.py:523-529
  get_rotate_crop_image ...... utils/ocr_utils.py:494-536
  resize_norm_img ............ rapidocr TextRecognizer.resize_norm_img (rec_image_shape [3, 48, 320])
  layout_preprocess .......... PPPreProcess.__call__                          pp_doclayout/pre_process.py:22-42

Fixed-point conventions restated (OpenCV): resize coefficients are 11-bit (`INTER_RESIZE_COEF_BITS`), computed in float32
and rounded half-to-even to int16; the horizontal pass keeps int32 rows, the vertical pass of the cubic kernel rounds
(v + 2^21) >> 22, that of the linear kernel uses ((b0 (S0 >> 4)) >> 16) + ((b1 (S1 >> 4)) >> 16) + 2) >> 2; remap-based warps
quantise source positions to 1/32 pixel (`INTER_BITS` = 5) and use a 32 x 32 table of 4 x 4 weights in 15-bit fixed point
whose sum is forced to 2^15 by adjusting the largest / smallest weight of its lower-right 2 x 2 quadrant.
"""
from __future__ import annotations

from functools import lru_cache
from typing import Tuple

import numpy as np

COEF_BITS = 11
COEF_SCALE = 1 << COEF_BITS
REMAP_BITS = 15
REMAP_SCALE = 1 << REMAP_BITS
TAB = 32          # INTER_TAB_SIZE


def _round_half_even_i(v: np.ndarray) -> np.ndarray:
    return np.rint(v).astype(np.int64)        # np.rint rounds half to even like cvRound (lrint)


def cubic_coeffs_f32(x: np.ndarray) -> np.ndarray:
    """interpolateCubic (imgproc: A = -0.75), float32 arithmetic, shape [..., 4]."""
    x = x.astype(np.float32)
    A = np.float32(-0.75)
    one, two, three = np.float32(1), np.float32(2), np.float32(3)
    c0 = ((A * (x + one) - np.float32(5) * A) * (x + one) + np.float32(8) * A) * (x + one) - np.float32(4) * A
    c1 = ((A + two) * x - (A + three)) * x * x + one
    c2 = ((A + two) * (one - x) - (A + three)) * (one - x) * (one - x) + one
    c3 = one - c0 - c1 - c2
    return np.stack([c0, c1, c2, c3], axis=-1).astype(np.float32)


def _axis_cubic(dst: int, src: int) -> Tuple[np.ndarray, np.ndarray]:
    """Per destination index: the 4 clamped source indices and the int16 coefficients (resize.cpp, cubic branch)."""
    scale = 1.0 / (float(dst) / float(src))                      # double, as cv::resize computes scale_x
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    co = cubic_coeffs_f32(f)
    ico = np.clip(_round_half_even_i(co * np.float32(COEF_SCALE)), -32768, 32767)
    idx = np.clip(s[:, None] + np.arange(-1, 3)[None, :], 0, src - 1)
    return idx, ico


def resize_cubic_u8(img: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (ow, oh), interpolation=cv2.INTER_CUBIC) for uint8 HWC."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    oh, ow = out_hw
    xi, xa = _axis_cubic(ow, W)
    yi, ya = _axis_cubic(oh, H)
    src = img.astype(np.int64)
    rows = np.zeros((H, ow, img.shape[2]), np.int64)             # horizontal pass: int rows at scale 2^11
    for k in range(4):
        rows += src[:, xi[:, k], :] * xa[None, :, k, None]
    out = np.zeros((oh, ow, img.shape[2]), np.int64)
    for k in range(4):
        out += rows[yi[:, k]] * ya[:, k, None, None]
    out = (out + (1 << (2 * COEF_BITS - 1))) >> (2 * COEF_BITS)
    return np.clip(out, 0, 255).astype(np.uint8)


def _axis_linear(dst: int, src: int) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    scale = 1.0 / (float(dst) / float(src))
    d = np.arange(dst, dtype=np.float64)
    f = ((d + 0.5) * scale - 0.5).astype(np.float32)
    s = np.floor(f).astype(np.int64)
    f = (f - s.astype(np.float32)).astype(np.float32)
    lo = s < 0
    f = np.where(lo, np.float32(0), f)
    s = np.where(lo, 0, s)
    hi = s >= src - 1
    f = np.where(hi, np.float32(0), f)
    s = np.where(hi, src - 1, s)
    a0 = np.clip(_round_half_even_i((np.float32(1) - f) * np.float32(COEF_SCALE)), -32768, 32767)
    a1 = np.clip(_round_half_even_i(f * np.float32(COEF_SCALE)), -32768, 32767)
    return s, np.minimum(s + 1, src - 1), np.stack([a0, a1], axis=1)


def resize_linear_u8(img: np.ndarray, out_hw: Tuple[int, int]) -> np.ndarray:
    """cv2.resize(img, (ow, oh)) (INTER_LINEAR, the default) for uint8 HWC."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    oh, ow = out_hw
    x0, x1, xa = _axis_linear(ow, W)
    y0, y1, ya = _axis_linear(oh, H)
    src = img.astype(np.int64)
    rows = src[:, x0, :] * xa[None, :, 0, None] + src[:, x1, :] * xa[None, :, 1, None]      # scale 2^11
    s0, s1 = rows[y0], rows[y1]
    b0, b1 = ya[:, 0, None, None], ya[:, 1, None, None]
    out = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2
    return np.clip(out, 0, 255).astype(np.uint8)


@lru_cache(maxsize=1)
def cubic_remap_table() -> np.ndarray:
    """initInterTab2D(INTER_CUBIC, fixpt=true): [32 * 32][4][4] int16 weights, each 4 x 4 block summing to 2^15."""
    t1 = cubic_coeffs_f32(np.arange(TAB, dtype=np.float32) * np.float32(1.0 / TAB))          # [32][4]
    tab = np.zeros((TAB * TAB, 4, 4), np.int64)
    for i in range(TAB):
        for j in range(TAB):
            v = (t1[i][:, None] * t1[j][None, :]).astype(np.float32)                          # rows = y taps, cols = x taps
            it = np.clip(_round_half_even_i(v * np.float32(REMAP_SCALE)), -32768, 32767)
            diff = int(it.sum()) - REMAP_SCALE
            if diff != 0:
                # initInterTab2D searches rows / columns ksize/2 .. ksize/2 + 1 (= 2, 3 for the 4-tap kernel), starting from
                # (2, 2).  (With the centre block an integer position - one weight of 32767 - would overflow int16.)
                mk, Mk = (2, 2), (2, 2)
                for k1 in (2, 3):
                    for k2 in (2, 3):
                        if it[k1, k2] < it[mk]:
                            mk = (k1, k2)
                        elif it[k1, k2] > it[Mk]:
                            Mk = (k1, k2)
                if diff < 0:
                    it[Mk] -= diff
                else:
                    it[mk] -= diff
            tab[i * TAB + j] = it
    return tab


def warp_perspective_cubic_u8(img: np.ndarray, M_dst_to_src: np.ndarray, out_wh: Tuple[int, int]) -> np.ndarray:
    """cv2.warpPerspective(..., flags=INTER_CUBIC, borderMode=BORDER_REPLICATE) for uint8 HWC, given the matrix that maps
    DESTINATION pixels to source coordinates (what OpenCV inverts `M` to)."""
    assert img.dtype == np.uint8 and img.ndim == 3
    H, W, _ = img.shape
    ow, oh = out_wh
    M = np.asarray(M_dst_to_src, dtype=np.float64).reshape(3, 3)
    x = np.arange(ow, dtype=np.float64)[None, :]
    y = np.arange(oh, dtype=np.float64)[:, None]
    Wd = M[2, 0] * x + M[2, 1] * y + M[2, 2]
    Wd = np.where(Wd != 0, TAB / np.where(Wd != 0, Wd, 1.0), 0.0)
    fx = np.clip((M[0, 0] * x + M[0, 1] * y + M[0, 2]) * Wd, -2147483648.0, 2147483647.0)
    fy = np.clip((M[1, 0] * x + M[1, 1] * y + M[1, 2]) * Wd, -2147483648.0, 2147483647.0)
    X, Y = _round_half_even_i(fx), _round_half_even_i(fy)
    sx, sy = X >> 5, Y >> 5
    al = (Y & (TAB - 1)) * TAB + (X & (TAB - 1))
    tab = cubic_remap_table()[al]                                                            # [oh][ow][4][4]
    src = img.astype(np.int64)
    acc = np.zeros((oh, ow, img.shape[2]), np.int64)
    for a in range(4):
        yy = np.clip(sy - 1 + a, 0, H - 1)
        for b in range(4):
            xx = np.clip(sx - 1 + b, 0, W - 1)
            acc += src[yy, xx] * tab[..., a, b, None]
    acc = (acc + (1 << (REMAP_BITS - 1))) >> REMAP_BITS
    return np.clip(acc, 0, 255).astype(np.uint8)


def perspective_dst_to_src(quad: np.ndarray) -> Tuple[np.ndarray, int, int]:
    """utils/ocr_utils.py:494-522: crop size from the quad's edge lengths, cv2.getPerspectiveTransform(quad -> rectangle);
    returned is its inverse (rectangle pixel -> page coordinates), which is what warpPerspective evaluates, plus (w, h)."""
    q = np.asarray(quad, dtype=np.float64).reshape(4, 2)
    cw = int(max(np.linalg.norm(q[0] - q[1]), np.linalg.norm(q[2] - q[3])))
    ch = int(max(np.linalg.norm(q[0] - q[3]), np.linalg.norm(q[1] - q[2])))
    cw, ch = max(cw, 1), max(ch, 1)
    dst = np.array([[0, 0], [cw, 0], [cw, ch], [0, ch]], dtype=np.float64)
    A, b = [], []
    for (x, y), (u, v) in zip(dst, q):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.asarray(A), np.asarray(b))
    return np.append(h, 1.0).reshape(3, 3), cw, ch


def get_rotate_crop_image(img: np.ndarray, quad: np.ndarray) -> np.ndarray:
    """utils/ocr_utils.py:494-536: rectify the quad with a cubic warp, rotate tall crops by 90 degrees (np.rot90)."""
    M, cw, ch = perspective_dst_to_src(quad)
    crop = warp_perspective_cubic_u8(img, M, (cw, ch))
    if ch * 1.0 / cw >= 2:            # rotate_radio = 2 (ocr_utils.py:533-535)
        crop = np.rot90(crop)
    return crop


def resize_norm_img(crop: np.ndarray, max_wh_ratio: float, img_h: int = 48) -> np.ndarray:
    """rapidocr TextRecognizer.resize_norm_img: width = ceil(48 * w / h) capped at int(48 * max_wh_ratio), linear resize,
    /255, (x - 0.5) / 0.5, zero right-padding.  Returns [3][48][img_w] float32."""
    img_w = int(img_h * max_wh_ratio)
    h, w = crop.shape[:2]
    ratio = w / float(h)
    rw = img_w if int(np.ceil(img_h * ratio)) > img_w else int(np.ceil(img_h * ratio))
    r = resize_linear_u8(crop, (img_h, rw)).astype(np.float32).transpose(2, 0, 1) / 255.0
    r = (r - 0.5) / 0.5
    out = np.zeros((3, img_h, img_w), np.float32)
    out[:, :, :rw] = r
    return out


def layout_preprocess(img: np.ndarray, S: int, mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0)) -> np.ndarray:
    """PPPreProcess.__call__ (pre_process.py:22-42): cubic resize, x / 255, (x - mean) / std in float64, CHW, float32."""
    r = resize_cubic_u8(img, (S, S))
    x = (r.astype("float32") * (1 / 255.0) - np.array(mean)) / np.array(std)
    return x.transpose(2, 0, 1)[None].astype(np.float32)
