#!/usr/bin/env python3
"""Golden vectors of the formula pre-process, minted by the reference's own PPPreProcess (build container only).

    python tests/golden/make_golden_formula_pre.py        # writes tests/golden/formula_pre.json

What runs: rapid_doc/model/formula/rapid_formula_self/model_handler/pp_formulanet_plus/pre_process.py, loaded by file path,
unmodified - UniMERNetImgDecode (crop_margin, the short-side resize arithmetic, thumbnail, centre padding: PIL is installed, so the
resampling is the reference's own), UniMERNetTestTransform, LatexImageFormat.
Its four cv2 calls are stood in for by their definitions: findNonZero / boundingRect (the bounding box of the non-zero pixels),
merge (channel stack), cvtColor(COLOR_BGR2GRAY) on float32 = 0.114 c0 + 0.587 c1 + 0.299 c2 in float32, summed left to right - the
order of that sum inside OpenCV is the one thing these vectors do not pin.
The JSON holds, per seeded input: shapes and crc32 of the uint8 384 x 384 image and of the float32 network input."""
import importlib.util
import json
import sys
import types
import zlib
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference")

CASES = [(40, 300), (200, 90), (384, 384), (50, 50), (700, 1200), (13, 9), (384, 200), (120, 385), (64, 64), (30, 900)]


def make_image(seed: int, h: int, w: int) -> np.ndarray:
    """A light page patch with dark strokes inside a sub-rectangle (so that the margin crop has something to do).  seed % 5 == 4:
    a uniform patch (the reference returns it uncropped); seed % 5 == 3: grey-level input (H x W)."""
    rng = np.random.default_rng(seed)
    img = rng.integers(225, 256, (h, w, 3)).astype(np.uint8)
    if seed % 5 == 4:
        return np.full((h, w, 3), 180, np.uint8)
    y0, y1 = sorted(rng.integers(0, h, 2).tolist())
    x0, x1 = sorted(rng.integers(0, w, 2).tolist())
    y1, x1 = max(y1, y0 + 1), max(x1, x0 + 1)
    for _ in range(12):
        yy, xx = int(rng.integers(y0, y1)), int(rng.integers(x0, x1))
        hh, ww = int(rng.integers(1, max(2, (y1 - y0) // 3 + 1))), int(rng.integers(1, max(2, (x1 - x0) // 3 + 1)))
        img[yy:min(yy + hh, y1), xx:min(xx + ww, x1)] = rng.integers(0, 90, 3).astype(np.uint8)
    return img[..., 1].copy() if seed % 5 == 3 else img


def fake_cv2():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2GRAY = 6

    def find_non_zero(gray):
        ys, xs = np.nonzero(gray)
        return None if len(xs) == 0 else np.stack([xs, ys], 1).reshape(-1, 1, 2).astype(np.int32)

    def bounding_rect(coords):
        if coords is None:
            return 0, 0, 0, 0
        c = coords.reshape(-1, 2)
        return int(c[:, 0].min()), int(c[:, 1].min()), int(c[:, 0].max() - c[:, 0].min() + 1), int(c[:, 1].max() - c[:, 1].min() + 1)

    def cvt_color(img, code):
        assert code == cv2.COLOR_BGR2GRAY and img.dtype == np.float32
        return (np.float32(0.114) * img[..., 0] + np.float32(0.587) * img[..., 1] + np.float32(0.299) * img[..., 2]).astype(np.float32)
    cv2.findNonZero, cv2.boundingRect, cv2.cvtColor = find_non_zero, bounding_rect, cvt_color
    cv2.merge = lambda chans: np.stack(chans, axis=-1)
    return cv2


def crc(a) -> int:
    return zlib.crc32(np.ascontiguousarray(a).tobytes())


def main():
    sys.modules["cv2"] = fake_cv2()
    spec = importlib.util.spec_from_file_location(
        "ref_formula_pre", REF / "rapid_doc/model/formula/rapid_formula_self/model_handler/pp_formulanet_plus/pre_process.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    pre = ref.PPPreProcess((384, 384))
    out = []
    for seed, (h, w) in enumerate(CASES * 2):
        img = make_image(seed, h, w)
        decoded = pre.uni_mer_net_img_decode([img])[0]
        x = pre([img])[0]
        out.append({"seed": seed, "hw": [h, w], "decoded_shape": list(decoded.shape), "decoded_crc32": crc(decoded),
                    "input_shape": list(x.shape), "input_dtype": str(x.dtype), "input_crc32": crc(x),
                    "input_sum": float(np.asarray(x, np.float64).sum())})
        print(seed, (h, w), img.shape, "->", decoded.shape, x.shape, x.dtype)
    (HERE / "formula_pre.json").write_text(json.dumps({"cases": out}))


if __name__ == "__main__":
    main()
