"""Host-side glue of the OCR path: the arithmetic the reference leaves to the third-party `rapidocr`
package (pinned >=3.4.0,<=3.9.0 in the reference's pyproject.toml:38, not vendored) restated from its call
sites in the reference and from the public PaddleOCR definitions of the same steps.

 * det pre-processing geometry ........ rapid_doc/model/ocr/rapid_ocr.py:517-518 (DetPreProcess, limit 960 'max')
 * rec batching / resize geometry ..... rapid_doc/model/ocr/rapid_ocr.py:404-472 (text_recognizer_call)
 * CTC greedy decode .................. rapid_doc/model/ocr/rapid_ocr.py:444-449 (CTCLabelDecode)
 * score formatting ................... rapid_doc/backend/pipeline/analyze_utils.py:278-292
"""
from __future__ import annotations

import math
from typing import List, Sequence, Tuple

import numpy as np

REC_IMG_H = 48
REC_IMG_W = 320


def rec_seq_len(W: int) -> int:
    """Time steps of PP-OCRv6 rec for input width W (== rd_rec_seq_len)."""
    if W < 16:
        return 0
    w1 = (W - 1) // 2 + 1
    w2 = (w1 - 1) // 2 + 1
    return (w2 - 2) // 2 + 1


def det_resize_shape(h: int, w: int, limit_side_len: int = 960, limit_type: str = "max") -> Tuple[int, int]:
    """PaddleOCR DetResizeForTest: scale so the max (or min) side meets the limit, then round to x32."""
    if limit_type == "max":
        ratio = float(limit_side_len) / max(h, w) if max(h, w) > limit_side_len else 1.0
    else:
        ratio = float(limit_side_len) / min(h, w) if min(h, w) < limit_side_len else 1.0
    rh, rw = int(h * ratio), int(w * ratio)
    rh = max(int(round(rh / 32) * 32), 32)
    rw = max(int(round(rw / 32) * 32), 32)
    return rh, rw


def build_characters(dict_lines: Sequence[str], use_space_char: bool = True) -> List[str]:
    """['blank'] + dictionary + [' ']  (class 0 is the CTC blank; reference SURVEY appendix A.2)."""
    chars = [ln.rstrip("\r\n") for ln in dict_lines]
    if use_space_char:
        chars.append(" ")
    return ["blank"] + chars


def load_characters(path: str) -> List[str]:
    with open(path, "r", encoding="utf-8") as f:
        return build_characters(f.readlines())


def ctc_decode(idx: np.ndarray, prob: np.ndarray, characters: Sequence[str]) -> List[Tuple[str, float]]:
    """Greedy CTC decode of per-step (argmax, max-prob): collapse repeats, drop blank (0), confidence = mean
    of the kept max-probabilities (0 when nothing is kept)."""
    idx = np.asarray(idx)
    prob = np.asarray(prob)
    out = []
    for b in range(idx.shape[0]):
        row = idx[b]
        sel = np.ones(len(row), dtype=bool)
        sel[1:] = row[1:] != row[:-1]
        sel &= row != 0
        conf = prob[b][sel]
        text = "".join(characters[int(i)] for i in row[sel])
        out.append((text, float(np.mean(conf)) if conf.size else 0.0))
    return out


def format_score(score: float) -> float:
    """analyze_utils.py:280: float(f'{score:.3f}')"""
    return float(f"{score:.3f}")


def rec_batches(wh_ratios: Sequence[float], rec_batch_num: int = 6, img_h: int = REC_IMG_H, img_w: int = REC_IMG_W,
                width_multiple: int = 1) -> List[Tuple[np.ndarray, int]]:
    """Reference batching (rapid_ocr.py:411-440): argsort by w/h, chunks of rec_batch_num, every chunk padded to
    imgW = int(img_h * max(img_w/img_h, chunk max ratio)).  Returns [(indices into the input, padded width)].
    `width_multiple` > 1 rounds the padded width up (bounds the number of distinct shapes on the GPU)."""
    order = np.argsort(np.asarray(wh_ratios, dtype=np.float64), kind="stable")
    out = []
    for beg in range(0, len(order), rec_batch_num):
        chunk = order[beg: beg + rec_batch_num]
        max_ratio = max(img_w / img_h, max(wh_ratios[i] for i in chunk))
        wpad = int(img_h * max_ratio)
        if width_multiple > 1:
            wpad = (wpad + width_multiple - 1) // width_multiple * width_multiple
        out.append((chunk, wpad))
    return out


def rec_resized_width(w: float, h: float, wpad: int, img_h: int = REC_IMG_H) -> int:
    """resize_norm_img: resized_w = min(imgW, ceil(imgH * w/h))."""
    return int(min(wpad, math.ceil(img_h * (w / h))))
