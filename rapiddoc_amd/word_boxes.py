"""Word / character boxes of a recognised text line - the `return_word_box=True` branch of the reference's table OCR
(`use_word_box` defaults to True: rapid_doc/backend/pipeline/analyze_utils.py:308, `_run_table_ocr` :478-540 ->
`RapidOcrModel.ocr(det=False, return_word_box=True)` rapid_doc/model/ocr/rapid_ocr.py:282-299 -> `calc_word_boxes` :301-331).

Provenance, function by function:
  * `get_word_info`, `cal_ocr_word_box`: RapidDoc's OWN replacements of rapidocr's methods (rapid_doc/model/ocr/ocr_patch.py:333-389 and
    :264-329) - restated here and PINNED to that code by tests/golden/word_box.json (tests/golden/make_golden_word_box.py runs the
    reference's functions).
  * `calc_word_boxes`, `map_boxes_to_original`: rapid_ocr.py:301-345, pinned by the same fixture.
  * everything else (`decode_word_info`'s line_txt_len scaling, `calc_box`, `calc_en_num_box`, `calc_avg_char_width`,
    `calc_all_char_avg_width`, `adjust_box_overlap`, `get_box_direction`, `reverse_rotate_crop_image`, `order_points`, `quads_to_rect_bbox`,
    `has_chinese_char`) lives in the third-party package rapidocr (pinned `>=3.4.0,<=3.9.0`, pyproject.toml:38), which is ABSENT from
    /root/reference: restated from the public rapidocr source (rapidocr/cal_rec_boxes/main.py, ch_ppocr_rec/utils.py, utils/utils.py) -
    **parity unpinned**.
Host code: a table holds tens of lines; the device side delivers, per line, the kept characters' time steps and probabilities
(`rd_ctc_collapse_lines`, kept_cols / kept_conf)."""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass, field
from typing import List, Optional, Sequence, Tuple

import numpy as np

CN, EN_NUM = "cn", "en&num"          # rapidocr WordType.CN / WordType.EN_NUM


@dataclass
class WordInfo:                      # rapidocr/ch_ppocr_rec/typings.py
    words: List[List[str]] = field(default_factory=list)
    word_cols: List[List[int]] = field(default_factory=list)
    word_types: List[str] = field(default_factory=list)
    line_txt_len: float = 0.0
    confs: List[float] = field(default_factory=list)


def has_chinese_char(text: str) -> bool:
    return bool(_cjk_mask(text).any())


def quads_to_rect_bbox(bbox: np.ndarray) -> Tuple[float, float, float, float]:
    """[N, 4, 2] quads -> (x_min, y_min, x_max, y_max) over all of them."""
    if bbox.ndim != 3:
        raise ValueError("bbox shape must be 3")
    if bbox.shape[1] != 4 or bbox.shape[2] != 2:
        raise ValueError("bbox shape must be (N, 4, 2)")
    all_x, all_y = bbox[:, :, 0].flatten(), bbox[:, :, 1].flatten()
    return float(np.min(all_x)), float(np.min(all_y)), float(np.max(all_x)), float(np.max(all_y))


# ---------------------------------------------------------------------------------------------------------------------
# RapidDoc's patched methods (pinned)
# ---------------------------------------------------------------------------------------------------------------------
def _cjk_mask(text: str) -> np.ndarray:
    return np.fromiter(("\u4e00" <= ch <= "\u9fff" for ch in text), dtype=bool, count=len(text))


def get_word_info(text: str, kept_cols: Sequence[int]) -> WordInfo:
    """What RapidDoc's replacement of `CTCLabelDecode.get_word_info` computes (ocr_patch.py:333-389; its `selection` mask is True at
    `kept_cols`): the line's characters cut into words.  Character i sits at time step kept_cols[i].  A word boundary falls in front of
    character i when it or its predecessor is white space (a space is a word of its own, typed "en&num"), when the script changes
    between two neighbours (CJK / everything else), or when their time steps are more than 5 apart.  Stated here as ONE boundary mask
    over the line and a split at its set positions."""
    steps = np.asarray(kept_cols, dtype=np.int64).reshape(-1)
    n = min(len(text), steps.size)           # (character i is paired with kept step i; a dictionary entry that decodes to nothing
    if n == 0:                               #  leaves more steps than characters - the surplus steps at the end stay unused)
        return WordInfo()
    chars, steps = text[:n], steps[:n]
    blank = np.fromiter((ch.isspace() for ch in chars), dtype=bool, count=n)
    cjk = _cjk_mask(chars)
    far = np.zeros(n, dtype=bool)
    far[1:] = np.diff(steps) > 5             # (the first character's nominal width, min(3 | 2, step 0), never exceeds 5)
    cut = np.ones(n, dtype=bool)
    cut[1:] = blank[1:] | blank[:-1] | (cjk[1:] != cjk[:-1]) | far[1:]
    starts = np.flatnonzero(cut)
    stops = np.append(starts[1:], n)
    words = [list(chars[a:e]) for a, e in zip(starts, stops)]
    cols = [steps[a:e].tolist() for a, e in zip(starts, stops)]
    kinds = [EN_NUM if blank[a] or not cjk[a] else CN for a in starts]
    return WordInfo(words=words, word_cols=cols, word_types=kinds)


def cal_ocr_word_box(rec_txt: str, bbox: np.ndarray, word_info: WordInfo, return_single_char_box: bool = False, helpers=None):
    """What RapidDoc's replacement of `CalRecBoxes.cal_ocr_word_box` returns (ocr_patch.py:264-329): (contents, boxes, confidences) - one
    entry per WORD when the whole line is latin / digits (and single-character boxes were not asked for), else one per CHARACTER.  A
    cell of the line is (box width) / line_txt_len wide; every word of two or more characters contributes one estimate of the
    character width, their pooled average places the boxes; an entry's confidence is the mean of its time steps' probabilities, rounded
    to five places (steps without a probability do not count, none at all gives 0).  `helpers`: an object with rapidocr's calc_*
    methods (the tests pin the call flow with recording stand-ins); default = this module's restatements."""
    geo = helpers if helpers is not None else _Helpers
    if not rec_txt or word_info.line_txt_len == 0:
        return [], [], []
    rect = quads_to_rect_bbox(bbox[None, ...])
    cell = (rect[2] - rect[0]) / word_info.line_txt_len
    by_word = not return_single_char_box and all(k == EN_NUM for k in word_info.word_types)
    prob_at = dict(zip((c for wc in word_info.word_cols for c in wc), word_info.confs))

    def mean_prob(steps) -> float:
        got = [prob_at[c] for c in steps if c in prob_at]
        return round(float(np.mean(got)), 5) if got else 0.0
    widths = [geo.calc_avg_char_width(wc, cell) for wc in word_info.word_cols if len(wc) != 1]
    if by_word:
        contents = ["".join(w) for w in word_info.words]
        groups = [wc for wc in word_info.word_cols]
        confs = [mean_prob(wc) for wc in word_info.word_cols]
    else:
        contents = [ch for w in word_info.words for ch in w]
        groups = [c for wc in word_info.word_cols for c in wc]
        confs = [mean_prob((c,)) for c in groups]
    char_w = geo.calc_all_char_avg_width(widths, rect[0], rect[2], len(rec_txt))
    place = geo.calc_en_num_box if by_word else geo.calc_box
    return contents, place(groups, char_w, cell, rect), confs


def map_boxes_to_original(dt_boxes: np.ndarray, op_record: dict, ori_h: int, ori_w: int) -> np.ndarray:
    """rapid_ocr.py:333-352 (the reference's copy of rapidocr's method): undo the recorded padding / resize, clip to the image."""
    for op in reversed(list(op_record.keys())):
        v = op_record[op]
        if "padding" in op:
            dt_boxes[:, :, 0] -= v.get("left")
            dt_boxes[:, :, 1] -= v.get("top")
        elif "preprocess" in op:
            dt_boxes[:, :, 0] *= v.get("ratio_w")
            dt_boxes[:, :, 1] *= v.get("ratio_h")
    dt_boxes[:, :, 0] = np.where(dt_boxes[:, :, 0] < 0, 0, dt_boxes[:, :, 0])
    dt_boxes[:, :, 0] = np.where(dt_boxes[:, :, 0] > ori_w, ori_w, dt_boxes[:, :, 0])
    dt_boxes[:, :, 1] = np.where(dt_boxes[:, :, 1] < 0, 0, dt_boxes[:, :, 1])
    dt_boxes[:, :, 1] = np.where(dt_boxes[:, :, 1] > ori_h, ori_h, dt_boxes[:, :, 1])
    return dt_boxes


def calc_word_boxes(word_lines: Sequence[Sequence[tuple]], raw_h: int, raw_w: int) -> tuple:
    """rapid_ocr.py:301-331 after `cal_rec_boxes`: per line the (text, score, box) words mapped to the original image (identity record:
    the table crop IS the original) as int32 points, words without a box dropped - and LINES WITHOUT WORDS DROPPED, so the result can be
    shorter than the line list (the caller zips it with the texts as it is: rapid_ocr.py:295)."""
    op_record = {"padding_1": {"left": 0, "top": 0}, "preprocess": {"ratio_h": 1.0, "ratio_w": 1.0}}
    origin_words = []
    for word_line in word_lines:
        item = []
        for txt, score, bbox in word_line:
            if bbox is None:
                continue
            pts = map_boxes_to_original(np.array([bbox]).astype(np.float64), op_record, raw_h, raw_w)
            item.append((txt, score, pts.astype(np.int32).tolist()[0]))
        if item:
            origin_words.append(tuple(item))
    return tuple(origin_words)


# ---------------------------------------------------------------------------------------------------------------------
# rapidocr's CalRecBoxes / CTCLabelDecode pieces (restated from the public package: parity unpinned)
# ---------------------------------------------------------------------------------------------------------------------
class _Helpers:
    @staticmethod
    def calc_avg_char_width(word_col: Sequence[int], each_col_width: float) -> float:
        return (word_col[-1] - word_col[0]) * each_col_width / (len(word_col) - 1)

    @staticmethod
    def calc_all_char_avg_width(width_list: Sequence[float], bbox_x0: float, bbox_x1: float, txt_len: int) -> float:
        if txt_len == 0:
            return 0.0
        if len(width_list) > 0:
            return sum(width_list) / len(width_list)
        return (bbox_x1 - bbox_x0) / txt_len

    @staticmethod
    def calc_box(line_cols: Sequence[int], avg_char_width: float, avg_col_width: float, bbox_points) -> list:
        x0, y0, x1, y1 = bbox_points
        results = []
        for col_idx in line_cols:
            center_x = (col_idx + 0.5) * avg_col_width                      # the middle of the time step's column
            char_x0 = max(int(center_x - avg_char_width / 2), 0) + x0
            char_x1 = min(int(center_x + avg_char_width / 2), x1 - x0) + x0
            results.append([[char_x0, y0], [char_x1, y0], [char_x1, y1], [char_x0, y1]])
        return sorted(results, key=lambda b: b[0][0])

    @staticmethod
    def calc_en_num_box(line_cols: Sequence[Sequence[int]], avg_char_width: float, avg_col_width: float, bbox_points) -> list:
        results = []
        for one_col in line_cols:
            cells = _Helpers.calc_box(one_col, avg_char_width, avg_col_width, bbox_points)
            x0, y0, x1, y1 = quads_to_rect_bbox(np.array(cells, dtype=np.float64))
            results.append([[x0, y0], [x1, y0], [x1, y1], [x0, y1]])
        return results


def decode_word_info(text: str, kept_cols: Sequence[int], kept_conf: Sequence[float], n_steps: int, wh_ratio: float, max_wh_ratio: float) -> WordInfo:
    """What rapidocr's CTCLabelDecode returns per line with return_word_box: RapidDoc's get_word_info on the kept time steps, the kept
    probabilities as `confs`, and line_txt_len = the line's time steps scaled by wh_ratio / max_wh_ratio, i.e. the steps the UNPADDED part
    of the recogniser input covers (the box's width is divided by it to get the width of one time step)."""
    info = get_word_info(text, kept_cols)
    info.line_txt_len = n_steps * (wh_ratio / max_wh_ratio)
    info.confs = [float(c) for c in kept_conf]
    return info


def adjust_box_overlap(word_box_list: list) -> list:
    for i in range(len(word_box_list) - 1):
        cur, nxt = word_box_list[i], word_box_list[i + 1]
        if cur[1][0] > nxt[0][0]:                      # neighbours overlap: meet in the middle
            distance = abs(cur[1][0] - nxt[0][0])
            cur[1][0] -= distance / 2
            cur[2][0] -= distance / 2
            nxt[0][0] += distance - distance / 2
            nxt[3][0] += distance - distance / 2
    return word_box_list


def get_box_direction(box: np.ndarray) -> str:
    w = int(max(np.linalg.norm(box[0] - box[1]), np.linalg.norm(box[2] - box[3])))
    h = int(max(np.linalg.norm(box[0] - box[3]), np.linalg.norm(box[1] - box[2])))
    return "h" if h * 1.0 / w >= 1.5 else "w"


def _perspective_transform(src: np.ndarray, dst: np.ndarray) -> np.ndarray:
    """cv2.getPerspectiveTransform: the 3x3 homography taking the four `src` points to `dst` (float64 solve)."""
    a, b = [], []
    for (x, y), (u, v) in zip(src.astype(np.float64), dst.astype(np.float64)):
        a.append([x, y, 1, 0, 0, 0, -u * x, -u * y])
        a.append([0, 0, 0, x, y, 1, -v * x, -v * y])
        b += [u, v]
    hvec = np.linalg.solve(np.asarray(a), np.asarray(b))
    return np.append(hvec, 1.0).reshape(3, 3)


def order_points(ori_box: Sequence[Sequence[float]]) -> List[List[int]]:
    """Four points -> top-left, top-right, bottom-right, bottom-left."""
    pts = sorted(([int(p[0]), int(p[1])] for p in ori_box), key=lambda p: p[0])
    left, right = sorted(pts[:2], key=lambda p: p[1]), sorted(pts[2:], key=lambda p: p[1])
    return [left[0], right[0], right[1], left[1]]


def reverse_rotate_crop_image(bbox_points: np.ndarray, word_points_list: list, direction: str = "w") -> list:
    """The inverse of get_rotate_crop_image: word boxes in crop coordinates -> points of the image the line was cropped from."""
    bbox_points = np.float32(bbox_points)
    left, top = int(np.min(bbox_points[:, 0])), int(np.min(bbox_points[:, 1]))
    bbox_points[:, 0] -= left
    bbox_points[:, 1] -= top
    crop_w = int(np.linalg.norm(bbox_points[0] - bbox_points[1]))
    crop_h = int(np.linalg.norm(bbox_points[0] - bbox_points[3]))
    pts_std = np.float32([[0, 0], [crop_w, 0], [crop_w, crop_h], [0, crop_h]])
    inv = np.linalg.inv(_perspective_transform(bbox_points, pts_std))
    out = []
    for word_points in word_points_list:
        new_points = []
        for point in word_points:
            px, py = float(point[0]), float(point[1])
            if direction == "h":                        # the crop was rotated by 90 degrees: turn the point back first
                ang = math.radians(-90)
                px, py = px * math.cos(ang) + py * math.sin(ang), py * math.cos(ang) - px * math.sin(ang)
                px += crop_w
            x, y, z = np.dot(inv, np.float32([px, py, 1]))
            new_points.append([int(x / z + left), int(y / z + top)])
        out.append(order_points(new_points))
    return out


def cal_rec_boxes(crop_hw: Sequence[Tuple[int, int]], dt_boxes: Sequence[np.ndarray], texts: Sequence[str], infos: Sequence[WordInfo],
                  return_single_char_box: bool = False) -> List[list]:
    """rapidocr CalRecBoxes.__call__: per line [(word, conf, box in the coordinates of the image the line was cropped from)]."""
    out = []
    for (h, w), box, txt, info in zip(crop_hw, dt_boxes, texts, infos):
        img_box = np.array([[0.0, 0.0], [w, 0.0], [w, h], [0.0, h]])
        contents, boxes, confs = cal_ocr_word_box(txt, img_box, info, return_single_char_box)
        boxes = adjust_box_overlap(copy.deepcopy(boxes))
        boxes = reverse_rotate_crop_image(copy.deepcopy(np.asarray(box, dtype=np.float32)), boxes, get_box_direction(np.asarray(box, dtype=np.float32)))
        out.append(list(zip(contents, confs, boxes)))
    return out
