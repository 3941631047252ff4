"""Host side of the table-structure path: what turns SLANet_plus outputs into HTML structure
tokens and cell boxes.  The SLANet_plus / UNet / classifier networks themselves exist only as ONNX files the reference
downloads (absent here), so this is the part of row a17/a18 that can be built and pinned today; it is written against
arrays so that a future GPU argmax (like the fused CTC head) can feed it (idx, prob) instead of the [B, L, V] tensor.

Restates `TableLabelDecode` (rapid_doc/model/table/rapid_table_self/table_structure/pp_structure/post_process.py:12-131)
and `wrap_with_html_struct` (.../table_structure/utils.py:7-13); pinned by tests/golden/table_decode_seed*.json, minted by
calling the reference class itself (tests/golden/make_golden.py)."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np

SLANET_PLUS_INPUT = 488          # post_process.py:119 (`resized = 488`)
_HTML_HEAD = ["<html>", "<body>", "<table>"]
_HTML_TAIL = ["</table>", "</body>", "</html>"]


class TableStructureDecoder:
    def __init__(self, dict_character: Sequence[str], slanet_plus: bool = True, merge_no_span_structure: bool = True):
        chars = list(dict_character)
        if merge_no_span_structure:          # post_process.py:14-18
            if "<td></td>" not in chars:
                chars.append("<td></td>")
            if "<td>" in chars:
                chars.remove("<td>")
        self.character = ["sos"] + chars + ["eos"]          # post_process.py:102-106
        self.index = {c: i for i, c in enumerate(self.character)}   # later duplicates win, like the reference's loop
        self.sos, self.eos = self.index["sos"], self.index["eos"]
        self.is_cell = np.array([c in ("<td>", "<td", "<td></td>") for c in self.character])
        self.slanet_plus = slanet_plus

    def decode_indices(self, idx: np.ndarray, prob: np.ndarray, bbox_preds: np.ndarray, shape_list: np.ndarray,
                       ori_shapes: Sequence[Tuple[int, int]]):
        """idx/prob [B, L] (argmax and max over the vocabulary), bbox_preds [B, L, 8] normalised, shape_list[b][:2] =
        (h, w) of the network input, ori_shapes[b] = (H, W) of the source image."""
        structs: List[Tuple[List[str], float]] = []
        boxes_out: List[np.ndarray] = []
        for b in range(idx.shape[0]):
            row = idx[b].astype(np.int64)
            stop = np.nonzero(row[1:] == self.eos)[0]          # an eos at position 0 does not stop (post_process.py:60)
            end = int(stop[0]) + 1 if len(stop) else len(row)
            pos = np.arange(end)
            keep = pos[(row[:end] != self.sos) & (row[:end] != self.eos)]
            tokens = [self.character[int(row[i])] for i in keep]
            with np.errstate(invalid="ignore"):
                score = float(np.mean(prob[b][keep])) if len(keep) else float("nan")     # np.mean([]) is nan there too
            cells = keep[self.is_cell[row[keep]]]
            h, w = float(shape_list[b][0]), float(shape_list[b][1])
            bb = np.array(bbox_preds[b][cells], dtype=bbox_preds.dtype, copy=True).reshape(-1, bbox_preds.shape[-1])
            bb[:, 0::2] *= w
            bb[:, 1::2] *= h
            if self.slanet_plus and len(bb):                   # rescale_cell_bboxes, post_process.py:114-125
                H, W = ori_shapes[b][:2]
                ratio = min(SLANET_PLUS_INPUT / H, SLANET_PLUS_INPUT / W)
                bb[:, 0::2] *= SLANET_PLUS_INPUT / (W * ratio)
                bb[:, 1::2] *= SLANET_PLUS_INPUT / (H * ratio)
            if len(bb):
                bb = bb[~np.all(bb == 0, axis=1)]              # placeholder boxes, post_process.py:127-131
            structs.append((_HTML_HEAD + tokens + _HTML_TAIL, score))
            boxes_out.append(bb)
        return structs, boxes_out

    def decode(self, bbox_preds: np.ndarray, structure_probs: np.ndarray, shape_list: np.ndarray, ori_imgs: Sequence[np.ndarray]):
        """Reference signature (post_process.py:39-80): structure_probs [B, L, V]."""
        return self.decode_indices(structure_probs.argmax(axis=2), structure_probs.max(axis=2), bbox_preds, shape_list,
                                   [im.shape[:2] for im in ori_imgs])


# ----------------------------------------------------------------------------------------------------------------------
# OCR text of a table line before it is matched into cells: rapid_doc/model/table/utils.py:7-36 `normalize_table_ocr_text`
# (two known mis-readings of single cells, "<digit>號" -> the digit, then HTML escaping).  The tables are data of the reference.
# ----------------------------------------------------------------------------------------------------------------------
import html as _html
import re as _re

_SINGLE_CELL_FIXES = {"香": "否", "哦樂": "哦"}
_FULLMATCH_FIXES = ((_re.compile(r"^([0-9])號$"), r"\1"),)


def normalize_table_ocr_text(text) -> str:
    if text is None:
        return ""
    text = str(text).strip()
    text = _SINGLE_CELL_FIXES.get(text, text)
    for pattern, replacement in _FULLMATCH_FIXES:
        m = pattern.fullmatch(text)
        if m:
            text = m.expand(replacement)
            break
    return _html.escape(text)
