"""CPU: the word-box branch of the table OCR (rapiddoc_amd/word_boxes.py) against vectors minted by the REFERENCE's own functions
(tests/golden/make_golden_word_box.py: RapidDoc's patched get_word_info / cal_ocr_word_box, RapidOcrModel.calc_word_boxes).  rapidocr's own
helpers (calc_box ...) are third-party and absent: their restatements are exercised by known-answer tests only - parity unpinned."""
import json

import numpy as np
import pytest

from rapiddoc_amd import word_boxes as WB


@pytest.fixture(scope="module")
def fx(golden_dir):
    return json.loads((golden_dir / "word_box.json").read_text())


def test_get_word_info_equals_the_reference(fx):
    assert len(fx["get_word_info"]) > 50
    for c in fx["get_word_info"]:
        info = WB.get_word_info(c["text"], c["cols"])
        assert (info.words, info.word_cols, info.word_types) == (c["words"], c["word_cols"], c["word_types"]), c["text"]


class _Recording:
    """The generator's closed-form helper stand-ins (same formulas), recording their calls."""
    def __init__(self):
        self.calls = []

    def calc_avg_char_width(self, word_col, each_col_width):
        self.calls.append(["calc_avg_char_width", list(word_col), each_col_width])
        return 1.5 * each_col_width + len(word_col)

    def calc_all_char_avg_width(self, width_list, x0, x1, txt_len):
        self.calls.append(["calc_all_char_avg_width", list(width_list), x0, x1, txt_len])
        return (sum(width_list) + 1.0) / (len(width_list) + 1)

    def calc_en_num_box(self, line_cols, avg_char_width, avg_col_width, bbox_points):
        self.calls.append(["calc_en_num_box", [list(c) for c in line_cols], avg_char_width, avg_col_width, list(bbox_points)])
        return [[[c[0] * avg_col_width, 0.0], [c[-1] * avg_col_width + avg_char_width, 0.0], [c[-1] * avg_col_width + avg_char_width, 9.0],
                 [c[0] * avg_col_width, 9.0]] for c in line_cols]

    def calc_box(self, line_cols, avg_char_width, avg_col_width, bbox_points):
        self.calls.append(["calc_box", list(line_cols), avg_char_width, avg_col_width, list(bbox_points)])
        return [[[c * avg_col_width, 1.0], [c * avg_col_width + avg_char_width, 1.0], [c * avg_col_width + avg_char_width, 8.0],
                 [c * avg_col_width, 8.0]] for c in line_cols]


def test_cal_ocr_word_box_flow_equals_the_reference(fx):
    """Same helper calls with the same arguments in the same order, same contents / boxes / confidences (bit for bit: the arithmetic
    around the helpers is float64 on both sides)."""
    assert len(fx["cal_ocr_word_box"]) > 100
    for c in fx["cal_ocr_word_box"]:
        info = WB.WordInfo(words=c["words"], word_cols=c["word_cols"], word_types=c["word_types"], line_txt_len=c["line_txt_len"], confs=c["confs"])
        rec = _Recording()
        contents, boxes, confs = WB.cal_ocr_word_box(c["text"], np.asarray(c["bbox"], dtype=np.float64), info, c["single"], helpers=rec)
        assert rec.calls == c["calls"], c["text"]
        assert contents == c["contents"] and boxes == c["boxes"] and confs == c["out_confs"], c["text"]


def test_calc_word_boxes_equals_the_reference(fx):
    """Boxes clipped to the image as int32 points, words without a box dropped, LINES WITHOUT WORDS DROPPED (the result is shorter than the
    line list then)."""
    assert any(len(c["out"]) < len(c["lines"]) for c in fx["calc_word_boxes"])
    for c in fx["calc_word_boxes"]:
        lines = [[(w[0], w[1], w[2]) for w in line] for line in c["lines"]]
        out = WB.calc_word_boxes(lines, c["raw_hw"][0], c["raw_hw"][1])
        assert json.loads(json.dumps(out)) == c["out"]


def test_rapidocr_helper_restatements_known_answers():
    """(unpinned) a 100 x 20 crop whose 10 time steps cover it: step k sits at x = 10 k .. 10 k + 10."""
    bbox = (0.0, 0.0, 100.0, 20.0)
    cells = WB._Helpers.calc_box([1, 4], 10.0, 10.0, bbox)
    assert cells == [[[10.0, 0.0], [20.0, 0.0], [20.0, 20.0], [10.0, 20.0]], [[40.0, 0.0], [50.0, 0.0], [50.0, 20.0], [40.0, 20.0]]]
    assert WB._Helpers.calc_en_num_box([[1, 2], [7]], 10.0, 10.0, bbox) == [[[10.0, 0.0], [30.0, 0.0], [30.0, 20.0], [10.0, 20.0]],
                                                                            [[70.0, 0.0], [80.0, 0.0], [80.0, 20.0], [70.0, 20.0]]]
    assert WB._Helpers.calc_avg_char_width([2, 4, 8], 10.0) == 30.0
    assert WB._Helpers.calc_all_char_avg_width([], 0.0, 100.0, 4) == 25.0 and WB._Helpers.calc_all_char_avg_width([10.0, 20.0], 0, 1, 4) == 15.0
    boxes = WB.adjust_box_overlap([[[0, 0], [12, 0], [12, 5], [0, 5]], [[8, 0], [20, 0], [20, 5], [8, 5]]])
    assert boxes[0][1][0] == boxes[1][0][0] == 10.0
    # an axis-aligned line box: crop coordinates map back by a shift; a tall box is read as a rotated crop
    quad = np.array([[50, 30], [150, 30], [150, 50], [50, 50]], np.float32)
    assert WB.get_box_direction(quad) == "w"
    back = WB.reverse_rotate_crop_image(quad.copy(), [[[10, 0], [20, 0], [20, 20], [10, 20]]], "w")
    assert back == [[[60, 30], [70, 30], [70, 50], [60, 50]]]
    tall = np.array([[50, 30], [70, 30], [70, 130], [50, 130]], np.float32)
    assert WB.get_box_direction(tall) == "h"


def test_word_results_of_a_line_end_to_end():
    """decode_word_info + cal_rec_boxes on a latin line and a mixed line (unpinned arithmetic, pinned flow): one box per latin word, one per
    character once a CJK character is present; boxes lie inside the line's box and are ordered left to right."""
    quad = np.array([[20, 10], [220, 10], [220, 42], [20, 42]], np.float32)
    info = WB.decode_word_info("ab cd", [2, 4, 7, 9, 11], [0.9, 0.8, 0.7, 0.6, 0.5], 40, 200 / 32, 200 / 32)
    assert info.words == [["a", "b"], [" "], ["c", "d"]] and info.line_txt_len == 40
    (line,) = WB.cal_rec_boxes([(32, 200)], [quad], ["ab cd"], [info])
    assert [w for w, _c, _b in line] == ["ab", " ", "cd"] and [c for _w, c, _b in line] == [0.85, 0.7, 0.55]
    xs = [b[0][0] for _w, _c, b in line]
    assert xs == sorted(xs) and all(20 <= p[0] <= 220 and 10 <= p[1] <= 42 for _w, _c, b in line for p in b)
    info = WB.decode_word_info("a汉b", [3, 10, 20], [0.9, 0.8, 0.7], 40, 200 / 32, 200 / 32)
    (line,) = WB.cal_rec_boxes([(32, 200)], [quad], ["a汉b"], [info])
    assert [w for w, _c, _b in line] == ["a", "汉", "b"]
