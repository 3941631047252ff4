#!/usr/bin/env python3
"""Golden vectors for the word-box branch of the table OCR, minted by the REFERENCE's own functions (build container only).

    python tests/golden/make_golden_word_box.py        # writes tests/golden/word_box.json

What runs is the reference's code, unmodified:
    rapid_doc/model/ocr/ocr_patch.py:259-389    patch_word_box(): RapidDoc's replacements of rapidocr's CalRecBoxes.cal_ocr_word_box and
                                                CTCLabelDecode.get_word_info (the module is loaded from its file; the rapidocr names it
                                                imports at the top are stand-in classes that carry what it patches)
    rapid_doc/model/ocr/rapid_ocr.py:301-352    RapidOcrModel.calc_word_boxes / map_boxes_to_original
Stood in for (rapidocr is absent from /root/reference): WordInfo / WordType (plain containers), has_chinese_char, quads_to_rect_bbox and
the four calc_* helpers cal_ocr_word_box calls on `self` - RECORDING stand-ins with simple closed-form answers, so that the fixture pins
the reference's flow (which helper, which arguments, what is done with the answers), not rapidocr's arithmetic."""
import importlib.util
import json
import sys
import types
from dataclasses import dataclass, field
from enum import Enum
from pathlib import Path
from typing import List

import numpy as np

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/rapid_doc")


class WordType(Enum):
    CN = "cn"
    EN_NUM = "en&num"


@dataclass
class WordInfo:
    words: List[List[str]] = field(default_factory=list)
    word_cols: List[List[int]] = field(default_factory=list)
    word_types: List[WordType] = field(default_factory=list)
    line_txt_len: float = 0.0
    confs: List[float] = field(default_factory=list)


def has_chinese_char(text):
    return any("一" <= ch <= "鿿" for ch in text)


def quads_to_rect_bbox(bbox):
    return float(bbox[:, :, 0].min()), float(bbox[:, :, 1].min()), float(bbox[:, :, 0].max()), float(bbox[:, :, 1].max())


class CalRecBoxes:                       # the helpers answer in closed form and record their arguments
    calls: list = []

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


class CTCLabelDecode:
    pass


def load_patch():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m
    anything = type("Anything", (), {"__getattr__": lambda self, k: None})
    mod("cv2")
    mod("pyclipper")
    mod("rapidocr")
    mod("rapidocr.cal_rec_boxes", CalRecBoxes=CalRecBoxes)
    mod("rapidocr.ch_ppocr_det", TextDetector=type("TextDetector", (), {"sorted_boxes": None}))
    mod("rapidocr.ch_ppocr_det.utils", DetPreProcess=anything, DBPostProcess=anything)
    mod("rapidocr.ch_ppocr_rec")
    mod("rapidocr.ch_ppocr_rec.typings", WordInfo=WordInfo, WordType=WordType)
    mod("rapidocr.ch_ppocr_rec.utils", CTCLabelDecode=CTCLabelDecode)
    mod("rapidocr.inference_engine")
    mod("rapidocr.inference_engine.base", get_engine=None)
    mod("rapidocr.utils")
    mod("rapidocr.utils.utils", has_chinese_char=has_chinese_char, quads_to_rect_bbox=quads_to_rect_bbox)
    mod("rapid_doc")
    mod("rapid_doc.utils")
    mod("rapid_doc.utils.model_utils", import_package=lambda name: None)
    import importlib.metadata as md
    md.version = lambda name: "3.4.0"
    spec = importlib.util.spec_from_file_location("ref_ocr_patch", REF / "model" / "ocr" / "ocr_patch.py")
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    m.patch_word_box()
    return m


def load_calc_word_boxes():
    """RapidOcrModel.calc_word_boxes / map_boxes_to_original as plain functions: the two `def`s are cut out of rapid_ocr.py by line and
    compiled on their own (the module's imports need rapidocr)."""
    import ast
    src = (REF / "model" / "ocr" / "rapid_ocr.py").read_text()
    tree = ast.parse(src)
    cls = next(n for n in tree.body if isinstance(n, ast.ClassDef) and n.name == "RapidOcrModel")
    fns = [n for n in cls.body if isinstance(n, ast.FunctionDef) and n.name in ("calc_word_boxes", "map_boxes_to_original")]
    module = ast.Module(body=[ast.ClassDef(name="M", bases=[], keywords=[], body=fns, decorator_list=[])], type_ignores=[])
    ast.fix_missing_locations(module)
    ns = {"np": np, "List": List, "Dict": dict, "Any": object, "TextRecOutput": object}
    exec(compile(module, "rapid_ocr.py[calc_word_boxes]", "exec"), ns)
    return ns["M"]


ALPHABET = list("abcXYZ019.,-") + list("汉字表格数据") + [" ", " ", "é", "Ω"]


def main():
    ref = load_patch()
    get_word_info, cal_ocr_word_box = CTCLabelDecode.get_word_info, CalRecBoxes.cal_ocr_word_box
    rng = np.random.default_rng(20240)
    fx = {"get_word_info": [], "cal_ocr_word_box": [], "calc_word_boxes": []}
    for case in range(60):
        n = int(rng.integers(1, 18))
        kind = case % 4
        pool = ALPHABET if kind == 0 else list("abc 019.-") if kind == 1 else list("汉字表格数据") if kind == 2 else list("ab汉 字1 2")
        text = "".join(pool[int(rng.integers(0, len(pool)))] for _ in range(n))
        T = int(rng.integers(n, 4 * n + 6))
        gaps = rng.integers(1, 9, n) if case % 5 else rng.integers(1, 3, n)
        cols = np.cumsum(gaps) - 1 + int(rng.integers(0, 4))
        T = max(T, int(cols[-1]) + 1)
        sel = np.zeros(T, dtype=bool)
        sel[cols] = True
        info = get_word_info(None, text, sel)
        fx["get_word_info"].append({"text": text, "cols": cols.tolist(), "words": info.words, "word_cols": info.word_cols,
                                    "word_types": [t.value for t in info.word_types]})
        # cal_ocr_word_box on that word info, with line_txt_len / confs as rapidocr's decode would set them
        info.line_txt_len = float(T) * float(rng.uniform(0.4, 1.0))
        info.confs = [round(float(c), 6) for c in rng.uniform(0.2, 1.0, n)]
        if case % 7 == 6:
            info.confs = info.confs[:-1]                       # a short conf list: zip() drops the last column
        w, h = float(rng.integers(20, 400)), float(rng.integers(8, 48))
        bbox = np.array([[0.0, 0.0], [w, 0.0], [w, h], [0.0, h]])
        for single in (False, True):
            CalRecBoxes.calls = []
            contents, boxes, confs = cal_ocr_word_box(CalRecBoxes(), text, bbox, info, single)
            fx["cal_ocr_word_box"].append({"text": text, "bbox": bbox.tolist(), "line_txt_len": info.line_txt_len, "confs": info.confs,
                                           "words": info.words, "word_cols": info.word_cols, "word_types": [t.value for t in info.word_types],
                                           "single": single, "calls": json.loads(json.dumps(CalRecBoxes.calls)), "contents": contents,
                                           "boxes": json.loads(json.dumps(boxes)), "out_confs": confs})
    # the degenerate inputs
    e = get_word_info(None, "", np.zeros(5, dtype=bool))
    fx["get_word_info"].append({"text": "", "cols": [], "words": e.words, "word_cols": e.word_cols, "word_types": []})
    CalRecBoxes.calls = []
    fx["cal_ocr_word_box"].append({"text": "", "bbox": [[0, 0], [5, 0], [5, 5], [0, 5]], "line_txt_len": 3.0, "confs": [], "words": [], "word_cols": [],
                                   "word_types": [], "single": False, "calls": [], "contents": [], "boxes": [], "out_confs": []})
    M = load_calc_word_boxes()
    calc = M()
    for case in range(12):
        lines = []
        for _ in range(int(rng.integers(0, 5))):
            words = []
            for _w in range(int(rng.integers(0, 4))):
                x0, y0 = float(rng.uniform(-20, 300)), float(rng.uniform(-10, 120))
                box = None if rng.uniform() < 0.2 else [[x0, y0], [x0 + 30.7, y0], [x0 + 30.7, y0 + 12.2], [x0, y0 + 12.2]]
                words.append(("w%d" % _w, round(float(rng.uniform(0, 1)), 5), box))
            lines.append(words)
        raw_h, raw_w = int(rng.integers(60, 130)), int(rng.integers(150, 320))
        seen = {}

        class Engine:
            @staticmethod
            def cal_rec_boxes(img, dt_boxes, rec_res, single):
                seen["args"] = (len(img), len(dt_boxes), single)
                return types.SimpleNamespace(word_results=lines)
        calc.ocr_engine = types.SimpleNamespace(cal_rec_boxes=Engine.cal_rec_boxes, return_single_char_box=False)
        out = calc.calc_word_boxes([None] * len(lines), [None] * len(lines), types.SimpleNamespace(), {"padding_1": {"left": 0, "top": 0},
                                   "preprocess": {"ratio_h": 1.0, "ratio_w": 1.0}}, raw_h, raw_w)
        fx["calc_word_boxes"].append({"lines": json.loads(json.dumps(lines)), "raw_hw": [raw_h, raw_w], "out": json.loads(json.dumps(out))})
    (HERE / "word_box.json").write_text(json.dumps(fx, ensure_ascii=False))
    print({k: len(v) for k, v in fx.items()}, "bytes", (HERE / "word_box.json").stat().st_size)


if __name__ == "__main__":
    main()
