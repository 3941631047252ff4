#!/usr/bin/env python3
"""The REFERENCE's own recogniser batching loop, run on seeded crop lists (build container only).

    python tests/golden/make_golden_recbatch.py        # writes tests/golden/rec_batching.json

What runs is `RapidOcrModel.text_recognizer_call` (rapid_doc/model/ocr/rapid_ocr.py:404-471, RapidDoc's copy of rapidocr's
TextRecognizer.__call__) from /root/reference, unmodified: the global `np.argsort(np.array(width_list))`, the chunks of
`rec_batch_num`, the `max_wh_ratio` of every chunk (starting from imgW / imgH), the order in which results are scattered back.
Stood in for: the rapidocr object it drives (`self.text_recognizer`: `resize_norm_img` returns a blank array of the width rapidocr
would produce - `int(imgH * max_wh_ratio)`, the one line of that function the fixture depends on, restated - and records its
arguments; `session` records the batch shape; `postprocess_op` names every line by its position) and the absent wheels (mocks, as in
make_golden_analyze.py).  The committed JSON holds the crop sizes and what the loop did with them: data only."""
import importlib
import json
import sys
import types
from pathlib import Path
from unittest import mock

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden_analyze as MGA  # noqa: E402  (the fake-package finder and the cv2 / wheel stand-ins)


def import_rapid_ocr():
    finder = MGA._RefFinder()
    sys.meta_path.insert(0, finder)
    sys.modules["cv2"] = MGA._fake_cv2()
    # module-level side effects of rapid_ocr.py that reach into the absent rapidocr package: the monkey patches of ocr_patch.py and
    # the seal helpers (neither is on the recogniser loop's path)
    patch = types.ModuleType("rapid_doc.model.ocr.ocr_patch")
    patch.apply_ocr_patch = lambda: None
    sys.modules["rapid_doc.model.ocr.ocr_patch"] = patch
    seal = types.ModuleType("rapid_doc.model.ocr.seal_crop")
    seal.SortPolyBoxes = seal.CropByPolys = type("Unused", (), {})
    sys.modules["rapid_doc.model.ocr.seal_crop"] = seal
    import importlib.metadata as md
    md.version = lambda name: "0.0.0"                  # the module asks for rapidocr's installed version at import time
    for _ in range(40):
        try:
            return importlib.import_module("rapid_doc.model.ocr.rapid_ocr")
        except ModuleNotFoundError as e:
            top = (e.name or "").split(".")[0]
            if not top or top == "rapid_doc" or top in finder.mocked:
                raise
            finder.mocked.add(top)
            for k in [k for k in sys.modules if k.startswith("rapid_doc.") and k not in ("rapid_doc.model.ocr.ocr_patch", "rapid_doc.model.ocr.seal_crop")]:
                del sys.modules[k]
    raise RuntimeError("could not import rapid_doc.model.ocr.rapid_ocr")


class RecordingRecognizer:
    """Stands where rapidocr's TextRecognizer stands (`self.text_recognizer`)."""

    def __init__(self, rec_batch_num, log):
        self.rec_batch_num = rec_batch_num
        self.rec_image_shape = [3, 48, 320]
        self.log = log
        self._cur = None

    def resize_norm_img(self, img, max_wh_ratio):
        imgC, imgH, imgW = self.rec_image_shape
        w = int(imgH * max_wh_ratio)                    # rapidocr: `imgW = int(imgH * max_wh_ratio)`
        if self._cur is None or self._cur["max_wh_ratio"] != float(max_wh_ratio) or self._cur["closed"]:
            self._cur = {"max_wh_ratio": float(max_wh_ratio), "imgW": w, "crops": [], "closed": False}
            self.log.append(self._cur)
        self._cur["crops"].append([int(img.shape[0]), int(img.shape[1])])
        return np.zeros((imgC, imgH, w), np.float32)

    def session(self, batch):
        self._cur["batch_shape"] = [int(v) for v in batch.shape]
        self._cur["closed"] = True
        return batch

    def postprocess_op(self, preds, return_word_box, wh_ratio_list=None, max_wh_ratio=None):
        self._cur["wh_ratio_list"] = [float(r) for r in wh_ratio_list]
        k0 = sum(len(c["crops"]) for c in self.log[:-1])
        return [(f"L{k0 + i}", 0.5) for i in range(preds.shape[0])], [None] * preds.shape[0]


def main():
    ro = import_rapid_ocr()
    fn = ro.RapidOcrModel.text_recognizer_call
    ro.TextRecOutput = lambda imgs, txts, scores, words, elapse: {"txts": list(txts), "scores": list(scores)}
    cases = []
    for seed, n, rec_batch_num in ((0, 45, 6), (1, 97, 6), (2, 13, 6), (3, 64, 16), (4, 1440, 6)):
        rng = np.random.default_rng(seed)
        hs = rng.integers(14, 48, n)
        # a list with many EQUAL ratios (np.argsort's default kind decides which of them cross a chunk border) and a wide spread
        ws = np.where(rng.random(n) < 0.3, hs * rng.integers(2, 30, n), rng.integers(20, 1200, n))
        crops = [np.zeros((int(h), int(w), 3), np.uint8) for h, w in zip(hs, ws)]
        log = []
        fake_self = types.SimpleNamespace(text_recognizer=RecordingRecognizer(rec_batch_num, log))
        out = fn(fake_self, types.SimpleNamespace(img=crops, return_word_box=False))
        cases.append({"seed": seed, "rec_batch_num": rec_batch_num, "crop_hw": [[int(h), int(w)] for h, w in zip(hs, ws)],
                      "numpy": np.__version__,
                      "chunks": [{"crops": c["crops"], "max_wh_ratio": c["max_wh_ratio"], "imgW": c["imgW"], "batch_shape": c["batch_shape"]}
                                 for c in log],
                      "txts": out["txts"]})
        print(f"seed {seed}: {n} crops -> {len(log)} chunks, widths {sorted({c['imgW'] for c in log})[:6]} ...")
    (HERE / "rec_batching.json").write_text(json.dumps({"source": "rapid_doc/model/ocr/rapid_ocr.py:404-471 (text_recognizer_call)", "cases": cases}))


if __name__ == "__main__":
    main()
