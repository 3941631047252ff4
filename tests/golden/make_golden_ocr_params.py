#!/usr/bin/env python3
"""The parameters RapidDoc configures rapidocr with, captured from the REFERENCE's own constructor (build container only).

    python tests/golden/make_golden_ocr_params.py        # writes tests/golden/ocr_default_params.json

What runs: `RapidOcrModel.__init__` (rapid_doc/model/ocr/rapid_ocr.py:44-158), unmodified, for the two ways the page driver creates it
(backend/pipeline/model_init.py:14-28,45-55: page OCR with box_thresh 0.3 / unclip 1.8, table OCR with 0.5 / 1.6 and no merging) with
the torch engine selected.  `rapidocr.RapidOCR` is absent, so it is a mock - and the `params` dict the constructor hands to it is exactly
what this script records (enum members as their names, model paths as file names).  Data only."""
import json
import re
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import make_golden_recbatch as MGR  # noqa: E402  (imports rapid_ocr.py with its module-level reach into rapidocr stubbed)


def main():
    ro = MGR.import_rapid_ocr()
    ro.get_device = lambda: "cpu"
    ro.check_openvino = lambda: False
    out = {}
    for name, kw in (("page", dict(det_db_box_thresh=0.3, lang="ch", ocr_config={"engine_type": ro.EngineType.TORCH}, use_dilation=True,
                                   det_db_unclip_ratio=1.8, enable_merge_det_boxes=True, is_seal=False)),
                     ("table", dict(det_db_box_thresh=0.5, lang="ch", ocr_config={"engine_type": ro.EngineType.TORCH}, use_dilation=True,
                                    det_db_unclip_ratio=1.6, enable_merge_det_boxes=False, is_seal=False))):
        ro.RapidOCR.reset_mock()
        model = ro.RapidOcrModel(**kw)
        params = ro.RapidOCR.call_args.kwargs["params"]
        clean = {}
        for k, v in params.items():
            if k.endswith("model_path") or k.endswith("keys_path") or k.endswith("model_root_dir"):
                v = Path(str(v)).name
            elif not isinstance(v, (int, float, str, bool, list, type(None))):
                m = re.search(r"([A-Za-z0-9_]+)'?(?: id=.*)?>?$", str(v))      # enum members (mocks of the absent rapidocr enums): their name
                v = m.group(1) if m else str(v)
            clean[k] = v
        out[name] = {"params": clean, "drop_score": model.drop_score, "enable_merge_det_boxes": model.enable_merge_det_boxes}
        print(name, json.dumps(clean)[:300])
    (HERE / "ocr_default_params.json").write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
