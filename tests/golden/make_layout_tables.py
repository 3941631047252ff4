"""Captures the per-model constant tables of the reference's layout wrapper (plain dict literals, data not code):
rapid_doc/model/layout/rapid_layout_self/utils/typings.py:14-140 (score thresholds, merge modes per class id).
Run in the build container:  python tests/golden/make_layout_tables.py   ->  rapiddoc_amd/data/layout_model_tables.json"""
import ast
import json
from pathlib import Path

REF = Path("/root/reference")
OUT = Path(__file__).resolve().parents[2] / "rapiddoc_amd" / "data" / "layout_model_tables.json"

tree = ast.parse((REF / "rapid_doc/model/layout/rapid_layout_self/utils/typings.py").read_text())
lits = {}
for node in tree.body:
    if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name) and isinstance(node.value, ast.Dict):
        try:
            lits[node.targets[0].id] = {int(k): v for k, v in ast.literal_eval(node.value).items()}
        except (ValueError, TypeError):
            pass
assert {"PP_DOCLAYOUT_PLUS_L_Threshold", "PP_DOCLAYOUT_L_Threshold", "PP_DOCLAYOUTV2_Threshold",
        "PP_DOCLAYOUT_PLUS_L_layout_merge_bboxes_mode", "PP_DOCLAYOUTV2_layout_merge_bboxes_mode"} <= set(lits), sorted(lits)
OUT.write_text(json.dumps(lits, indent=0, sort_keys=True))
print("wrote", OUT, {k: len(v) for k, v in lits.items()})
