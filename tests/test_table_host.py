"""Table-structure token / cell-box decode against vectors minted by the reference's own
TableLabelDecode (tests/golden/make_golden.py: table_decode_golden)."""
import json

import numpy as np
import pytest

from rapiddoc_amd.table_host import TableStructureDecoder


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_table_decode_matches_reference(golden_dir, seed):
    g = json.loads((golden_dir / f"table_decode_seed{seed}.json").read_text())
    dec = TableStructureDecoder(g["vocab"], slanet_plus=g["slanet_plus"])
    probs = np.asarray(g["probs"], dtype=np.float32)
    bbox = np.asarray(g["bbox"], dtype=np.float32)
    shapes = np.asarray(g["shapes"], dtype=np.float32)
    oris = [np.zeros((h, w, 3), np.uint8) for h, w in g["ori_shapes"]]
    structs, boxes = dec.decode(bbox, probs, shapes, oris)
    assert len(structs) == len(g["structs"])
    for (tok, score), (rtok, rscore), bb, rbb in zip(structs, g["structs"], boxes, g["boxes"]):
        assert tok == rtok                                   # token strings identical
        assert score == pytest.approx(rscore, rel=1e-6)
        rbb = np.asarray(rbb, dtype=np.float64).reshape(-1, 8)
        assert bb.shape == rbb.shape
        assert np.allclose(bb, rbb, rtol=1e-6, atol=1e-4)
    # the (idx, prob) entry point a fused GPU argmax would feed gives the same answer
    s2, b2 = dec.decode_indices(probs.argmax(2), probs.max(2), bbox, shapes, g["ori_shapes"])
    assert [t for t, _ in s2] == [t for t, _ in structs]
    assert all(np.array_equal(x, y) for x, y in zip(b2, boxes))


def test_table_decode_edge_cases():
    vocab = ["<tr>", "</tr>", "<td>", "</td>"]
    dec = TableStructureDecoder(vocab)
    V = len(dec.character)
    # a table with no cell at all: the reference raises on the empty box list; here it is an empty [0, 8] array
    probs = np.zeros((1, 5, V), np.float32)
    probs[0, :, dec.index["<tr>"]] = 1.0
    probs[0, 2, dec.eos] = 2.0
    structs, boxes = dec.decode(np.ones((1, 5, 8), np.float32), probs, np.array([[488, 488]], np.float32), [np.zeros((100, 200, 3), np.uint8)])
    assert structs[0][0] == ["<html>", "<body>", "<table>", "<tr>", "<tr>", "</table>", "</body>", "</html>"]
    assert boxes[0].shape == (0, 8)
    # `<td>` was merged into `<td></td>` (merge_no_span_structure) exactly like the reference constructor does
    assert "<td>" not in dec.character and "<td></td>" in dec.character
