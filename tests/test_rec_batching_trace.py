"""CPU: `ocr_host.rec_batches(strict=True)` replays the REFERENCE's own recogniser batching loop.

tests/golden/rec_batching.json was recorded by running RapidDoc's `RapidOcrModel.text_recognizer_call`
(rapid_doc/model/ocr/rapid_ocr.py:404-471) on seeded crop lists with a recording stand-in for the rapidocr recogniser object
(tests/golden/make_golden_recbatch.py): which crops it put into which chunk, in which order, and the `max_wh_ratio` it handed to
`resize_norm_img` for every chunk.  The strict mode of this repo must cut the same lists the same way: same chunks, same crop order
inside a chunk (the global `np.argsort` with numpy's default kind, ties included), same padded width, and the position at which
every line's result is scattered back."""
import json

import numpy as np
import pytest

from rapiddoc_amd import ocr_host


def _cases(golden_dir):
    return json.loads((golden_dir / "rec_batching.json").read_text())["cases"]


@pytest.mark.parametrize("idx", [0, 1, 2, 3, 4])
def test_strict_rec_batches_equal_the_reference_loop(golden_dir, idx):
    case = _cases(golden_dir)[idx]
    hw = case["crop_hw"]
    ratios = [w / float(h) for h, w in hw]
    bat = ocr_host.rec_batches(ratios, case["rec_batch_num"], strict=True)
    ref = case["chunks"]
    assert len(bat) == len(ref)
    same_numpy = np.__version__ == case["numpy"]          # np.argsort's default kind is an implementation detail of the numpy build
    k = 0
    for (chunk, wpad), rc in zip(bat, ref):
        assert wpad == rc["imgW"] == rc["batch_shape"][3] and len(chunk) == rc["batch_shape"][0] == len(rc["crops"])
        assert abs(max(320 / 48, max(ratios[i] for i in chunk)) - rc["max_wh_ratio"]) < 1e-12
        got = [hw[i] for i in chunk.tolist()]
        if same_numpy:
            assert got == rc["crops"]                      # the very crops, in the very order (ties at chunk borders included)
        else:
            assert sorted(w / h for h, w in got) == pytest.approx(sorted(w / h for h, w in rc["crops"]))
        # the reference writes chunk position j of chunk c back to rec_res[indices[beg + j]]: txts[i] names the call position of line i
        if same_numpy:
            for j, i in enumerate(chunk.tolist()):
                assert case["txts"][i] == f"L{k + j}"
        k += len(chunk)
    assert k == len(hw)


def test_merged_equal_width_chunks_keep_the_reference_chunks_per_line(golden_dir):
    """`merge_equal_width=True` (what PagePipeline's strict mode launches) only concatenates chunks of equal padded width: every line
    keeps the padded width the reference gave it."""
    case = _cases(golden_dir)[4]
    hw = case["crop_hw"]
    ratios = [w / float(h) for h, w in hw]
    plain = ocr_host.rec_batches(ratios, 6, strict=True)
    merged = ocr_host.rec_batches(ratios, 6, strict=True, merge_equal_width=True)
    w_of = {}
    for chunk, wpad in plain:
        for i in chunk.tolist():
            w_of[i] = wpad
    assert len(merged) < len(plain) and sum(len(c) for c, _ in merged) == len(hw)
    assert len({w for _, w in merged}) == len(merged)
    for chunk, wpad in merged:
        assert all(w_of[i] == wpad for i in chunk.tolist())
