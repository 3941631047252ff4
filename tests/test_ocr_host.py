"""CPU: host-side OCR glue (rapiddoc_amd/ocr_host.py) - known-answer tests minted from the algorithm
definitions at the reference call sites (rapid_ocr.py:404-472, analyze_utils.py:278-292)."""
import numpy as np

from rapiddoc_amd import ocr_host as H


def test_ctc_decode_known_answers():
    chars = H.build_characters(["a\n", "b\n", "c\n"])  # blank a b c ' '
    assert chars == ["blank", "a", "b", "c", " "]
    idx = np.array([[0, 1, 1, 0, 1, 2, 2, 4, 4, 3, 0],
                    [0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0],
                    [3, 3, 3, 3, 3, 3, 3, 3, 3, 3, 3]])
    prob = np.tile(np.linspace(0.1, 1.0, 11, dtype=np.float32), (3, 1))
    out = H.ctc_decode(idx, prob, chars)
    assert out[0][0] == "aab c"
    kept = [1, 4, 5, 7, 9]
    assert abs(out[0][1] - float(np.mean(prob[0][kept]))) < 1e-7
    assert out[1] == ("", 0.0)
    assert out[2][0] == "c" and abs(out[2][1] - 0.1) < 1e-7


def test_score_formatting():
    assert H.format_score(0.98765) == 0.988
    assert H.format_score(0.4994) == 0.499


def test_rec_batching_matches_reference_rule():
    ratios = [10.0, 2.0, 30.0, 6.0, 7.0, 1.0, 12.5]
    b = H.rec_batches(ratios, rec_batch_num=3)
    assert [list(c) for c, _ in b] == [[5, 1, 3], [4, 0, 6], [2]]
    # imgW = int(48 * max(320/48, chunk max))
    assert [w for _, w in b] == [320, int(48 * 12.5), int(48 * 30.0)]
    assert H.rec_resized_width(100, 20, 600) == 240
    assert H.rec_resized_width(1000, 20, 600) == 600
    b32 = H.rec_batches(ratios, rec_batch_num=3, width_multiple=32)
    assert all(w % 32 == 0 for _, w in b32)


def test_det_resize_shape():
    assert H.det_resize_shape(1792, 1344) == (960, 704)   # 64-px bucket of a 1684x1191 page + 50 px margins
    assert H.det_resize_shape(100, 37) == (96, 32)
    assert H.det_resize_shape(640, 480) == (640, 480)


def test_seq_len_matches_oracle_shapes():
    import torch
    from oracle import nets as O
    from rapiddoc_amd import weights as W
    from pathlib import Path
    g = Path(__file__).resolve().parent / "golden"
    st = O.as_torch_state(W.synth_state_dict(W.load_manifest(g / "manifest_ppocrv6_rec.json"), 0))
    for w in (16, 50, 97, 321):
        lg = O.rec_forward(st, torch.zeros(1, 3, 48, w))
        assert lg.shape[1] == H.rec_seq_len(w)


def test_synth_pages_are_deterministic():
    from rapiddoc_amd.pages import synth_page
    a, ba = synth_page(3)
    b, bb = synth_page(3)
    assert (a == b).all() and (ba == bb).all() and a.shape == (1684, 1191, 3) and len(ba) == 45


import json
from pathlib import Path

import pytest


@pytest.mark.parametrize("seed", range(5))
def test_box_helpers_match_reference_golden(seed):
    """sorted_boxes / merge_det_boxes / update_det_boxes / calculate_is_angle vs outputs of the reference functions
    (rapid_doc/utils/ocr_utils.py) captured by tests/golden/make_golden.py."""
    g = json.loads((Path(__file__).resolve().parent / "golden" / f"boxes_seed{seed}.json").read_text())
    quads = [np.array(q, dtype=np.float32) for q in g["quads"]]
    assert [H.quad_is_tilted(q) for q in quads] == g["is_angle"]
    got = [np.asarray(b).tolist() for b in H.sorted_boxes(np.array(quads))]
    assert got == g["sorted"]
    got = [np.asarray(b).tolist() for b in H.merge_det_boxes([q.copy() for q in quads])]
    assert got == g["merged"]
    got = [np.asarray(b).tolist() for b in H.update_det_boxes([q.copy() for q in quads], g["formulas"])]
    assert got == g["updated"]


def test_det_buckets_follow_reference_grouping():
    """analyze_utils.py:150-189: language groups in first-appearance order, then (ceil64 h, ceil64 w) buckets in
    first-appearance order, members in input order, batch = min(len, Det.rec_batch_num)."""
    hw = [(100, 200), (64, 64), (65, 129), (128, 256), (120, 250), (64, 60), (1, 1), (300, 10)]
    langs = ["ch", "en", "ch", "ch", "ch", "en", "en", "ch"]
    got = H.det_buckets(hw, langs, det_batch_num=2)
    assert got == [
        ("ch", (128, 256), [0, 3, 4], 2),
        ("ch", (128, 192), [2], 1),
        ("ch", (320, 64), [7], 1),
        ("en", (64, 64), [1, 5, 6], 2),
    ]
    assert H.det_buckets([], [], 4) == []
    # default Det.rec_batch_num is 1 in the reference (analyze_utils.py:113)
    assert [b for *_, b in H.det_buckets(hw, langs)] == [1, 1, 1, 1]
    img = np.arange(2 * 3 * 3, dtype=np.uint8).reshape(2, 3, 3)
    pad = H.pad_to_bucket(img, (64, 64))
    assert pad.shape == (64, 64, 3) and (pad[:2, :3] == img).all() and pad[2:].min() == 255 and pad[:, 3:].min() == 255



def test_numpy_float32_mean_scheme_is_the_one_the_ctc_kernel_implements():
    """rd_ctc_collapse reproduces np.mean of a float32 vector bit for bit by restating numpy's pairwise summation
    (kernels_image.hip: np_pairwise_sum_f32).  This is the same scheme in Python, pinned against np.mean itself."""
    def pairwise(a):
        n = len(a)
        if n < 8:
            r = np.float32(0)
            for v in a:
                r = np.float32(r + v)
            return r
        if n <= 128:
            r = [np.float32(v) for v in a[:8]]
            i = 8
            while i < n - (n % 8):
                for j in range(8):
                    r[j] = np.float32(r[j] + a[i + j])
                i += 8
            res = np.float32(np.float32(np.float32(r[0] + r[1]) + np.float32(r[2] + r[3])) + np.float32(np.float32(r[4] + r[5]) + np.float32(r[6] + r[7])))
            while i < n:
                res = np.float32(res + a[i])
                i += 1
            return res
        n2 = n // 2
        n2 -= n2 % 8
        return np.float32(pairwise(a[:n2]) + pairwise(a[n2:]))
    rng = np.random.default_rng(0)
    for n in list(range(1, 40)) + [63, 64, 65, 100, 127, 128, 129, 136, 200, 255, 256, 257, 400]:
        for _ in range(6):
            a = rng.uniform(0.05, 1.0, n).astype(np.float32)
            mine = np.float32(np.float32(np.float32(0) + pairwise(a)) / np.float32(n))
            assert mine.tobytes() == np.mean(a).astype(np.float32).tobytes(), (n, mine, np.mean(a))


def test_char_table_and_row_parser_roundtrip():
    from rapiddoc_amd import ocr_host
    chars = ["blank", "a", "\u6587", "\u00e9", " ", "\U0001f600"]
    tab, max_len = ocr_host.char_table(chars)
    assert tab.shape == (6, 1 + max_len) and max_len == 5 and tab[2, 0] == 3 and bytes(tab[5, 1:5]) == "\U0001f600".encode()
    rows = np.zeros((2, 48), np.uint8)
    t = "a\u6587 ".encode()
    rows[0, :4] = np.frombuffer(np.int32(len(t)).tobytes(), np.uint8)
    rows[0, 4:8] = np.frombuffer(np.float32(0.875).tobytes(), np.uint8)
    rows[0, 16:16 + len(t)] = np.frombuffer(t, np.uint8)
    assert ocr_host.parse_ctc_rows(rows) == [("a\u6587 ", 0.875), ("", 0.0)]


@pytest.mark.parametrize("planner", ["dp", "greedy"])
def test_rec_batches_adaptive_partitions_the_sorted_list(planner):
    """Throughput-mode chunking with width-dependent chunk sizes: every line in exactly one chunk, chunks are runs of the aspect-sorted
    order, the padded width is the reference's int(48 * max ratio) of the chunk rounded up to the multiple, sizes stay inside the
    bounds, and the cost model it optimises prefers a chunk that fills whole rounds of the chip (62 lines of width 1056 = 3 rounds of
    C = 192 mixer tiles) over one that spills into a fourth (64 lines)."""
    from rapiddoc_amd import ocr_host as H
    rng = np.random.default_rng(3)
    ratios = rng.uniform(2.0, 44.0, 1440).tolist()
    bat = H.rec_batches_adaptive(ratios, width_multiple=32, planner=planner)
    order = np.concatenate([c for c, _ in bat])
    assert sorted(order.tolist()) == list(range(1440))
    assert np.all(np.diff(np.asarray(ratios)[order]) >= 0)
    for chunk, wpad in bat:
        assert 16 <= len(chunk) <= 160 or chunk is bat[-1][0]
        need = int(48 * max(320 / 48, max(ratios[i] for i in chunk)))
        assert wpad % 32 == 0 and need <= wpad < need + 32
    assert len(bat) < len(H.rec_batches(ratios, 64, width_multiple=32)) + 8
    c62, c64 = float(H.rec_chunk_cost(62, 1056)), float(H.rec_chunk_cost(64, 1056))
    assert c62 / 62 < c64 / 64
    # few lines: one chunk
    assert [len(c) for c, _ in H.rec_batches_adaptive([5.0] * 9, planner=planner)] == [9]
    assert H.rec_batches_adaptive([], planner=planner) == []


def test_rec_chunk_planner_is_optimal_and_matches_the_python_cost_model():
    """rd_rec_plan_chunks (C++ dynamic programme, the default planner): its cost model equals ocr_host.rec_chunk_cost, its plan costs
    no more than the greedy plan or fixed chunks of 64, and on a short list it equals a brute-force search over all compositions."""
    import ctypes as C
    import itertools

    from rapiddoc_amd import _lib
    from rapiddoc_amd import ocr_host as H
    lib = _lib.load()
    rng = np.random.default_rng(11)
    for n, w in zip(rng.integers(1, 200, 200).tolist(), (rng.integers(10, 110, 200) * 32).tolist()):
        assert abs(lib.rd_rec_chunk_cost(n, w, 256) - float(H.rec_chunk_cost(n, w))) < 1e-6
    ratios = rng.uniform(2.0, 44.0, 1440).tolist()
    total = lambda bat: sum(float(H.rec_chunk_cost(len(c), w)) for c, w in bat)
    dp, gr, fx = (H.rec_batches_adaptive(ratios, planner="dp"), H.rec_batches_adaptive(ratios, planner="greedy"), H.rec_batches(ratios, 64, width_multiple=32))
    assert total(dp) <= total(gr) + 1e-6 and total(dp) < total(fx)
    # brute force on 12 lines with sizes {2, 4, 6} (+ a free last chunk)
    wp = np.sort(rng.integers(10, 60, 12) * 32).astype(np.int32)
    sizes = np.zeros(12, np.int32)
    n_out = C.c_int32(0)
    assert lib.rd_rec_plan_chunks(wp.ctypes.data, 12, 2, 6, 2, 256, sizes.ctypes.data, 12, C.byref(n_out)) == 0
    got = sizes[: n_out.value].tolist()
    assert sum(got) == 12

    def cost_of(comp):
        j, c = 0, 0.0
        for s_ in comp:
            j += s_
            c += lib.rd_rec_chunk_cost(s_, int(wp[j - 1]), 256)
        return c
    best = min(cost_of(comp) for k in range(1, 7) for comp in itertools.product(range(1, 7), repeat=k)
               if sum(comp) == 12 and all(s_ in (2, 4, 6) for s_ in comp[:-1]))
    assert abs(cost_of(got) - best) < 1e-6


def test_detector_and_recogniser_settings_are_the_ones_rapiddoc_configures(golden_dir):
    """tests/golden/ocr_default_params.json = the `params` RapidOcrModel.__init__ hands to rapidocr (captured from the reference's own
    constructor by make_golden_ocr_params.py), for the page OCR and the table OCR.  Everything the product hard-codes must equal it."""
    import json
    from rapiddoc_amd import analyze, ocr_host, pipeline
    g = json.loads((golden_dir / "ocr_default_params.json").read_text())
    page, table = g["page"]["params"], g["table"]["params"]
    assert tuple(page["Det.mean"]) == ocr_host.DET_MEAN and tuple(page["Det.std"]) == ocr_host.DET_STD       # NOT rapidocr's 0.5 / 0.5
    assert tuple(table["Det.mean"]) == ocr_host.DET_MEAN and tuple(table["Det.std"]) == ocr_host.DET_STD
    assert page["Det.limit_side_len"] == pipeline.DET_LIMIT == 960 and page["Det.limit_type"] == "max"
    assert page["Det.use_dilation"] is True and page["Global.use_cls"] is False
    import inspect
    sig = inspect.signature(analyze.RegionOcr.__init__).parameters
    assert (sig["box_thresh"].default, sig["unclip_ratio"].default) == (page["Det.box_thresh"], page["Det.unclip_ratio"]) == (0.3, 1.8)
    t = analyze.TableOcr(pipeline=None)
    assert (t.det.box_thresh, t.det.unclip_ratio) == (table["Det.box_thresh"], table["Det.unclip_ratio"]) == (0.5, 1.6)
    assert g["page"]["enable_merge_det_boxes"] is True and g["table"]["enable_merge_det_boxes"] is False
    assert page["Rec.rec_keys_path"] == "ppocrv6_small_dict.txt" and page["Det.model_path"] == "ch_PP-OCRv6_det_small.safetensors"


def test_degenerate_quads_are_skipped_not_warped():
    """quads_to_crop_matrices: a quad with three (nearly) collinear corners has no usable homography - `ok` is False for it at
    any coordinate scale (the reference's cv2.warpPerspective raises there), and well-formed neighbours are untouched."""
    from rapiddoc_amd.pipeline import quads_to_crop_matrices
    good = np.array([[100, 200], [900, 200], [900, 232], [100, 232]], np.float32)
    collinear = np.array([[100, 200], [500, 200], [900, 200], [100, 232]], np.float32)
    nearly = np.array([[100, 200], [500, 200.0001], [900, 200], [100, 232]], np.float32)
    repeated = np.array([[100, 200], [100, 200], [900, 232], [100, 232]], np.float32)
    skewed = np.array([[100, 200], [900, 210], [905, 242], [98, 236]], np.float32)
    mats, cw, ch, ok = quads_to_crop_matrices(np.stack([good, collinear, nearly, repeated, skewed, good * 0.1]))
    assert ok.tolist() == [True, False, False, False, True, True]
    assert np.allclose(mats[0], [1, 0, 100, 0, 1, 200, 0, 0, 1]) and (cw[0], ch[0]) == (800.0, 32.0)
    assert np.isfinite(mats).all()


def test_rec_batches_lines_keeps_reference_widths_in_gpu_sized_launches():
    """ocr_host.rec_batches_lines: the launches are runs of the reference's sorted order, every line carries the imgW of its own
    reference chunk of six (rapid_ocr.py:411-440), a launch is as wide as its widest line rounded up to 32."""
    rng = np.random.default_rng(5)
    for n in (1, 5, 6, 7, 95, 1440):
        ratios = np.concatenate([rng.uniform(0.3, 6.0, n // 3), rng.uniform(6.0, 40.0, n - n // 3)]).tolist()
        ref = H.rec_batches(ratios, 6, strict=True)
        launches, line_w = H.rec_batches_lines(ratios)
        order = np.concatenate([c for c, _w in launches])
        assert order.tolist() == np.concatenate([c for c, _w in ref]).tolist()
        assert line_w.tolist() == [w for c, w in ref for _ in c]
        assert all(w >= 320 for w in line_w.tolist()) and (np.diff(line_w) >= 0).all()
        pos = 0
        for c, W_launch in launches:
            ws = line_w[pos: pos + len(c)]
            assert W_launch % 32 == 0 and W_launch == (int(ws.max()) + 31) // 32 * 32
            pos += len(c)
        if n >= 95:
            assert len(launches) < len(ref) / 2 and max(len(c) for c, _w in launches) <= 160
    empty = H.rec_batches_lines([])
    assert empty[0] == [] and len(empty[1]) == 0
