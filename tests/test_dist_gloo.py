"""CPU, world_size 2 (gloo): page sharding + result gather used by bench.py --gpus N."""
import os
import sys
from pathlib import Path

import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from rapiddoc_amd.dist import gather_page_results, shard_pages
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    mine = shard_pages(7, rank, world)
    local = [(i, [("text-%d-%d" % (i, j), 0.5 + 0.01 * j) for j in range(i % 3 + (rank == 1) * 40)]) for i in mine]
    merged = gather_page_results(local, dist)
    q.put((rank, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_world2():
    world, port = 2, 29731
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1]
    assert [i for i, _ in got[0]] == list(range(7))
    for i, lines in got[0]:
        owner = i % world
        assert len(lines) == i % 3 + (owner == 1) * 40
        assert all(t == "text-%d-%d" % (i, j) for j, (t, _) in enumerate(lines))


def test_shard_pages_partition():
    from rapiddoc_amd.dist import shard_pages
    for world in (1, 2, 4, 8):
        seen = sorted(i for r in range(world) for i in shard_pages(37, r, world))
        assert seen == list(range(37))


def test_wire_format_roundtrip_and_rejects_garbage():
    import pytest
    from rapiddoc_amd.dist import decode_page_results, encode_page_results
    pages = [(3, [("\u6587\u5b57 abc", 0.987), ("", 0.0)]), (10 ** 12, []), (0, [("x" * 3000, 1.0)])]
    blob = encode_page_results(pages)
    assert decode_page_results(blob) == pages
    assert decode_page_results(encode_page_results([])) == []
    with pytest.raises(Exception):
        decode_page_results(blob[:-3])
    with pytest.raises(ValueError):
        decode_page_results(blob + b"\0")


# ---------------------------------------------------------------------------------------------------------------------
# wire format v2: whole per-page results (PageAnalyzer's / BatchAnalyze's List[List[dict]])
# ---------------------------------------------------------------------------------------------------------------------
def _reference_page_outputs():
    """Per-page `layout_dets` lists the REFERENCE's BatchAnalyze returned (tests/golden/analyze_trace_*.json "output"): text spans,
    formulas with latex, tables with html / formula_boxes / img_boxes, polygons - every value type the product's results carry."""
    import json
    pages = []
    g = ROOT / "tests" / "golden"
    for name in ("analyze_trace_seed0", "analyze_trace_seed3", "analyze_trace_table_traditional", "analyze_trace_table_custom"):
        pages += json.loads((g / f"{name}.json").read_text())["output"]
    return pages


def _worker_v2(rank, world, port, q):
    sys.path.insert(0, str(ROOT))
    import torch.distributed as dist
    from rapiddoc_amd.dist import gather_page_dets, shard_pages
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pages = _reference_page_outputs()
    local = [(i, pages[i]) for i in shard_pages(len(pages), rank, world)]
    merged = gather_page_dets(local, dist)
    q.put((rank, merged))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_whole_page_results_world2_equals_single_rank():
    """Rank-merged output == the single-rank output, dict for dict (VERDICT r3 missing #4; pipeline_analyze.py:221-228)."""
    from rapiddoc_amd.dist import gather_page_dets
    pages = _reference_page_outputs()
    assert len(pages) >= 6 and any("html" in d for p in pages for d in p) and any("latex" in d for p in pages for d in p)
    single = gather_page_dets([(i, p) for i, p in enumerate(pages)])
    assert [i for i, _ in single] == list(range(len(pages))) and [p for _, p in single] == pages
    world, port = 2, 29741
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_v2, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got[0] == got[1] == single


def test_wire_format_v2_roundtrip_types_and_garbage():
    import numpy as np
    import pytest
    from rapiddoc_amd.dist import decode_page_dets, encode_page_dets
    det = {"category_id": 15, "poly": [1, 2.5, 3, 4, 5, 6, 7, 8], "score": 0.987, "text": "文字 abc", "latex": "", "html": None,
           "spans": [{"bbox": (1, 2, 3, 4), "ok": True, "no": False, "nested": [[1.0, 2], []]}], 7: "int key",
           "mask": np.arange(12, dtype=np.uint8).reshape(3, 4), "f32": np.float32(0.1), "i64": np.int64(-5), "big": -10 ** 30,
           "raw": b"\x00\xff", "empty": {}, "polygon_points": np.zeros((0, 2), np.float32)}
    pages = [(10 ** 12, [det, {}]), (0, []), (5, None)]
    back = decode_page_dets(encode_page_dets(pages))
    assert [i for i, _ in back] == [10 ** 12, 0, 5] and back[1][1] == [] and back[2][1] is None
    d = back[0][1][0]
    assert d["spans"] == det["spans"] and isinstance(d["spans"][0]["bbox"], tuple) and d["spans"][0]["ok"] is True
    assert d["poly"] == det["poly"] and [type(v) for v in d["poly"]] == [type(v) for v in det["poly"]]
    assert d[7] == "int key" and d["big"] == -10 ** 30 and d["raw"] == b"\x00\xff" and d["empty"] == {} and d["html"] is None
    assert d["f32"] == float(np.float32(0.1)) and d["i64"] == -5 and isinstance(d["i64"], int)
    assert d["mask"].dtype == np.uint8 and np.array_equal(d["mask"], det["mask"]) and d["polygon_points"].shape == (0, 2)
    assert list(d) == list(det)                               # key order survives
    blob = encode_page_dets(pages)
    for bad in (blob[:-1], blob + b"\0", b"XXXX" + blob[4:], blob[:8] + b"?" * 20):
        with pytest.raises(ValueError):
            decode_page_dets(bad)
    with pytest.raises(TypeError):
        encode_page_dets([(0, [{"x": object()}])])
    with pytest.raises(ValueError):                            # a length field that promises more than the blob holds
        decode_page_dets(blob[:8] + __import__("struct").pack("<q", 0) + b"l" + __import__("struct").pack("<I", 2 ** 31))


def test_wire_format_v2_malformed_peer_bytes_raise_value_error_only():
    """whatever bytes a peer sends, the v2 decoder's contract is ValueError - not TypeError (an unhashable dictionary key,
    a junk dtype string), UnicodeDecodeError, or a CPU-burning integer conversion."""
    import struct
    import numpy as np
    import pytest
    from rapiddoc_amd.dist import _MAGIC2, decode_page_dets, encode_page_dets
    head = _MAGIC2 + struct.pack("<I", 1) + struct.pack("<q", 0)
    s = lambda b: b"s" + struct.pack("<I", len(b)) + b
    bad = [
        head + b"m" + struct.pack("<I", 1) + b"l" + struct.pack("<I", 0) + b"N",                       # a list as dictionary key
        head + b"m" + struct.pack("<I", 1) + b"m" + struct.pack("<I", 0) + b"N",                       # a dict as dictionary key
        head + b"m" + struct.pack("<I", 1) + b"t" + struct.pack("<I", 1) + b"l" + struct.pack("<I", 0) + b"N",   # tuple holding a list as key
        head + b"a" + struct.pack("<B", 4) + b"junk" + struct.pack("<B", 0),                           # junk dtype string
        head + b"a" + struct.pack("<B", 2) + b"O8" + struct.pack("<B", 0),                             # object dtype
        head + b"a" + struct.pack("<B", 3) + b"<U4" + struct.pack("<B", 1) + struct.pack("<q", 1) + b"\0" * 16,   # string dtype
        head + b"a" + struct.pack("<B", 3) + b"<M8" + struct.pack("<B", 0) + b"\0" * 8,                # datetime dtype
        head + b"a" + struct.pack("<B", 3) + b"<f4" + struct.pack("<B", 3) + struct.pack("<3q", 2 ** 40, 2 ** 40, 2 ** 40),   # huge shape
        head + s(b"\xff\xfe"),                                                                         # not UTF-8
        head + b"I" + struct.pack("<I", 5000) + b"9" * 5000,                                           # a 5000-digit integer
        head + b"I" + struct.pack("<I", 3) + b"1x2",                                                   # not an integer
    ]
    for blob in bad:
        with pytest.raises(ValueError):
            decode_page_dets(blob)
    # what the encoder itself writes still passes: complex and bool arrays, tuple keys
    ok = [(3, {(1, "a"): np.array([1 + 2j], dtype=np.complex128), "b": np.array([True, False])})]
    back = decode_page_dets(encode_page_dets(ok))
    assert back[0][0] == 3 and back[0][1][(1, "a")].dtype == np.complex128 and back[0][1]["b"].dtype == np.bool_
