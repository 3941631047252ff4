"""CPU, two gloo ranks: dist.GlobalLineWidths gives every rank, for its own lines, exactly the padded width and max_wh_ratio the
reference's text_recognizer_call assigns them when it pools the lines of the WHOLE page batch (one np.argsort, chunks of 6,
imgW = int(48 * max ratio of the chunk): rapid_ocr.py:404-449) - checked against the chunks the reference's own loop produced
(tests/golden/rec_batching.json, minted by make_golden_recbatch.py) with the crops dealt to the ranks page by page."""
import json
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
GOLD = ROOT / "tests" / "golden" / "rec_batching.json"


def _cases():
    return json.loads(GOLD.read_text())["cases"]


def _reference_widths(case):
    """Per crop (in pooled order): the padded width of ITS chunk as the reference computed it."""
    # txts[i] = "L<k>": line i was the k-th crop the reference handed to the recogniser, i.e. member k % 6 of chunk k // 6
    n = len(case["crop_hw"])
    w = np.zeros(n, np.int64)
    r = np.zeros(n, np.float64)
    for i, name in enumerate(case["txts"]):
        ch = case["chunks"][int(name[1:]) // case["rec_batch_num"]]
        w[i], r[i] = int(ch["imgW"]), float(ch["max_wh_ratio"])
    return w, r


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from rapiddoc_amd.dist import GlobalLineWidths
    sync = GlobalLineWidths(dist)
    out = []
    for case in _cases():
        sync.rec_batch_num = case["rec_batch_num"]
        ratios = np.array([w / float(h) for h, w in case["crop_hw"]])
        n = len(ratios)
        # pages of 7 lines, page p on rank p % world: the pooled order is page-major, a page's lines in their own order
        page = np.arange(n) // 7
        mine = np.nonzero(page % world == rank)[0]
        w, r = sync(page[mine], ratios[mine])
        out.append((mine.tolist(), w.tolist(), r.tolist()))
    # a rank without any line still takes part
    sync.rec_batch_num = 6
    w, r = sync(np.zeros(0, np.int64) if rank == 1 else np.array([0, 0, 0]), np.zeros(0) if rank == 1 else np.array([1.0, 9.5, 9.5]))
    out.append(([], w.tolist(), r.tolist()))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


def test_golden_file_has_what_the_test_needs():
    cases = _cases()
    assert len(cases) >= 3 and all({"crop_hw", "chunks", "txts", "rec_batch_num"} <= set(c) for c in cases)
    assert all({"imgW", "max_wh_ratio"} <= set(c["chunks"][0]) for c in cases)
    assert any(len(c["crop_hw"]) >= 1000 for c in cases)          # the 1440-crop case with tied ratios is among them


def test_single_process_widths_equal_the_reference_chunks():
    from rapiddoc_amd.dist import GlobalLineWidths
    sync = GlobalLineWidths(None)
    for case in _cases():
        sync.rec_batch_num = case["rec_batch_num"]
        ratios = np.array([w / float(h) for h, w in case["crop_hw"]])
        w, r = sync(np.arange(len(ratios)) // 7, ratios)
        rw, rr = _reference_widths(case)
        assert np.array_equal(w, rw) and np.array_equal(r, rr)


def test_two_ranks_get_the_global_widths():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 400)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cases = _cases()
    for ci, case in enumerate(cases):
        rw, rr = _reference_widths(case)
        seen = 0
        for rank in (0, 1):
            mine, w, r = got[rank][ci]
            assert np.array_equal(np.asarray(w), rw[mine]) and np.array_equal(np.asarray(r), rr[mine]), (ci, rank)
            seen += len(mine)
        assert seen == len(rw)
    # the extra call: rank 0's three lines are one chunk padded to int(48 * 9.5) = 456, rank 1 holds nothing
    assert got[0][-1][1] == [456, 456, 456] and got[1][-1][1] == []
