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
