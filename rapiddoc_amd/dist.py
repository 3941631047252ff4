"""Page-sharded data parallelism: pages are independent units, so ranks never exchange
activations or weights.  The only collective is the result reassembly: ragged per-page results are
serialised, lengths all-gathered, then one padded uint8 all-gather (RCCL over xGMI on GPU, gloo on CPU).
The payload is KBs per page - latency-bound, a single collective per page batch.

Wire format (flat little-endian bytes, no pickle - a peer's bytes are data, never code):
    u32 n_pages, then per page: i64 global page index, u32 n_lines, then per line: f64 score, u32 n_utf8, utf-8 bytes."""
from __future__ import annotations

import struct
from typing import List, Optional, Sequence, Tuple

import torch

PageLines = List[Tuple[str, float]]


def shard_pages(n_pages: int, rank: int, world: int) -> List[int]:
    """Interleaved assignment (page i -> rank i % world) so long documents balance across ranks."""
    return list(range(rank, n_pages, world))


def encode_page_results(local: Sequence[Tuple[int, PageLines]]) -> bytes:
    out = [struct.pack("<I", len(local))]
    for idx, lines in local:
        out.append(struct.pack("<qI", int(idx), len(lines)))
        for text, score in lines:
            b = str(text).encode("utf-8")
            out.append(struct.pack("<dI", float(score), len(b)))
            out.append(b)
    return b"".join(out)


def decode_page_results(blob: bytes) -> List[Tuple[int, PageLines]]:
    view = memoryview(blob)
    (n,), pos = struct.unpack_from("<I", view, 0), 4
    pages = []
    for _ in range(n):
        idx, nl = struct.unpack_from("<qI", view, pos)
        pos += 12
        lines = []
        for _ in range(nl):
            score, nb = struct.unpack_from("<dI", view, pos)
            pos += 12
            lines.append((bytes(view[pos:pos + nb]).decode("utf-8"), score))
            pos += nb
        pages.append((idx, lines))
    if pos != len(blob):
        raise ValueError("page-result blob: trailing bytes")
    return pages


def gather_page_results(local: Sequence[Tuple[int, PageLines]], dist=None, device: Optional[torch.device] = None) -> List[Tuple[int, PageLines]]:
    """local: [(global_page_idx, [(text, score), ...])] of this rank -> the full list sorted by page index, on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(((int(i), [(str(t), float(s)) for t, s in l]) for i, l in local), key=lambda t: t[0])
    world = dist.get_world_size()
    backend = dist.get_backend()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    blob = encode_page_results(local)
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    buf[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    out = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(out, buf)
    merged: List[Tuple[int, PageLines]] = []
    for r in range(world):
        merged.extend(decode_page_results(out[r][: sizes[r]].cpu().numpy().tobytes()))
    return sorted(merged, key=lambda t: t[0])
