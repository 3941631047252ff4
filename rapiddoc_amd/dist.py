"""Page-sharded data parallelism: pages are independent units, so ranks never exchange
activations or weights.  The only collective is the result reassembly: ragged per-page results are
serialised, lengths all-gathered, then one padded uint8 all-gather (RCCL over xGMI on GPU, gloo on CPU).
The payload is KBs per page - latency-bound, a single collective per page batch."""
from __future__ import annotations

import pickle
from typing import Any, List, Optional, Sequence, Tuple

import numpy as np
import torch


def shard_pages(n_pages: int, rank: int, world: int) -> List[int]:
    """Interleaved assignment (page i -> rank i % world) so long documents balance across ranks."""
    return list(range(rank, n_pages, world))


def gather_page_results(local: Sequence[Tuple[int, Any]], dist=None, device: Optional[torch.device] = None) -> List[Tuple[int, Any]]:
    """local: [(global_page_idx, result)] of this rank -> the full list sorted by page index, on every rank."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return sorted(local, key=lambda t: t[0])
    world = dist.get_world_size()
    backend = dist.get_backend()
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    blob = pickle.dumps(list(local), protocol=pickle.HIGHEST_PROTOCOL)
    n = torch.tensor([len(blob)], dtype=torch.int64, device=device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    buf = torch.zeros(cap, dtype=torch.uint8, device=device)
    buf[: len(blob)] = torch.frombuffer(bytearray(blob), dtype=torch.uint8).to(device)
    out = [torch.empty(cap, dtype=torch.uint8, device=device) for _ in range(world)]
    dist.all_gather(out, buf)
    merged: List[Tuple[int, Any]] = []
    for r in range(world):
        merged.extend(pickle.loads(out[r][: sizes[r]].cpu().numpy().tobytes()))
    return sorted(merged, key=lambda t: t[0])
