"""Deterministic synthetic page images for benchmarks and tests (SURVEY.md section 8d recipe): white A4-ish
page 1684x1191, 45 dark textured "text lines", one noise "figure".  Returns the page and the line boxes."""
from __future__ import annotations

from typing import List, Tuple

import numpy as np

PAGE_H, PAGE_W = 1684, 1191


def synth_page(page_idx: int, n_lines: int = 45) -> Tuple[np.ndarray, np.ndarray]:
    rng = np.random.default_rng(page_idx)
    page = np.full((PAGE_H, PAGE_W, 3), 255, dtype=np.uint8)
    # one "figure" block of uniform noise, 500 wide x 400 tall, on the right; the text lines beside it are short
    fy, fx, fh, fw = 1000, 650, 400, 500
    page[fy: fy + fh, fx: fx + fw] = rng.integers(0, 256, size=(fh, fw, 3), dtype=np.uint8)
    boxes = []
    y = 60
    for _ in range(n_lines):
        h = int(rng.integers(24, 33))
        w = int(rng.integers(300, 1051))
        x0 = 90
        if y + h > fy - 4 and y < fy + fh + 4:
            w = min(w, fx - x0 - 20)
        w = min(w, PAGE_W - x0 - 20)
        line = rng.integers(0, 61, size=(h, w, 1), dtype=np.uint8).repeat(3, axis=2)
        # random 2-6 px vertical bars give the det/rec nets texture
        x = 0
        while x < w:
            bw = int(rng.integers(2, 7))
            if rng.random() < 0.5:
                line[:, x: x + bw] = 255 - line[:, x: x + bw] // 4
            x += bw + int(rng.integers(1, 4))
        page[y: y + h, x0: x0 + w] = line
        boxes.append((x0, y, x0 + w, y + h))
        y += 36
    return page, np.asarray(boxes, dtype=np.float32)


def synth_batch(first_idx: int, n: int):
    pages, boxes = [], []
    for i in range(n):
        p, b = synth_page(first_idx + i)
        pages.append(p)
        boxes.append(b)
    return np.stack(pages), boxes


def synth_pages(indices):
    """Pages of a global document list by index (what a rank gets from `dist.shard_pages`)."""
    pages, boxes = [], []
    for i in indices:
        p, b = synth_page(int(i))
        pages.append(p)
        boxes.append(b)
    return np.stack(pages), boxes
