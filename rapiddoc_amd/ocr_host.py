"""Host-side glue of the OCR path: the arithmetic the reference leaves to the third-party `rapidocr`
package (pinned >=3.4.0,<=3.9.0 in the reference's pyproject.toml:38, not vendored) restated from its call
sites in the reference and from the public PaddleOCR definitions of the same steps.

 * det pre-processing geometry ........ rapid_doc/model/ocr/rapid_ocr.py:517-518 (DetPreProcess, limit 960 'max')
 * rec batching / resize geometry ..... rapid_doc/model/ocr/rapid_ocr.py:404-472 (text_recognizer_call)
 * CTC greedy decode .................. rapid_doc/model/ocr/rapid_ocr.py:444-449 (CTCLabelDecode)
 * score formatting ................... rapid_doc/backend/pipeline/analyze_utils.py:278-292
"""
from __future__ import annotations

import math
from typing import List, Optional, Sequence, Tuple

import numpy as np

REC_IMG_H = 48
REC_IMG_W = 320


def rec_seq_len(W: int) -> int:
    """Time steps of PP-OCRv6 rec for input width W (== rd_rec_seq_len)."""
    if W < 16:
        return 0
    w1 = (W - 1) // 2 + 1
    w2 = (w1 - 1) // 2 + 1
    return (w2 - 2) // 2 + 1


def det_resize_shape(h: int, w: int, limit_side_len: int = 960, limit_type: str = "max") -> Tuple[int, int]:
    """PaddleOCR DetResizeForTest: scale so the max (or min) side meets the limit, then round to x32."""
    if limit_type == "max":
        ratio = float(limit_side_len) / max(h, w) if max(h, w) > limit_side_len else 1.0
    else:
        ratio = float(limit_side_len) / min(h, w) if min(h, w) < limit_side_len else 1.0
    rh, rw = int(h * ratio), int(w * ratio)
    rh = max(int(round(rh / 32) * 32), 32)
    rw = max(int(round(rw / 32) * 32), 32)
    return rh, rw


# Normalisation of the detector's input.  rapidocr's own default is mean = std = 0.5; RapidDoc overrides it with the ImageNet constants
# the PP-OCR detectors were trained with (`"Det.mean"` / `"Det.std"` of RapidOcrModel's default_params, rapid_doc/model/ocr/rapid_ocr.py:
# 61-62, read by the patched TextDetector.__init__, ocr_patch.py:139-142, and handed to DetPreProcess).  They are applied to the channels
# in the order the image has at that point - BGR - exactly as listed (B gets 0.485), like the reference does.
DET_MEAN = (0.485, 0.456, 0.406)
DET_STD = (0.229, 0.224, 0.225)


def det_buckets(region_hw: Sequence[Tuple[int, int]], langs: Sequence[str], det_batch_num: int = 1, stride: int = 64):
    """Grouping of the text-region crops for batched detection (`_run_ocr_det_batch`,
    rapid_doc/backend/pipeline/analyze_utils.py:150-189): first by language in order of first appearance, then by the
    crop size rounded UP to multiples of 64 (first-appearance order again, insertion order inside a group); every group is
    padded with 255 to its (H64, W64) and handed to `det_batch_predict` with batch = min(len(group), Det.rec_batch_num).
    Returns [(lang, (H64, W64), [region indices], batch_size)].  The engine takes a whole group as one batch tensor
    (`PagePipeline.det_forward` does this for full pages); `batch_size` is what the reference would have used."""
    by_lang = {}
    for i, lang in enumerate(langs):
        by_lang.setdefault(lang, []).append(i)
    out = []
    for lang, idxs in by_lang.items():
        groups = {}
        for i in idxs:
            h, w = int(region_hw[i][0]), int(region_hw[i][1])
            key = (-(-h // stride) * stride, -(-w // stride) * stride)
            groups.setdefault(key, []).append(i)
        for key, members in groups.items():
            out.append((lang, key, members, min(len(members), int(det_batch_num))))
    return out


def pad_to_bucket(img: np.ndarray, bucket_hw: Tuple[int, int]) -> np.ndarray:
    """White (255) bottom/right padding of one crop to its bucket (analyze_utils.py:180-184)."""
    h, w = img.shape[:2]
    out = np.full((bucket_hw[0], bucket_hw[1], 3), 255, dtype=np.uint8)
    out[:h, :w] = img
    return out


def build_characters(dict_lines: Sequence[str], use_space_char: bool = True) -> List[str]:
    """['blank'] + dictionary + [' ']  (class 0 is the CTC blank; reference SURVEY appendix A.2)."""
    chars = [ln.rstrip("\r\n") for ln in dict_lines]
    if use_space_char:
        chars.append(" ")
    return ["blank"] + chars


def load_characters(path: str) -> List[str]:
    with open(path, "r", encoding="utf-8") as f:
        return build_characters(f.readlines())


def ctc_decode(idx: np.ndarray, prob: np.ndarray, characters: Sequence[str]) -> List[Tuple[str, float]]:
    """Greedy CTC decode of per-step (argmax, max-prob): collapse repeats, drop blank (0), confidence = mean
    of the kept max-probabilities (0 when nothing is kept)."""
    idx = np.asarray(idx)
    prob = np.asarray(prob)
    out = []
    for b in range(idx.shape[0]):
        row = idx[b]
        sel = np.ones(len(row), dtype=bool)
        sel[1:] = row[1:] != row[:-1]
        sel &= row != 0
        conf = prob[b][sel]
        text = "".join(characters[int(i)] for i in row[sel])
        out.append((text, float(np.mean(conf)) if conf.size else 0.0))
    return out


def char_table(characters: Sequence[str]):
    """Dictionary as the byte table `rd_ctc_collapse` reads: [n_classes][1 + max_len] uint8 = (UTF-8 length, bytes)."""
    enc = [c.encode("utf-8") for c in characters]
    max_len = max(1, max(len(b) for b in enc))
    tab = np.zeros((len(enc), 1 + max_len), np.uint8)
    for i, b in enumerate(enc):
        tab[i, 0] = len(b)
        tab[i, 1:1 + len(b)] = np.frombuffer(b, np.uint8)
    return tab, max_len


def parse_ctc_rows(rows: np.ndarray) -> List[Tuple[str, float]]:
    """Rows written by `rd_ctc_collapse` (int32 n_text_bytes, float32 confidence, int32 n_kept, int32 0, UTF-8 text) ->
    [(text, confidence)] - what `ctc_decode` returns for the same (idx, prob)."""
    rows = np.ascontiguousarray(rows)
    head = rows[:, :8].copy()
    nbytes = head[:, :4].view("<i4")[:, 0]
    conf = head[:, 4:8].view("<f4")[:, 0]
    return [(rows[b, 16:16 + int(nbytes[b])].tobytes().decode("utf-8"), float(conf[b])) for b in range(rows.shape[0])]


def format_score(score: float) -> float:
    """analyze_utils.py:280: float(f'{score:.3f}')"""
    return float(f"{score:.3f}")


def rec_batches(wh_ratios: Sequence[float], rec_batch_num: int = 6, img_h: int = REC_IMG_H, img_w: int = REC_IMG_W,
                width_multiple: int = 1, strict: bool = False, merge_equal_width: bool = False) -> List[Tuple[np.ndarray, int]]:
    """Reference batching (rapid_ocr.py:411-440): argsort by w/h, chunks of rec_batch_num, every chunk padded to
    imgW = int(img_h * max(img_w/img_h, chunk max ratio)).  Returns [(indices into the input, padded width)].
    `width_multiple` > 1 rounds the padded width up (bounds the number of distinct shapes on the GPU).

    `strict`: the sort is the very call the reference makes, `np.argsort(np.array(width_list))` with numpy's DEFAULT kind
    (rapid_ocr.py:414; introsort / the SIMD sort of the running numpy build - equal ratios at a chunk border may land in
    either chunk, exactly as they would in the reference on the same machine); otherwise a stable sort.
    `merge_equal_width`: chunks with the same padded width become one launch.  A line's logits depend on the line and on
    its padded width only (every layer is per sample; LightSVTR attends over the line's own padded columns), so the merge
    changes the launch count, not the result."""
    ratios = np.array([float(r) for r in wh_ratios])
    order = np.argsort(ratios) if strict else np.argsort(ratios, kind="stable")
    out: List[Tuple[np.ndarray, int]] = []
    for beg in range(0, len(order), rec_batch_num):
        chunk = order[beg: beg + rec_batch_num]
        max_ratio = max(img_w / img_h, max(wh_ratios[i] for i in chunk))
        wpad = int(img_h * max_ratio)
        if width_multiple > 1:
            wpad = (wpad + width_multiple - 1) // width_multiple * width_multiple
        out.append((chunk, wpad))
    if merge_equal_width:
        by_w: dict = {}
        for chunk, wpad in out:          # first-appearance order of the widths, chunk order inside a width
            by_w.setdefault(wpad, []).append(chunk)
        out = [(np.concatenate(chunks), wpad) for wpad, chunks in by_w.items()]
    return out


# Round-quantisation model of one recogniser chunk on a 256-CU MI355X (microseconds): the persistent kernels of the
# backbone run whole ROUNDS of workgroup tiles - a chunk whose tile count is 3.09 x 256 costs four rounds.  Per layer family:
# (rounds per unit of work, microseconds per round), measured at 64 x 48 x 1056 (tools/op_profile.py).
_CHUNK_FIXED_US = 150.0           # ~59 launches of a backbone forward
_CHUNK_LINEAR_US = 0.14           # per line and token column: depthwise convs, stem, pools (no tile quantisation to speak of)


def rec_chunk_cost(n, wpad, n_cu: int = 256):
    """Estimated GPU time (us) of a recogniser backbone forward on `n` lines padded to width `wpad` (numpy arrays broadcast)."""
    n = np.asarray(n, dtype=np.float64)
    t = np.asarray(wpad, dtype=np.float64) // 8                        # token columns per line row
    px96, px192, px384 = n * 24 * t, n * 12 * t, n * 6 * t            # pixels at the C = 96 / 192 / 384 stages
    rnd = lambda x: np.ceil(x / n_cu - 1e-9)
    mt384, mt192 = np.ceil(px384 / 256), np.ceil(px192 / 256)          # 256-row GEMM tiles
    c = 6 * 37.0 * rnd(px192 / 128)                                    # six C = 192 weight-streaming mixers, 128-pixel tiles
    c = c + 3 * 21.0 * np.ceil(px96 / 16 / (16 * n_cu) - 1e-9)         # three C = 96 resident mixers, 16-pixel wavefront tiles
    c = c + (2 * 51.0 + 28.0 + 19.0) * rnd(mt384 * 3)                  # N = 384 GEMMs of the C = 384 blocks (256 x 128 tiles)
    c = c + 2 * 27.0 * rnd(mt384 * 6)                                  # N = 768
    c = c + (17.0 + 13.0) * rnd(mt192 * 2)                             # N = 192 at the C = 192 stage
    return c + _CHUNK_LINEAR_US * n * t + _CHUNK_FIXED_US


def rec_batches_adaptive(wh_ratios: Sequence[float], img_h: int = REC_IMG_H, img_w: int = REC_IMG_W, width_multiple: int = 32,
                         n_min: int = 16, n_max: int = 160, n_step: int = 2, n_cu: int = 256, planner: str = "dp") -> List[Tuple[np.ndarray, int]]:
    """Throughput-mode chunking of the aspect-sorted line list with chunk SIZES chosen so that the persistent kernels' tile counts
    fill whole rounds of the chip (`rec_chunk_cost`).  `planner="dp"`: the cut positions that minimise the summed cost, a dynamic
    programme in the library (`rd_rec_plan_chunks`, ~1 ms for 1440 lines); `"greedy"`: chunk by chunk the size with the most lines per
    estimated microsecond (pure numpy).  Same return format as `rec_batches`; every line is in exactly one chunk, chunks are runs
    of the sorted order, the padded width is the reference's `int(img_h * max_ratio)` of the chunk rounded up to `width_multiple`."""
    import os
    planner = os.environ.get("RD_REC_PLANNER", planner)       # (developer A/B switch)
    ratios = np.array([float(r) for r in wh_ratios])
    order = np.argsort(ratios, kind="stable")
    rs = ratios[order]
    wp = (img_h * np.maximum(img_w / img_h, rs)).astype(np.int64)
    if width_multiple > 1:
        wp = (wp + width_multiple - 1) // width_multiple * width_multiple
    total = len(rs)
    if total == 0:
        return []
    if planner == "dp":
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        w32 = np.ascontiguousarray(wp, dtype=np.int32)
        sizes = np.zeros(total, dtype=np.int32)
        n_out = C.c_int32(0)
        rc = lib.rd_rec_plan_chunks(w32.ctypes.data, total, n_min, n_max, n_step, n_cu, sizes.ctypes.data, total, C.byref(n_out))
        if rc != 0:
            raise RuntimeError("rd_rec_plan_chunks failed")
        out, i = [], 0
        for n in sizes[: n_out.value].tolist():
            out.append((order[i: i + n], int(wp[i + n - 1])))
            i += n
        return out
    out: List[Tuple[np.ndarray, int]] = []
    i = 0
    cand = np.arange(n_min, n_max + 1, n_step)
    while i < total:
        left = total - i
        if left <= n_min:
            n = left
        else:
            ns = np.minimum(cand, left)
            cost = rec_chunk_cost(ns, wp[i + ns - 1], n_cu)
            n = int(ns[int(np.argmax(ns / cost))])
            if 0 < left - n < n_min:          # do not leave a sliver behind
                n = left if left <= n_max else left - n_min
        out.append((order[i: i + n], int(wp[i + n - 1])))
        i += n
    return out


def rec_reference_widths(wh_ratios: Sequence[float], rec_batch_num: int = 6, img_h: int = REC_IMG_H, img_w: int = REC_IMG_W):
    """Per line of the POOLED list (in its pooled order): the padded width the reference recognises it at - the imgW of its chunk of
    `rec_batch_num` lines of the one global `np.argsort` (rapid_ocr.py:411-440) - and that chunk's `max_wh_ratio` (what CTCLabelDecode
    scales the line's time steps by).  -> (int64 [n], float64 [n])."""
    n = len(wh_ratios)
    line_w, line_ratio = np.zeros(n, np.int64), np.zeros(n, np.float64)
    for c, w in rec_batches(wh_ratios, rec_batch_num, img_h, img_w, width_multiple=1, strict=True):
        line_w[c] = w
        line_ratio[c] = max(img_w / img_h, max(float(wh_ratios[j]) for j in c))
    return line_w, line_ratio


def rec_batches_lines(wh_ratios: Sequence[float], rec_batch_num: int = 6, img_h: int = REC_IMG_H, img_w: int = REC_IMG_W,
                      launch_multiple: int = 32, n_min: int = 16, n_max: int = 160, n_step: int = 2,
                      n_cu: int = 256, with_ratio: bool = False, given: Optional[Tuple[np.ndarray, np.ndarray]] = None):
    """The reference's batching RESULT at GPU launch sizes.  Every line keeps the padded width the reference gives it - the imgW of
    its own chunk of `rec_batch_num` lines of the one global `np.argsort` (`rec_batches(strict=True)`, rapid_ocr.py:411-440) - and
    the launches are runs of that sorted list whose sizes `rd_rec_plan_chunks` picks for the chip, each launch tensor as wide as
    its widest line rounded up to `launch_multiple` (the recogniser computes a line at its own width inside the wider tensor:
    rd_rec_backbone_forward_lines).  Returns ([(indices into the input, launch width)], reference width per line in the order of
    the concatenated indices); `with_ratio=True` adds the chunk's `max_wh_ratio` per line (what CTCLabelDecode scales a line's time steps
    by when word boxes are asked for).
    `given` = (reference width, max_wh_ratio) per INPUT line, decided elsewhere: a rank of a page-sharded run holds only some of the
    lines the reference would have pooled, sorted and chunked together, and gets every line's width from the global list
    (dist.GlobalLineWidths); the launches are then runs of the local lines sorted by that width."""
    if given is not None:
        gw, gr = np.asarray(given[0], dtype=np.int64), np.asarray(given[1], dtype=np.float64)
        assert len(gw) == len(gr) == len(wh_ratios)
        if len(gw) == 0:
            return ([], np.zeros(0, np.int64), np.zeros(0)) if with_ratio else ([], np.zeros(0, np.int64))
        order = np.lexsort((np.asarray(wh_ratios, dtype=np.float64), gw))            # by width, then ratio (stable)
        line_w, ratio_sorted = gw[order], gr[order]
        ref = None
    else:
        ref = rec_batches(wh_ratios, rec_batch_num, img_h, img_w, width_multiple=1, strict=True)
        if not ref:
            return ([], np.zeros(0, np.int64), np.zeros(0)) if with_ratio else ([], np.zeros(0, np.int64))
        order = np.concatenate([c for c, _w in ref])
        line_w = np.concatenate([np.full(len(c), w, dtype=np.int64) for c, w in ref])      # non-decreasing: the chunks are sorted by ratio
    total = len(order)
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    w32 = np.ascontiguousarray((line_w + launch_multiple - 1) // launch_multiple * launch_multiple, dtype=np.int32)
    sizes = np.zeros(total, dtype=np.int32)
    n_out = C.c_int32(0)
    rc = lib.rd_rec_plan_chunks(w32.ctypes.data, total, n_min, n_max, n_step, n_cu, sizes.ctypes.data, total, C.byref(n_out))
    if rc != 0:
        raise RuntimeError("rd_rec_plan_chunks failed")
    out, i = [], 0
    for n in sizes[: n_out.value].tolist():
        out.append((order[i: i + n], int(w32[i + n - 1])))
        i += n
    assert i == total
    if with_ratio:
        if ref is None:
            return out, line_w, ratio_sorted
        line_ratio = np.concatenate([np.full(len(c), max(img_w / img_h, max(float(wh_ratios[j]) for j in c))) for c, _w in ref])
        return out, line_w, line_ratio
    return out, line_w


def rec_resized_width(w: float, h: float, wpad: int, img_h: int = REC_IMG_H) -> int:
    """resize_norm_img: resized_w = min(imgW, ceil(imgH * w/h))."""
    return int(min(wpad, math.ceil(img_h * (w / h))))


# ------------------------------------------------------------------------------------------------------------------
# DB post-process (C++ in librapiddoc_mi355.so, host side) and the box ordering that follows it
# ------------------------------------------------------------------------------------------------------------------
TEXT_BOX_DTYPE = np.dtype([("pts", "<f4", (8,)), ("score", "<f4")])


def db_postprocess(prob: np.ndarray, src_hw: Sequence[Tuple[int, int]], thresh: float = 0.3, box_thresh: float = 0.5,
                   unclip_ratio: float = 1.6, use_dilation: bool = True, max_candidates: int = 1000, max_out: int = 2048,
                   n_threads: int = 0) -> List[Tuple[np.ndarray, List[float]]]:
    """prob: [B,H,W] or [B,1,H,W] float32 DB probability maps (host).  Returns per image (boxes [n,4,2] int32 in
    source pixels ordered tl,tr,br,bl, scores) - the (boxes, scores) pair rapidocr's DBPostProcess returns."""
    import ctypes as C

    from . import _lib
    lib = _lib.load()
    p = np.ascontiguousarray(prob, dtype=np.float32)
    if p.ndim == 4:
        p = p[:, 0]
    B, H, W = p.shape
    p = np.ascontiguousarray(p)
    hw = np.ascontiguousarray(np.asarray(src_hw, dtype=np.int32).reshape(B, 2))
    out = np.zeros((B, max_out), dtype=TEXT_BOX_DTYPE)
    n = np.zeros(B, dtype=np.int32)
    rc = lib.rd_db_postprocess(p.ctypes.data, B, H, W, hw.ctypes.data, thresh, box_thresh, unclip_ratio, 1 if use_dilation else 0,
                               max_candidates, out.ctypes.data, max_out, n.ctypes.data, n_threads)
    if rc != 0:
        raise RuntimeError("rd_db_postprocess failed")
    res = []
    for b in range(B):
        k = int(n[b])
        res.append((out["pts"][b, :k].reshape(k, 4, 2).astype(np.int32), out["score"][b, :k].tolist()))
    return res


def db_postprocess_device(prob_dev, src_hw: Sequence[Tuple[int, int]], thresh: float = 0.3, box_thresh: float = 0.5,
                          unclip_ratio: float = 1.6, use_dilation: bool = True, max_candidates: int = 1000, max_out: int = 2048,
                          max_runs: int = 65536, stats: dict = None, cache: dict = None) -> List[Tuple[np.ndarray, List[float]]]:
    """`db_postprocess` with NOTHING on the host (`rd_db_boxes_device`): `prob_dev` is the CUDA tensor [B,1,H,W] /
    [B,H,W] the det forward wrote; bitmap runs, region labelling, min-area rectangles, scores, unclip and the final filter run
    as six kernels on the current stream, and ONE device-to-host copy brings the finished boxes (a few KB).  Same boxes in the
    same order as the host path (tests/test_gpu_image_ops.py).  Falls back to the host path if a page has more than `max_runs`
    bitmap runs (noise maps).

    `cache` is the CALLER's dict of device workspaces / result buffers keyed by shape: one per PagePipeline / RegionOcr, i.e.
    one per host thread and HIP stream (PagePipelinePool runs one pipeline per thread; a process-wide cache would hand two
    streams the same union-find arrays).  None = allocate for this call only."""
    import time

    import torch

    from . import _lib
    lib = _lib.load()
    p = prob_dev if prob_dev.dim() == 3 else prob_dev[:, 0]
    assert p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()
    B, H, W = p.shape
    dev = p.device.index or 0
    st = torch.cuda.current_stream().cuda_stream
    t0 = time.perf_counter()
    mo = int(min(max_out, max_candidates))
    key = (dev, B, H, W, max_runs, max_candidates, mo)
    bufs = cache.get(key) if cache is not None else None
    if bufs is None:
        nbytes = lib.rd_db_boxes_workspace(B, H, W, max_runs, max_candidates)
        if cache is not None and len(cache) > 8:
            cache.clear()
        # results: int32 [B + 1] counts (+ overflow flag), padded to 64 bytes, then [B][mo] boxes of 9 floats - one buffer,
        # one copy
        head = (4 * (B + 1) + 63) // 64 * 64
        bufs = {"ws": torch.empty(nbytes, dtype=torch.uint8, device=p.device), "head": head,
                "res": torch.empty(head + B * mo * 36, dtype=torch.uint8, device=p.device),
                "res_h": torch.empty(head + B * mo * 36, dtype=torch.uint8, pin_memory=True),
                "hw": torch.empty((B, 2), dtype=torch.int32, device=p.device), "hw_key": None}
        if cache is not None:
            cache[key] = bufs
    hw_np = np.ascontiguousarray(np.asarray(src_hw, dtype=np.int32).reshape(B, 2))
    if bufs["hw_key"] != hw_np.tobytes():
        bufs["hw"].copy_(torch.from_numpy(hw_np), non_blocking=False)
        bufs["hw_key"] = hw_np.tobytes()
    res, head = bufs["res"], bufs["head"]
    rc = lib.rd_db_boxes_device(dev, p.data_ptr(), B, H, W, bufs["hw"].data_ptr(), thresh, box_thresh, unclip_ratio, 1 if use_dilation else 0,
                                max_candidates, max_runs, bufs["ws"].data_ptr(), bufs["ws"].numel(), res.data_ptr() + head, mo,
                                res.data_ptr(), st)
    if rc != 0:
        raise RuntimeError("rd_db_boxes_device failed")
    bufs["res_h"].copy_(res, non_blocking=True)
    torch.cuda.current_stream().synchronize()          # the one wait: det forward + post-process + copy
    t1 = time.perf_counter()
    raw = bufs["res_h"].numpy()
    counts = raw[: 4 * (B + 1)].view("<i4")
    if int(counts[B]) != 0:
        return db_postprocess(p.cpu().numpy(), src_hw, thresh, box_thresh, unclip_ratio, use_dilation, max_candidates, max_out)
    boxes = raw[head:].view(TEXT_BOX_DTYPE).reshape(B, mo)
    out = []
    for b in range(B):
        k = int(counts[b])
        out.append((boxes["pts"][b, :k].reshape(k, 4, 2).astype(np.int32), boxes["score"][b, :k].tolist()))
    if stats is not None:
        stats["t_wait_maps_ms"] = (t1 - t0) * 1e3
        stats["t_db_post_ms"] = (time.perf_counter() - t1) * 1e3
    return out


def sorted_boxes(dt_boxes: Sequence[np.ndarray]) -> List[np.ndarray]:
    """Reading order: top-to-bottom, left-to-right, boxes whose top-left y differ by < 10 px count as one row
    (rapid_doc/utils/ocr_utils.py:105-127)."""
    boxes = sorted(list(dt_boxes), key=lambda b: (b[0][1], b[0][0]))
    for i in range(len(boxes) - 1):
        for j in range(i, -1, -1):
            if abs(boxes[j + 1][0][1] - boxes[j][0][1]) < 10 and boxes[j + 1][0][0] < boxes[j][0][0]:
                boxes[j], boxes[j + 1] = boxes[j + 1], boxes[j]
            else:
                break
    return boxes


# ------------------------------------------------------------------------------------------------------------------
# Text-box bookkeeping between det and rec (rapid_doc/utils/ocr_utils.py:16-67,130-317,478-485), restated.
# Checked against the reference functions themselves through tests/golden/boxes_seed*.json.
# ------------------------------------------------------------------------------------------------------------------
LINE_WIDTH_TO_HEIGHT_RATIO_THRESHOLD = 4  # ocr_utils.py:13


def quad_is_tilted(quad) -> bool:
    """calculate_is_angle: the diagonal's vertical extent differs from the mean side height by more than 20 %."""
    p1, p2, p3, p4 = quad
    height = ((p4[1] - p1[1]) + (p3[1] - p2[1])) / 2
    return not (0.8 * height <= (p3[1] - p1[1]) <= 1.2 * height)


def _quad_to_bbox(q):
    return [q[0][0], q[0][1], q[1][0], q[2][1]]


def _bbox_to_quad(b) -> np.ndarray:
    x0, y0, x1, y1 = b
    return np.array([[x0, y0], [x1, y0], [x1, y1], [x0, y1]]).astype("float32")


def _y_overlap_exceeds(b1, b2, thr: float) -> bool:
    ov = max(0, min(b1[3], b2[3]) - max(b1[1], b2[1]))
    mh = min(b1[3] - b1[1], b2[3] - b2[1])
    return (ov / mh) > thr if mh > 0 else False


def merge_det_boxes(dt_boxes) -> List[np.ndarray]:
    """Group axis-aligned boxes into lines (y-overlap > 0.6 with the previous box, after sorting by top y) and, for
    lines wider than 4x their height, fuse horizontally overlapping boxes; tilted boxes pass through at the end."""
    flat, tilted = [], []
    for q in dt_boxes:
        if quad_is_tilted(q):
            tilted.append(q)
        else:
            flat.append(_quad_to_bbox(q))
    flat.sort(key=lambda b: b[1])
    lines: List[List[list]] = []
    for b in flat:
        if lines and _y_overlap_exceeds(b, lines[-1][-1], 0.6):
            lines[-1].append(b)
        else:
            lines.append([b])
    out: List[np.ndarray] = []
    for line in lines:
        lw = max(b[2] for b in line) - min(b[0] for b in line)
        lh = max(b[3] for b in line) - min(b[1] for b in line)
        if lw > lh * LINE_WIDTH_TO_HEIGHT_RATIO_THRESHOLD:
            merged: List[tuple] = []
            for b in sorted(line, key=lambda t: t[0]):
                if merged and not merged[-1][2] < b[0]:
                    m = merged.pop()
                    b = (min(m[0], b[0]), min(m[1], b[1]), max(m[2], b[2]), max(m[3], b[3]))
                merged.append(tuple(b))
            out.extend(_bbox_to_quad(m) for m in merged)
        else:
            out.extend(_bbox_to_quad(b) for b in line)
    out.extend(tilted)
    return out


def _subtract_intervals(span, masks):
    masks = sorted([list(m) for m in masks], key=lambda m: m[0])
    merged: List[list] = []
    for m in masks:
        if merged and not merged[-1][1] < m[0]:
            merged[-1][1] = max(merged[-1][1], m[1])
        else:
            merged.append(m)
    start, end = span
    res = []
    for ms, me in merged:
        if ms > end or me < start:
            continue
        if start < ms:
            res.append([start, ms - 1])
        start = max(me + 1, start)
    if start <= end:
        res.append([start, end])
    return res


def update_det_boxes(dt_boxes, formula_boxes) -> List[np.ndarray]:
    """Cut text boxes around inline formulas: remove, from every non-tilted text box, the x-ranges of the formula
    boxes that overlap it vertically by > 0.8 of the smaller height.  formula_boxes: [{'bbox': [x0,y0,x1,y1]}]."""
    out, tilted = [], []
    for q in dt_boxes:
        if quad_is_tilted(q):
            tilted.append(q)
            continue
        tb = _quad_to_bbox(q)
        masks = [[f["bbox"][0], f["bbox"][2]] for f in formula_boxes if _y_overlap_exceeds(tb, f["bbox"], 0.8)]
        for x0, x1 in _subtract_intervals([tb[0], tb[2]], masks):
            out.append(_bbox_to_quad([x0, tb[1], x1, tb[3]]))
    out.extend(tilted)
    return out
