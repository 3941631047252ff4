"""Device-resident page-batch hot path: layout backbone -> OCR det -> line crops -> OCR rec -> CTC decode.

Mirrors the stage order of the reference's `BatchAnalyze.__call__`
(rapid_doc/backend/pipeline/batch_analyze.py:78-164) for the stages whose networks are readable in the
reference: page images stay in HBM as u8 HWC, every resize / normalise / crop is a HIP
kernel, the three forwards run through the C-ABI, and only (idx, prob) per CTC time step and the text-line
boxes cross PCIe.
"""
from __future__ import annotations

import os
import time
import ctypes as C
from concurrent.futures import ThreadPoolExecutor
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _lib, ocr_host
from .engine import RdEngine, pad_tail_lengths, rec_line_table, preproc_resize_norm, preproc_resize_norm_batch

LAYOUT_SIZE = 800          # PP-DocLayout-L/plus-L/V2/V3 input (pp_doclayout/main.py:17-29)
DET_LIMIT = 960            # rapidocr Det.limit_side_len (rapid_ocr.py:517-518)


class CropDesc(C.Structure):
    _fields_ = [("page", C.c_int32), ("out_w", C.c_int32), ("crop_w", C.c_float), ("crop_h", C.c_float),
                ("m", C.c_float * 9), ("rot90", C.c_int32), ("pad_", C.c_int32)]


CROP_DTYPE = np.dtype([("page", "<i4"), ("out_w", "<i4"), ("crop_w", "<f4"), ("crop_h", "<f4"), ("m", "<f4", (9,)),
                       ("rot90", "<i4"), ("pad_", "<i4")])
assert CROP_DTYPE.itemsize == C.sizeof(CropDesc)
# rd_line_crop_desc (include/rapiddoc_mi355.h): the reference-shaped two-stage crop (cubic warp to the integer-sized uint8
# crop, rotation of tall crops, linear resize to height 48), homography in float64 like cv2's
LINE_DTYPE = np.dtype([("page", "<i4"), ("out_w", "<i4"), ("crop_w", "<i4"), ("crop_h", "<i4"), ("rot90", "<i4"),
                       ("scratch_off", "<i4"), ("m", "<f8", (9,))])
assert LINE_DTYPE.itemsize == 96


def quad_to_crop_matrix(quad: np.ndarray) -> Tuple[np.ndarray, float, float]:
    """Inverse of the reference's perspective rectification (utils/ocr_utils.py:494-536): returns the 3x3 matrix
    that maps rectified-crop pixel (x, y, 1) to page coordinates plus the crop size.  `quad`: 4x2 points
    (tl, tr, br, bl)."""
    q = np.asarray(quad, dtype=np.float64).reshape(4, 2)
    cw = max(np.linalg.norm(q[0] - q[1]), np.linalg.norm(q[2] - q[3]))
    ch = max(np.linalg.norm(q[0] - q[3]), np.linalg.norm(q[1] - q[2]))
    cw, ch = float(int(cw)), float(int(ch))  # ocr_utils.py:508-519 casts to int
    cw, ch = max(cw, 1.0), max(ch, 1.0)
    src = np.array([[0, 0], [cw, 0], [cw, ch], [0, ch]], dtype=np.float64)
    # solve the homography src -> q (8 unknowns)
    A, b = [], []
    for (x, y), (u, v) in zip(src, q):
        A.append([x, y, 1, 0, 0, 0, -u * x, -u * y]); b.append(u)
        A.append([0, 0, 0, x, y, 1, -v * x, -v * y]); b.append(v)
    h = np.linalg.solve(np.asarray(A), np.asarray(b))
    return np.append(h, 1.0).astype(np.float32), cw, ch


def quads_to_crop_matrices(quads: np.ndarray) -> Tuple[np.ndarray, np.ndarray, np.ndarray, np.ndarray]:
    """Vectorised `quad_to_crop_matrix` for [n,4,2] quads -> ([n,9] float64, crop widths, crop heights, ok mask).
    Crop size as `get_rotate_crop_image` computes it (utils/ocr_utils.py:508-519): `int(max(np.linalg.norm(p0 - p1),
    np.linalg.norm(p2 - p3)))` on FLOAT32 points - float32 subtraction, float32 norm (sqrt of the float32 sum of the two
    float32 squares), truncation - so an edge whose length is within float32 rounding of an integer gets the crop size
    the reference gives it.  `ok` is False for quads no homography exists for (zero-size crop, collinear corners): the
    reference's cv2.warpPerspective raises on them; callers here skip them."""
    q32 = np.asarray(quads, dtype=np.float32).reshape(-1, 4, 2)
    q = q32.astype(np.float64)
    n = len(q)

    def norm32(a, b):
        d = a - b                                               # float32
        return np.sqrt(d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1])   # float32 throughout, like np.linalg.norm on a float32 vector
    cw = np.maximum(norm32(q32[:, 0], q32[:, 1]), norm32(q32[:, 2], q32[:, 3])).astype(np.float64)
    ch = np.maximum(norm32(q32[:, 0], q32[:, 3]), norm32(q32[:, 1], q32[:, 2])).astype(np.float64)
    cw, ch = np.floor(cw), np.floor(ch)
    ok = (cw >= 1.0) & (ch >= 1.0) & np.isfinite(cw) & np.isfinite(ch)
    cw, ch = np.where(ok, cw, 1.0), np.where(ok, ch, 1.0)
    z = np.zeros(n)
    src = np.stack([np.stack([z, z], 1), np.stack([cw, z], 1), np.stack([cw, ch], 1), np.stack([z, ch], 1)], axis=1)  # [n,4,2]
    A = np.zeros((n, 8, 8))
    b = np.zeros((n, 8))
    x, y, u, v = src[..., 0], src[..., 1], q[..., 0], q[..., 1]
    A[:, 0::2, 0], A[:, 0::2, 1], A[:, 0::2, 2] = x, y, 1.0
    A[:, 0::2, 6], A[:, 0::2, 7] = -u * x, -u * y
    A[:, 1::2, 3], A[:, 1::2, 4], A[:, 1::2, 5] = x, y, 1.0
    A[:, 1::2, 6], A[:, 1::2, 7] = -v * x, -v * y
    b[:, 0::2], b[:, 1::2] = u, v
    # a singular system (three collinear corners, repeated points) must not abort the whole page batch
    # scale-aware test: a homography of the crop rectangle exists iff no three corners of the quad are collinear - every corner
    # triangle must have an area that is not vanishing next to the crop's (an absolute bound on det(A), whose entries reach
    # coordinate^2, lets near-collinear quads through)
    with np.errstate(all="ignore"):
        for i, j, k in ((0, 1, 2), (1, 2, 3), (2, 3, 0), (3, 0, 1)):
            cross = (q[:, j, 0] - q[:, i, 0]) * (q[:, k, 1] - q[:, i, 1]) - (q[:, j, 1] - q[:, i, 1]) * (q[:, k, 0] - q[:, i, 0])
            ok &= np.isfinite(cross) & (np.abs(cross) > 1e-6 * cw * ch)
    A[~ok] = np.eye(8)
    h = np.linalg.solve(A, b[..., None])[..., 0]
    return np.concatenate([h, np.ones((n, 1))], axis=1), cw, ch, ok


def boxes_to_quads(boxes_xyxy: np.ndarray) -> np.ndarray:
    b = np.asarray(boxes_xyxy, dtype=np.float32).reshape(-1, 4)
    return np.stack([b[:, [0, 1]], b[:, [2, 1]], b[:, [2, 3]], b[:, [0, 3]]], axis=1)


@dataclass
class PageResult:
    lines: List[Tuple[np.ndarray, str, float]] = field(default_factory=list)  # (quad 4x2, text, score)
    layout_feats: Optional[List[torch.Tensor]] = None


class PagePipeline:
    def __init__(self, states: Dict[str, object], device: int = 0, characters: Optional[Sequence[str]] = None,
                 rec_batch_num: Optional[int] = None, rec_width_multiple: int = 32, keep_feats: bool = False, n_rec_streams: int = 4,
                 rec_mode: Optional[str] = None, rec_chunking: Optional[str] = None):
        """`states`: {'ppocrv6_det': ..., 'ppocrv6_rec': ..., 'pphgnetv2_b4': ...}, each a .safetensors path,
        bytes, or name->ndarray dict.

        `rec_mode`: how the text lines are batched for the recogniser.  A line's logits depend on its batch's padded width
        (LightSVTR attends over the zero-padded columns), so the batching is part of the result:
          "strict"      (the default) the reference's batching, result for result: every line of the call pooled, ONE
                        `np.argsort(ratios)` with numpy's default sort, chunks of `rec_batch_num` = 6 (rapidocr's default),
                        every line padded to int(48 * max ratio of ITS chunk) (rapid_ocr.py:404-449).  The launches are
                        GPU-sized all the same: runs of the sorted list (sizes from `rd_rec_plan_chunks`) share one tensor
                        and the backbone computes every line at its own padded width inside it
                        (`rd_rec_backbone_forward_lines`; `ocr_host.rec_batches_lines`).  `rec_batch_num` /
                        `rec_width_multiple` are ignored.  (With RD_REC_TWO_STAGE=0: one launch per distinct padded width.)
          "throughput"  (chosen when `rec_batch_num` or `rec_chunking` is given without a mode) GPU-sized chunks (`rec_batch_num` lines, default 64) of the same aspect-sorted list, padded width
                        rounded up to `rec_width_multiple`: same per-tensor parity with the oracle, different padded
                        widths than the reference would have used.  `rec_chunking="adaptive"` (the default when no
                        `rec_batch_num` is given - what bench.py measures) lets the chunk SIZE follow the width:
                        `ocr_host.rec_batches_adaptive` picks, chunk by chunk, the size with the most lines per estimated
                        microsecond, i.e. tile counts of the persistent kernels that fill whole rounds of the 256 CUs (64 lines
                        of width 1056 are 3.09 rounds of mixer tiles and cost four)."""
        if rec_mode is None:
            rec_mode = "strict" if rec_batch_num is None and rec_chunking is None else "throughput"
        if rec_mode not in ("strict", "throughput"):
            raise ValueError("rec_mode must be 'strict' or 'throughput'")
        if rec_chunking is None:
            rec_chunking = "adaptive" if rec_batch_num is None else "fixed"
        if rec_batch_num is None:
            rec_batch_num = 64
        if rec_chunking not in ("fixed", "adaptive"):
            raise ValueError("rec_chunking must be 'fixed' or 'adaptive'")
        self.rec_mode = rec_mode
        self.rec_chunking = rec_chunking
        if rec_mode == "strict":
            rec_batch_num, rec_width_multiple = 6, 1
        self.device = device
        self.tdev = torch.device("cuda", device)
        self.n_cu = int(torch.cuda.get_device_properties(self.tdev).multi_processor_count)   # the chunk planners' round size
        # several forwards are in flight at once here, so the engines defer the split-fp16 range guard: run_batch (det,
        # layout) and rec_forward_lines (rec) call check_range_and_fallback() before any result is used
        # (reuse_outputs: the engines hand out the same output tensors for the same shape - every result is consumed inside the
        #  step that produced it - so that buffer addresses repeat from step to step and the library replays a forward as ONE
        #  hipGraph launch; the buffers this class owns are kept per role for the same reason, `_buf`)
        self.det = RdEngine("ppocrv6_det", device, guard="deferred", reuse_outputs=True).load_weights(states["ppocrv6_det"])
        # rec batches are independent: they alternate between `n_rec_streams` HIP streams (one engine handle = one
        # workspace per stream) so that the launch gaps / tails of one batch are filled by kernels of the other
        self.rec_engines = [RdEngine("ppocrv6_rec", device, guard="deferred").load_weights(states["ppocrv6_rec"]) for _ in range(max(1, n_rec_streams))]
        self.rec = self.rec_engines[0]
        self.rec_streams = [torch.cuda.Stream(device=self.tdev) for _ in self.rec_engines]
        # neck + CTC head of the two-stage recogniser: own handle (own workspace) and stream, it runs under the next backbones
        self.rec_tail = RdEngine("ppocrv6_rec", device, guard="deferred").load_weights(states["ppocrv6_rec"])
        self.tail_stream = torch.cuda.Stream(device=self.tdev)
        self.layout_stream = torch.cuda.Stream(device=self.tdev)
        # det runs on a stream of its own, so that the NEXT batch's det + layout forwards can be enqueued under this batch's
        # recognition (`run_batch(..., prefetch=next_pages)`): `_front` / `_prefetched`
        self.front_stream = torch.cuda.Stream(device=self.tdev)
        self._prefetched: Optional[dict] = None
        self._after_rec_enqueue = None    # one-shot hook of _rec_forward_sources_once: called with the backbones' events once they are all enqueued
        # where in a batch's recognition the next batch's det forward may start: after this fraction of the rec launches (1.0 = when
        # the last backbone is done, i.e. under the neck + CTC head, the decode and the step boundary, where the GPU runs under-filled;
        # earlier, its small kernels only take CUs away from the persistent recognition kernels: measured slower)
        self.prefetch_gate = float(os.environ.get("RD_PREFETCH_GATE", "1.0"))
        self.layout = RdEngine("pphgnetv2_b4", device, guard="deferred", reuse_outputs=True).load_weights(states["pphgnetv2_b4"]) if "pphgnetv2_b4" in states else None
        self._bufs: Dict[tuple, torch.Tensor] = {}
        ncls = self.rec.num_classes
        self.characters = list(characters) if characters is not None else ["blank"] + [chr(0x4E00 + i) for i in range(ncls - 2)] + [" "]
        assert len(self.characters) == ncls, (len(self.characters), ncls)
        self.rec_batch_num = rec_batch_num
        self.rec_width_multiple = rec_width_multiple
        self.keep_feats = keep_feats
        # CTC greedy decode on the device (rd_ctc_collapse): only finished strings + confidences cross PCIe
        tab, self._ctc_max_len = ocr_host.char_table(self.characters)
        self._ctc_table = torch.from_numpy(tab).to(self.tdev)
        self.device_ctc = True
        self.device_db = True            # DB post-process with the maps staying in HBM (ocr_host.db_postprocess_device)
        self.db_ws: dict = {}            # its device workspaces / pinned result buffers: per pipeline = per host thread and stream
        # recogniser in two stages: the batches run only the backbone (each into its slice of one token buffer) on their
        # streams, then the LightSVTR neck + CTC head run ONCE over all lines (rd_rec_tail_forward); RD_REC_TWO_STAGE=0: whole
        # network batch by batch
        self.rec_two_stage = os.environ.get("RD_REC_TWO_STAGE", "1") != "0"
        # page-sharded runs (dist.GlobalLineWidths): callable (pooling keys int64 [n], aspect ratios float64 [n]) -> (reference width, max_wh_ratio)
        # per line, decided over the lines of EVERY rank; None = this process holds the whole page batch
        self.rec_width_sync = None
        self._sync_this_call = True       # False inside a rec_forward_sources(pooled=False) call: its widths are its own lines'
        self._width_sync_cache = None
        self._width_sync_epoch, self._in_run_batch = 0, False
        self.keep_rec_inputs = False      # tests: keep every rec batch's input tensor and raw (idx, prob) in last_rec_batches
        self.last_rec_batches: List[Tuple[np.ndarray, torch.Tensor, torch.Tensor, torch.Tensor]] = []
        self._lib = _lib.load()
        self.stats: Dict[str, float] = {}

    # ---------------------------------------------------------------- stages
    def _buf(self, role, numel: int, dtype=torch.float32) -> torch.Tensor:
        """A persistent 1-D device buffer per role, grown (x 1.25) when a call needs more: the same role gets the same address
        step after step (hipGraph replays are keyed on the addresses)."""
        key = (role, dtype)
        t = self._bufs.get(key)
        if t is None or t.numel() < numel:
            t = self._bufs[key] = torch.empty(int(numel * 1.25) + 64, dtype=dtype, device=self.tdev)
        return t[:numel]

    def layout_preprocess(self, pages: torch.Tensor) -> torch.Tensor:
        """PPPreProcess for the whole batch in one launch: INTER_CUBIC resize, /255, mean 0 / std 1 (pre_process.py:14-42)."""
        P = pages.shape[0]
        out = self._buf("layout_x", P * 3 * LAYOUT_SIZE * LAYOUT_SIZE).view(P, 3, LAYOUT_SIZE, LAYOUT_SIZE)
        return preproc_resize_norm_batch(pages, (LAYOUT_SIZE, LAYOUT_SIZE), interp=2, out=out)

    def layout_forward(self, pages: torch.Tensor) -> List[torch.Tensor]:
        return self.layout.backbone_forward(self.layout_preprocess(pages))

    def det_preprocess(self, pages: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, int]]:
        """64-px bucket of the page region with its 50-px white margin, then DetPreProcess (analyze_utils.py:129-188): BGR,
        (x/255 - mean)/std with RapidDoc's Det.mean / Det.std (ocr_host.DET_MEAN), one launch for the batch."""
        P, H, W, _ = pages.shape
        bh, bw = -(-(H + 100) // 64) * 64, -(-(W + 100) // 64) * 64
        dh, dw = ocr_host.det_resize_shape(bh, bw, DET_LIMIT, "max")
        out = self._buf("det_x", P * 3 * dh * dw).view(P, 3, dh, dw)
        x = preproc_resize_norm_batch(pages, (dh, dw), mean=ocr_host.DET_MEAN, std=ocr_host.DET_STD, interp=1, swap_rb=True, out=out)
        return x, (dh, dw)

    def det_forward(self, pages: torch.Tensor) -> Tuple[torch.Tensor, Tuple[int, int]]:
        x, det_hw = self.det_preprocess(pages)
        return self.det.det_forward(x), det_hw

    def rec_forward_lines(self, pages: torch.Tensor, quads_per_page: Sequence[np.ndarray]):
        """All text lines of ONE image array -> [(text, score)] per image: `rec_forward_sources` with a single source."""
        return self.rec_forward_sources([(pages, quads_per_page)])[0]

    def rec_forward_sources(self, sources: Sequence[Tuple[torch.Tensor, Sequence[np.ndarray]]],
                            image_keys: Optional[Sequence[Sequence[int]]] = None, want_words: bool = False,
                            pooled: bool = True):
        """`sources`: [(images [P,H,W,3] u8 on the GPU, text-line quads per image)] - image arrays of DIFFERENT sizes whose
        lines are recognised TOGETHER, the way the reference pools every line of a page batch per language before it sorts
        and chunks them (analyze_utils.py:216-252 -> rapid_ocr.py:404-472).  Returns, per source, per image, [(text, score)]
        in the order of the quads.  `image_keys[source][image]`: the pooled line list is ordered by this key first (stable;
        e.g. the page a region crop came from - the reference pools page by page), which only matters to the strict mode's
        sort when two lines have exactly the same aspect ratio.  Carries the split-fp16 range guard of its rec engines (every caller - run_batch,
        analyze.RegionOcr, RegionTextModel - gets it): a tripped engine is switched to native fp32 and the lines are
        recognised again.  `want_words` (strict two-stage mode only): every line comes back as (text, score, words) with
        words = {cols, confs, n_steps, wh_ratio, max_wh_ratio, crop_hw} - the kept characters' time steps and probabilities and the
        numbers rapidocr's CTCLabelDecode / cal_rec_boxes turn into word boxes (rapiddoc_amd/word_boxes.py; table OCR, analyze_utils.py:308).
        `pooled=False`: a call whose lines the reference does NOT pool across pages (one table's OCR, analyze_utils.py:478-540: one
        recogniser call per table) - its widths depend on its own lines only, so a page-sharded run needs no exchange for it and
        `rec_width_sync` is not consulted (ranks hold different numbers of tables: a collective here would not pair up)."""
        self._sync_this_call = bool(pooled)
        if not self._in_run_batch:          # a call of its own: a new round of the width collective (run_batch opens one per batch)
            self._width_sync_epoch += 1
        out = self._rec_forward_sources_once(sources, image_keys, want_words)
        # a pass can trip a LATER stage only once the earlier one runs in fp32 (an overflowed backbone feeds the tail NaNs or
        # finite-but-huge tokens), so check after every pass; a tripped engine stays in fp32, which bounds the loop
        for _ in range(3):
            if not any([e.check_range_and_fallback() for e in self.rec_engines + [self.rec_tail]]):     # list: check every engine
                break
            self.stats["range_fallbacks"] = self.stats.get("range_fallbacks", 0) + 1
            out = self._rec_forward_sources_once(sources, image_keys, want_words)
        return out

    def _synced_widths(self, sync, keys: np.ndarray, ratios: np.ndarray):
        """One collective per recogniser CALL (per page batch inside run_batch): a range-guard repeat (same lines) re-uses the first
        pass's answer - a rank that repeats must not make a collective call its peers do not make."""
        sig = (self._width_sync_epoch, keys.tobytes(), ratios.tobytes())
        if self._width_sync_cache is not None and self._width_sync_cache[0] == sig:
            return self._width_sync_cache[1]
        res = sync(keys, ratios)
        self._width_sync_cache = (sig, res)
        return res

    def _collapse_rows(self, idx: torch.Tensor, prob: torch.Tensor, nb: int, st, record: bool = True):
        """Device CTC collapse of one batch's (idx, prob) [nb, T] on stream `st` + the D2H of its rows into pinned memory;
        returns (rows or None, event recorded after it or None)."""
        rows = None
        with torch.cuda.stream(st):
            if self.device_ctc:
                T = idx.shape[1]
                row_bytes = (16 + T * self._ctc_max_len + 15) // 16 * 16
                rows = torch.empty((nb, row_bytes), dtype=torch.uint8, device=idx.device)
                rc = self._lib.rd_ctc_collapse(self.device, idx.data_ptr(), prob.data_ptr(), nb, T, self._ctc_table.data_ptr(),
                                               self._ctc_max_len, len(self.characters), rows.data_ptr(), row_bytes, st.cuda_stream)
                if rc != 0:
                    raise RuntimeError("rd_ctc_collapse failed")
                rows_h = torch.empty((nb, row_bytes), dtype=torch.uint8, pin_memory=True)
                rows_h.copy_(rows, non_blocking=True)
                rows = rows_h
            done = None
            if record:
                done = torch.cuda.Event()
                done.record(st)
        return rows, done

    def _collapse_lines(self, idx_all: torch.Tensor, prob_all: torch.Tensor, tables: torch.Tensor, n_lines: int, max_tokens: int, st,
                        want_cols: bool = False):
        """Device CTC collapse of the RAGGED lines of one tail call (`tables` = its seg / tokinfo tensor; the first `n_lines` are the
        real ones) on stream `st` + the D2H of the rows into pinned memory; returns (rows, event) or (rows, event, kept columns)."""
        with torch.cuda.stream(st):
            row_bytes = (16 + max_tokens * self._ctc_max_len + 15) // 16 * 16
            rows = torch.empty((n_lines, row_bytes), dtype=torch.uint8, device=idx_all.device)
            cols = torch.empty((n_lines, max_tokens), dtype=torch.int16, device=idx_all.device) if want_cols else None
            confs = torch.empty((n_lines, max_tokens), dtype=torch.float32, device=idx_all.device) if want_cols else None
            rc = self._lib.rd_ctc_collapse_lines(self.device, idx_all.data_ptr(), prob_all.data_ptr(), n_lines, tables.data_ptr(), max_tokens,
                                                 self._ctc_table.data_ptr(), self._ctc_max_len, len(self.characters), rows.data_ptr(),
                                                 row_bytes, cols.data_ptr() if want_cols else None, confs.data_ptr() if want_cols else None,
                                                 st.cuda_stream)
            if rc != 0:
                raise RuntimeError("rd_ctc_collapse_lines failed")
            rows_h = torch.empty((n_lines, row_bytes), dtype=torch.uint8, pin_memory=True)
            rows_h.copy_(rows, non_blocking=True)
            cols_h = None
            if want_cols:
                cols_h = (torch.empty((n_lines, max_tokens), dtype=torch.int16, pin_memory=True),
                          torch.empty((n_lines, max_tokens), dtype=torch.float32, pin_memory=True))
                cols_h[0].copy_(cols, non_blocking=True)
                cols_h[1].copy_(confs, non_blocking=True)
            done = torch.cuda.Event()
            done.record(st)
        return (rows_h, done, cols_h) if want_cols else (rows_h, done)

    def _rec_forward_sources_once(self, sources, image_keys=None, want_words=False):
        t0 = time.perf_counter()
        if not sources:                 # nothing to read on this rank: it still takes part in the batch's width exchange
            if self.rec_width_sync is not None and self.rec_mode == "strict" and self.rec_two_stage and self._sync_this_call:
                self._synced_widths(self.rec_width_sync, np.zeros(0, np.int64), np.zeros(0))
            return []
        dev = sources[0][0].device
        # flat line list in source-major, image-major order (the reference's pooled order: pages, then a page's spans)
        src_of, page_of, quad_list, n_img = [], [], [], []
        for si, (imgs, quads_per_img) in enumerate(sources):
            assert imgs.is_cuda and imgs.dtype == torch.uint8 and imgs.dim() == 4 and imgs.is_contiguous()
            n_img.append(imgs.shape[0])
            for pi, q in enumerate(quads_per_img):
                q = np.asarray(q, dtype=np.float64).reshape(-1, 4, 2)
                if len(q):
                    quad_list.append(q)
                    src_of.append(np.full(len(q), si, np.int32))
                    page_of.append(np.full(len(q), pi, np.int32))
        empty = [[[] for _ in range(k)] for k in n_img]
        sync = self.rec_width_sync if (self.rec_mode == "strict" and self.rec_two_stage and self._sync_this_call) else None
        if not quad_list:
            if sync is not None:
                self._synced_widths(sync, np.zeros(0, np.int64), np.zeros(0))       # every rank makes the same calls
            return empty
        quads = np.concatenate(quad_list, axis=0)
        src_of, page_of = np.concatenate(src_of), np.concatenate(page_of)
        if image_keys is not None:      # pooled order: by key (page), then source / image / line as collected
            key = np.array([image_keys[si][pi] for si, pi in zip(src_of.tolist(), page_of.tolist())], dtype=np.int64)
            perm = np.argsort(key, kind="stable")
            quads, src_of, page_of, key = quads[perm], src_of[perm], page_of[perm], key[perm]
        else:
            perm = None
        mats, cws_a, chs_a, ok = quads_to_crop_matrices(quads)
        n_all = len(quads)
        keep = np.nonzero(ok)[0]                      # degenerate quads (zero-area / collinear corners) get ("", 0.0)
        n = len(keep)
        texts: List[tuple] = [("", 0.0, None) if want_words else ("", 0.0)] * n_all
        if n == 0:
            if sync is not None:
                self._synced_widths(sync, np.zeros(0, np.int64), np.zeros(0))
            return self._scatter_texts(texts, src_of, page_of, n_img, perm)
        mats, cws_a, chs_a = mats[keep], cws_a[keep], chs_a[keep]
        rots_a = (chs_a / cws_a >= 2.0).astype(np.int32)  # ocr_utils.py:531-534 rotates crops with h/w >= 2
        eff_w = np.where(rots_a == 1, chs_a, cws_a)
        eff_h = np.where(rots_a == 1, cws_a, chs_a)
        ratios = (eff_w / eff_h).tolist()
        strict = self.rec_mode == "strict"
        two_stage = self.rec_two_stage
        lines_mode = strict and two_stage          # reference widths per LINE inside GPU-sized launches
        if want_words and not lines_mode:
            raise RuntimeError("word boxes need the strict two-stage recogniser (rec_mode='strict', RD_REC_TWO_STAGE unset)")
        if lines_mode:
            given = None
            if sync is not None:        # the widths come from the pooled lines of every rank; key = the image key (global page index)
                line_keys = key[keep] if image_keys is not None else page_of[keep].astype(np.int64)
                given = self._synced_widths(sync, line_keys, np.asarray(ratios, dtype=np.float64))
            batches, line_w, line_ratio = ocr_host.rec_batches_lines(ratios, n_cu=self.n_cu, with_ratio=True, given=given)
        elif not strict and self.rec_chunking == "adaptive":
            batches = ocr_host.rec_batches_adaptive(ratios, width_multiple=self.rec_width_multiple, n_cu=self.n_cu)
        else:
            batches = ocr_host.rec_batches(ratios, self.rec_batch_num, width_multiple=self.rec_width_multiple, strict=strict,
                                           merge_equal_width=strict)
        order_all = np.concatenate([c for c, _ in batches])
        wpad_all = np.concatenate([np.full(len(c), w) for c, w in batches])
        if not lines_mode:
            line_w = wpad_all                           # every line of a launch is as wide as the launch
        descs = np.zeros(n, dtype=LINE_DTYPE)
        descs["page"] = page_of[keep][order_all]
        descs["out_w"] = np.minimum(line_w, np.ceil(ocr_host.REC_IMG_H * (eff_w / eff_h)[order_all])).astype(np.int32)
        descs["crop_w"] = cws_a[order_all].astype(np.int32)
        descs["crop_h"] = chs_a[order_all].astype(np.int32)
        descs["m"] = mats[order_all]
        descs["rot90"] = rots_a[order_all]
        # packed uint8 scratch of the rectified crops, one region per rec batch (batches run on different streams)
        crop_bytes = (descs["crop_w"].astype(np.int64) * descs["crop_h"] * 3 + 15) // 16 * 16
        batch_lens = [len(c) for c, _ in batches]
        starts = np.cumsum([0] + batch_lens)
        offs = np.zeros(n, np.int64)
        batch_scratch = []
        for b in range(len(batches)):
            cb = crop_bytes[starts[b]:starts[b + 1]]
            offs[starts[b]:starts[b + 1]] = np.cumsum(cb) - cb
            batch_scratch.append(int(cb.sum()))
        if max(batch_scratch) >= 2 ** 31:
            raise RuntimeError("a rec batch needs >= 2 GB of crop scratch: lower rec_batch_num")
        descs["scratch_off"] = offs.astype(np.int32)
        batch_base = np.cumsum([0] + batch_scratch)
        if getattr(self, "_crop_scratch", None) is None or self._crop_scratch.numel() < int(batch_base[-1]):
            self._crop_scratch = torch.empty(int(batch_base[-1] * 1.25) + 1024, dtype=torch.uint8, device=dev)
        # stage-1 (warp) launches: per rec batch, one per source image array present in it.  A single source (the page
        # pipeline) needs no second descriptor array: the batch's own descriptors are the launch's.
        multi = len(sources) > 1
        src_sorted = src_of[keep][order_all]
        crop_px = descs["crop_w"].astype(np.int64) * descs["crop_h"]
        warp_jobs: List[List[Tuple[int, int, int, int]]] = []        # per batch: (source, first desc, count, max crop pixels)
        if multi:
            warp_order = np.concatenate([starts[b] + np.argsort(src_sorted[starts[b]:starts[b + 1]], kind="stable")
                                         for b in range(len(batches))])
            descs_warp = descs[warp_order]
            ws_sorted = src_sorted[warp_order]
            px_sorted = crop_px[warp_order]
            for b in range(len(batches)):
                jobs, lo = [], int(starts[b])
                hi = int(starts[b + 1])
                cuts = [lo] + [i for i in range(lo + 1, hi) if ws_sorted[i] != ws_sorted[i - 1]] + [hi]
                for a, e in zip(cuts[:-1], cuts[1:]):
                    jobs.append((int(ws_sorted[a]), a, e - a, int(px_sorted[a:e].max())))
                warp_jobs.append(jobs)
        else:
            for b in range(len(batches)):
                lo, hi = int(starts[b]), int(starts[b + 1])
                warp_jobs.append([(0, lo, hi - lo, int(crop_px[lo:hi].max()))])
        self.stats["t_descs_ms"] = (time.perf_counter() - t0) * 1e3
        descs_dev = torch.from_numpy(descs.view(np.uint8)).to(dev, non_blocking=True)
        descs_warp_dev = torch.from_numpy(descs_warp.view(np.uint8)).to(dev, non_blocking=True) if multi else descs_dev
        outs = []
        pos = 0
        main = torch.cuda.current_stream()
        ready = torch.cuda.Event()
        ready.record(main)
        S = len(self.rec_engines)
        if two_stage:
            # the batches run the backbone only, each into its slice of one token buffer; the neck + CTC head then run once per
            # GROUP of S consecutive batches (one per stream) on the tail stream, under the backbones of the next group
            w2_ = (np.asarray(line_w, dtype=np.int64) - 1) // 2 + 1
            seq_line = ((w2_ - 1) // 2 + 1) // 2                  # tokens per line (rd_rec_seq_len of its padded width)
            groups = [list(range(g, min(g + S, len(batches)))) for g in range(0, len(batches), S)]
            group_lens = [seq_line[int(starts[grp[0]]): int(starts[grp[-1] + 1])] for grp in groups]
            # every group's tail call is padded with dummy lines onto a coarse (lines, longest line, tokens) grid: the plan cache
            # of the tail handle then sees a handful of keys instead of a new one per group (engine.pad_tail_lengths)
            group_pad = [pad_tail_lengths(l) for l in group_lens]
            group_base = np.cumsum([0] + [int(lp.sum()) for lp, _t in group_pad])
            first_tok = np.zeros(n, np.int64)                      # first token of every line in the token buffer
            for gi, grp in enumerate(groups):
                lo, hi = int(starts[grp[0]]), int(starts[grp[-1] + 1])
                first_tok[lo:hi] = int(group_base[gi]) + np.cumsum(seq_line[lo:hi]) - seq_line[lo:hi]
            tok_lo = [int(first_tok[int(starts[bi])]) for bi in range(len(batches))]
            dim = self.rec.rec_token_dim
            tokens = self._buf("rec_tokens", int(group_base[-1]) * dim).view(int(group_base[-1]), dim)
            for gi in range(len(groups)):         # the dummy lines' tokens: finite values (their outputs are ignored)
                tokens[int(group_base[gi]) + int(group_lens[gi].sum()): int(group_base[gi + 1])].zero_()
            group_tables = [self.rec_tail.rec_tail_tables(lp, dev, out=self._buf(("tail_tab", gi), 2 * len(lp) + int(lp.sum()), torch.int32))
                            for gi, (lp, _t) in enumerate(group_pad)]   # uploaded now, used later
            if lines_mode:
                tab_h = torch.from_numpy(rec_line_table(line_w, first_tok)).pin_memory()
                linetab = self._buf("rec_linetab", 4 * n, torch.int32).view(n, 4)
                linetab.copy_(tab_h, non_blocking=True)
                self._keep_linetab = tab_h                        # the pinned source must outlive the asynchronous copy
            ready.record(main)
            self.tail_stream.wait_event(ready)
        group_rows = {}
        bb_events: List[torch.cuda.Event] = []        # per rec launch: its backbone (or whole network) is done

        def run_tail(gi):
            grp = groups[gi]
            t_lo, t_hi = int(group_base[gi]), int(group_base[gi + 1])
            for bi in grp:
                self.tail_stream.wait_event(outs[bi][2])
            with torch.cuda.stream(self.tail_stream):
                n_tok = t_hi - t_lo
                idx_all, prob_all = self.rec_tail.rec_tail_forward(
                    tokens[t_lo:t_hi], group_pad[gi][0], group_tables[gi], max_tokens=group_pad[gi][1],
                    out=(self._buf(("tail_idx", gi), n_tok, torch.int32), self._buf(("tail_prob", gi), n_tok)))
            if lines_mode:
                # ragged lines: ONE collapse launch over the group's real lines (the seg table of the tail call), one copy back
                n_real = len(group_lens[gi])
                cols_h = None
                if want_words:
                    rows, done, cols_h = self._collapse_lines(idx_all, prob_all, group_tables[gi], n_real, group_pad[gi][1], self.tail_stream, True)
                else:
                    rows, done = self._collapse_lines(idx_all, prob_all, group_tables[gi], n_real, group_pad[gi][1], self.tail_stream)
                kept = None
                if self.keep_rec_inputs:
                    with torch.cuda.stream(self.tail_stream):        # (after the tail call, on its stream)
                        kept = (idx_all.clone(), prob_all.clone())
                group_rows[gi] = (rows, done, kept, cols_h)
                for bi in grp:
                    outs[bi] = (None, None, done, outs[bi][3], None)
                return
            done = None
            for bi in grp:
                nb, t = len(batches[bi][0]), int(seq_line[int(starts[bi])])
                lo = tok_lo[bi] - t_lo
                hi = lo + nb * t
                idx, prob = idx_all[lo:hi].view(nb, t), prob_all[lo:hi].view(nb, t)
                rows, ev = self._collapse_rows(idx, prob, nb, self.tail_stream, record=bi == grp[-1])
                done = ev or done
                outs[bi] = (idx, prob, None, outs[bi][3], rows)
            for bi in grp:
                outs[bi] = outs[bi][:2] + (done,) + outs[bi][3:]

        for bi, (chunk, wpad) in enumerate(batches):
            nb = len(chunk)
            k = bi % S
            st = self.rec_streams[k]
            if bi < S:
                st.wait_event(ready)          # descriptors (and the pages, the token buffer) are ready
            with torch.cuda.stream(st):
                # one input buffer per stream (batches bi, bi + S, ... of a stream run one after the other)
                x = self._buf(("rec_x", k), nb * 3 * ocr_host.REC_IMG_H * wpad).view(nb, 3, ocr_host.REC_IMG_H, wpad)
                scratch = self._crop_scratch.data_ptr() + int(batch_base[bi])
                for si, first, cnt, max_px in warp_jobs[bi]:            # stage 1: page / canvas -> rectified uint8 crops
                    imgs = sources[si][0]
                    rc = self._lib.rd_line_warp_batch(self.device, imgs.data_ptr(), imgs.shape[0], imgs.shape[1], imgs.shape[2],
                                                      descs_warp_dev.data_ptr() + first * LINE_DTYPE.itemsize, cnt, max_px, scratch,
                                                      st.cuda_stream)
                    if rc != 0:
                        raise RuntimeError("rd_line_warp_batch failed")
                rc = self._lib.rd_line_resize_norm_batch(self.device, descs_dev.data_ptr() + pos * LINE_DTYPE.itemsize, nb, scratch,
                                                         ocr_host.REC_IMG_H, wpad, 1, x.data_ptr(), st.cuda_stream)   # stage 2
                if rc != 0:
                    raise RuntimeError("rd_line_resize_norm_batch failed")
                if lines_mode:
                    self.rec_engines[k].rec_backbone_forward_lines(x, linetab[pos: pos + nb], tokens)
                    ev = torch.cuda.Event()
                    ev.record(st)
                    outs.append((None, None, ev, x.clone() if self.keep_rec_inputs else x, None))
                elif two_stage:
                    self.rec_engines[k].rec_backbone_forward(x, tokens[tok_lo[bi]: tok_lo[bi] + nb * int(seq_line[pos])])
                    ev = torch.cuda.Event()
                    ev.record(st)
                    outs.append((None, None, ev, x.clone() if self.keep_rec_inputs else x, None))
                else:
                    idx, prob, _ = self.rec_engines[k].rec_forward(x)
                    rows, done = self._collapse_rows(idx, prob, nb, st)
                    outs.append((idx, prob, done, x.clone() if self.keep_rec_inputs else x, rows))
            bb_events.append(outs[-1][2])
            pos += nb
            if two_stage and (bi + 1) % S == 0:
                run_tail(bi // S)
        if two_stage and len(batches) % S:
            run_tail(len(groups) - 1)
        self.stats["t_rec_enqueue_ms"] = (time.perf_counter() - t0) * 1e3 - self.stats["t_descs_ms"]
        hook, self._after_rec_enqueue = self._after_rec_enqueue, None
        if hook is not None:
            t_h = time.perf_counter()
            m = max(1, min(len(batches), int(np.ceil(self.prefetch_gate * len(batches)))))
            hook(bb_events[max(0, m - S): m])          # streams run their launches in order: the last S up to m cover all before them
            self.stats["t_front_next_ms"] = (time.perf_counter() - t_h) * 1e3
        # D2H per batch result (small) as soon as that batch is done, host CTC decode (rapidocr CTCLabelDecode)
        # overlaps the GPU work of the batches still in flight
        t_dec = 0.0
        if self.keep_rec_inputs:
            self.last_rec_crop_sizes = (cws_a.astype(np.int64), chs_a.astype(np.int64), rots_a, keep)
        if lines_mode:
            self.last_rec_batches = []
            for gi, grp in enumerate(groups):
                rows, done, kept, cols_h = group_rows[gi]
                done.synchronize()
                t1 = time.perf_counter()
                lo = int(starts[grp[0]])
                rows_np = rows.numpy()
                dec = ocr_host.parse_ctc_rows(rows_np)
                if want_words:
                    n_kept = rows_np[:, 8:12].copy().view("<i4")[:, 0]
                    cols_np, confs_np = cols_h[0].numpy(), cols_h[1].numpy()
                for j, (t, sc) in enumerate(dec):
                    if want_words:       # (raw score: the reference's ocr(det=False) returns rapidocr's float as it is, rapid_ocr.py:295-297)
                        i = int(order_all[lo + j])
                        k = int(n_kept[j])
                        words = {"cols": cols_np[j, :k].astype(np.int64).tolist(), "confs": confs_np[j, :k].astype(np.float64).tolist(),
                                 "n_steps": int(seq_line[lo + j]), "wh_ratio": float(ratios[i]), "max_wh_ratio": float(line_ratio[lo + j]),
                                 "crop_hw": (int(eff_h[i]), int(eff_w[i]))}
                        texts[int(keep[i])] = (t, sc, words)
                        continue
                    texts[int(keep[order_all[lo + j]])] = (t, ocr_host.format_score(sc))
                t_dec += time.perf_counter() - t1
                if kept is not None:        # tests: per rec batch (pooled line ids, input tensor, reference width per line, per-line idx, prob)
                    g0 = int(group_base[gi])
                    for bi in grp:
                        a, e = int(starts[bi]), int(starts[bi + 1])
                        sl = [(int(first_tok[i]) - g0, int(seq_line[i])) for i in range(a, e)]
                        self.last_rec_batches.append((keep[np.asarray(batches[bi][0])], outs[bi][3], np.asarray(line_w[a:e]),
                                                      [kept[0][o: o + t] for o, t in sl], [kept[1][o: o + t] for o, t in sl]))
        else:
            if self.keep_rec_inputs:
                self.last_rec_batches = [(keep[np.asarray(chunk)], x, idx, prob) for (chunk, _w), (idx, prob, _d, x, _r) in zip(batches, outs)]
            for (chunk, wpad), (idx, prob, done, _x, rows) in zip(batches, outs):
                done.synchronize()
                if rows is not None:
                    t1 = time.perf_counter()
                    dec = ocr_host.parse_ctc_rows(rows.numpy())
                else:
                    idx_h, prob_h = idx.cpu().numpy(), prob.cpu().numpy()
                    t1 = time.perf_counter()
                    dec = ocr_host.ctc_decode(idx_h, prob_h, self.characters)
                for j, i in enumerate(chunk):
                    t, s = dec[j]
                    texts[int(keep[i])] = (t, ocr_host.format_score(s))
                t_dec += time.perf_counter() - t1
        self.stats["t_decode_ms"] = t_dec * 1e3
        for st in self.rec_streams + [self.tail_stream]:
            main.wait_stream(st)
        self.stats["rec_lines"] = n
        self.stats["rec_batches"] = len(batches)
        self.last_pool_order = perm
        return self._scatter_texts(texts, src_of, page_of, n_img, perm)

    @staticmethod
    def _scatter_texts(texts, src_of, page_of, n_img, perm=None):
        """pooled-order results -> [source][image][line], lines in the order their quads were given"""
        if perm is not None:             # undo the key sort: collection order is (source, image, line)
            inv = np.argsort(perm, kind="stable")
            texts = [texts[i] for i in inv.tolist()]
            src_of, page_of = src_of[inv], page_of[inv]
        out = [[[] for _ in range(k)] for k in n_img]
        for t, si, pi in zip(texts, src_of.tolist(), page_of.tolist()):
            out[si][pi].append(t)
        return out

    # ---------------------------------------------------------------- det maps -> text-line quads (host)
    def boxes_from_maps(self, maps_host: np.ndarray, page_hw: Tuple[int, int], box_thresh: float = 0.3,
                        unclip_ratio: float = 1.8) -> List[np.ndarray]:
        """DB post-process + reading-order sort + same-line merge (rapid_ocr.py:537-538; analyze_utils.py:196-204).
        box_thresh 0.3 / unclip 1.8 are the page-OCR settings (backend/pipeline/model_init.py:73)."""
        P = maps_host.shape[0]
        res = ocr_host.db_postprocess(maps_host, [page_hw] * P, thresh=0.3, box_thresh=box_thresh, unclip_ratio=unclip_ratio)
        out = []
        for boxes, _scores in res:
            if len(boxes) == 0:
                out.append(np.zeros((0, 4, 2), np.float32))
                continue
            b = ocr_host.sorted_boxes(boxes.astype(np.float32))
            b = ocr_host.merge_det_boxes(b)
            out.append(np.asarray(b, dtype=np.float32).reshape(-1, 4, 2))
        return out

    def boxes_from_maps_device(self, maps_dev: torch.Tensor, page_hw: Tuple[int, int], box_thresh: float = 0.3,
                               unclip_ratio: float = 1.8) -> List[np.ndarray]:
        """`boxes_from_maps` with the maps staying in HBM (ocr_host.db_postprocess_device)."""
        P = maps_dev.shape[0]
        res = ocr_host.db_postprocess_device(maps_dev, [page_hw] * P, thresh=0.3, box_thresh=box_thresh, unclip_ratio=unclip_ratio,
                                             stats=self.stats, cache=self.db_ws)
        out = []
        for boxes, _scores in res:
            if len(boxes) == 0:
                out.append(np.zeros((0, 4, 2), np.float32))
                continue
            b = ocr_host.merge_det_boxes(ocr_host.sorted_boxes(boxes.astype(np.float32)))
            out.append(np.asarray(b, dtype=np.float32).reshape(-1, 4, 2))
        return out

    def _boxes_via_host_maps(self, src: torch.Tensor, page_hw: Tuple[int, int]) -> List[np.ndarray]:
        """Round-1 path: the float maps cross PCIe (pinned buffer) and the whole post-process runs on the host."""
        if getattr(self, "_maps_host", None) is None or self._maps_host.shape != src.shape:
            self._maps_host = torch.empty(src.shape, dtype=torch.float32, pin_memory=True)
        t0 = time.perf_counter()
        self._maps_host.copy_(src, non_blocking=True)
        torch.cuda.current_stream().synchronize()
        t1 = time.perf_counter()
        quads = self.boxes_from_maps(self._maps_host.numpy(), page_hw)
        self.stats["t_wait_maps_ms"] = (t1 - t0) * 1e3
        self.stats["t_db_post_ms"] = (time.perf_counter() - t1) * 1e3
        return quads

    # ---------------------------------------------------------------- whole batch
    def run_batch(self, pages: torch.Tensor, quads_per_page: Optional[Sequence[np.ndarray]] = None,
                  det_maps_override: Optional[torch.Tensor] = None, prefetch: Optional[torch.Tensor] = None,
                  page_keys: Optional[Sequence[int]] = None) -> List[PageResult]:
        """`page_keys`: the pages' positions in the GLOBAL page list of a page-sharded run (the pooling key of `rec_width_sync`,
        dist.GlobalLineWidths); default: the local page index.
        `_run_batch_once` plus the range guard of the split-fp16 kernels (include/rapiddoc_mi355.h, rd_range_status):
        an engine that met an operand outside the fp16 range is switched to native fp32 for good and the batch is
        repeated, so a result is never silently wrong.  `pages`: [P,H,W,3] uint8 RGB on the GPU, or host pages (numpy / CPU tensor,
        what the reference's caller holds: batch_analyze.py:108-111), which are uploaded first - a stream of batches should go through
        `PageUploader` instead, whose copies run under the previous batch.

        `prefetch`: the NEXT batch's pages, already on the GPU (valid on the current stream).  Their det and layout forwards are
        enqueued as soon as this batch's det maps have been consumed, so they run under this batch's recognition instead of in
        front of the next batch's; the next `run_batch` call on that very tensor picks them up (any other call drops them).  The
        tensor is recognised by address and shape: its CONTENTS must not change between the two calls.  Results are the same with or
        without it (the same kernels on the same inputs); `last_det` then already belongs to the next batch when this call returns."""
        if isinstance(pages, np.ndarray) or not pages.is_cuda:
            src = torch.from_numpy(np.ascontiguousarray(pages)) if isinstance(pages, np.ndarray) else pages.contiguous()
            pages = src.to(self.tdev, non_blocking=src.is_pinned())
        self._page_keys = None if page_keys is None else [int(k) for k in page_keys]
        assert self._page_keys is None or len(self._page_keys) == pages.shape[0]
        if self.rec_width_sync is not None and page_keys is None and getattr(self.rec_width_sync, "world_size", 1) > 1:
            # local page indices would interleave the ranks' pages in the pooled order (every rank has a page 0)
            raise ValueError("rec_width_sync over more than one rank needs page_keys: the pages' positions in the GLOBAL page list")
        self._width_sync_epoch += 1
        self._in_run_batch = True
        try:
            return self._run_batch_guarded(pages, quads_per_page, det_maps_override, prefetch)
        finally:
            self._in_run_batch = False

    def _run_batch_guarded(self, pages, quads_per_page, det_maps_override, prefetch):
        results = self._run_batch_once(pages, quads_per_page, det_maps_override, prefetch)
        # det and the layout backbone (the rec engines are guarded inside rec_forward_lines)
        engines = [self.det] + ([self.layout] if self.layout is not None else [])
        if os.environ.get("RD_DEV_SKIP_RANGE_CHECK") == "1":
            engines = []
        if any([e.check_range_and_fallback() for e in engines]):
            self.stats["range_fallbacks"] = self.stats.get("range_fallbacks", 0) + 1
            self._prefetched = None        # made by the engine that has just been replaced: the next batch computes its own
            results = self._run_batch_once(pages, quads_per_page, det_maps_override, prefetch)
        return results

    @staticmethod
    def _front_key(pages: torch.Tensor) -> tuple:
        return (pages.data_ptr(), tuple(pages.shape))

    def _front(self, pages: torch.Tensor, n_results: int, gate: Sequence[torch.cuda.Event] = ()) -> dict:
        """Enqueue the batch's det forward (front stream) and, behind it, its layout backbone (layout stream).  The det forward starts
        after everything enqueued on the current stream so far - the pages' upload, and the previous batch's use of the det maps,
        whose buffer it overwrites - and after the `gate` events."""
        cur = torch.cuda.current_stream()
        h = {"key": self._front_key(pages), "feats": None, "layout_done": None}
        self.front_stream.wait_stream(cur)
        for ev in gate:
            self.front_stream.wait_event(ev)
        with torch.cuda.stream(self.front_stream):
            h["prob_maps"], h["det_hw"] = self.det_forward(pages)
            h["det_done"] = torch.cuda.Event()
            h["det_done"].record(self.front_stream)
        # the layout backbone keeps the GPU busy (on its own stream, so it also overlaps the recognition batches that follow)
        # while the host turns the det maps into boxes
        if self.layout is not None:
            self.layout_stream.wait_event(h["det_done"])       # det first: it is on the batch's critical path, the layout features are not
            with torch.cuda.stream(self.layout_stream):
                feats = self.layout_forward(pages)
                if self.keep_feats:                   # copies: the engine re-uses its output tensors on the next call of this shape
                    h["feats"] = [[f[i].clone() for f in feats] for i in range(n_results)]
                h["layout_done"] = torch.cuda.Event()
                h["layout_done"].record(self.layout_stream)
        return h

    def _run_batch_once(self, pages: torch.Tensor, quads_per_page: Optional[Sequence[np.ndarray]] = None,
                        det_maps_override: Optional[torch.Tensor] = None, prefetch: Optional[torch.Tensor] = None) -> List[PageResult]:
        """pages: [P,H,W,3] uint8 RGB on the GPU.  Text-line quads come from the DB post-process of the det maps
        unless `quads_per_page` is given.  `det_maps_override` (benchmark / tests with random weights, whose maps carry
        no text): device maps [P,1,h,w] that replace the network's output as the post-process input - the det forward
        still runs."""
        assert pages.is_cuda and pages.dtype == torch.uint8 and pages.dim() == 4
        P, H, W, _ = pages.shape
        results = [PageResult() for _ in range(P)]
        cur = torch.cuda.current_stream()
        h, self._prefetched = self._prefetched, None
        if h is None or h["key"] != self._front_key(pages):
            h = self._front(pages, P)
            self.stats["front_prefetched"] = 0.0
        else:
            self.stats["front_prefetched"] = 1.0
        cur.wait_event(h["det_done"])
        if self.rec_width_sync is not None and self.det.precision != "fp32" and os.environ.get("RD_DEV_SKIP_RANGE_CHECK") != "1":
            # page-sharded strict mode: the det maps decide which (key, ratio) list this rank hands to the width collective, so the det
            # range guard is settled BEFORE the recogniser stage - an overflowed map must neither feed the peers' widths nor make this rank
            # repeat a collective its peers make once (without the exchange the deferred check of _run_batch_guarded is enough)
            if self.det.check_range_and_fallback():
                self.stats["range_fallbacks"] = self.stats.get("range_fallbacks", 0) + 1
                h = self._front(pages, P)             # det (and layout) again, the det engine now on fp32 MFMA
                self.stats["front_prefetched"] = 0.0
                cur.wait_event(h["det_done"])
        prob_maps, det_hw = h["prob_maps"], h["det_hw"]
        self.last_det = (prob_maps, det_hw)         # a VIEW of the engine's output buffer: valid until the next det forward of this shape
        if quads_per_page is None:
            src = det_maps_override if det_maps_override is not None else prob_maps
            quads_per_page = self.boxes_from_maps_device(src.contiguous(), (H, W)) if self.device_db else self._boxes_via_host_maps(src, (H, W))
        if prefetch is not None:
            assert prefetch.is_cuda and prefetch.dtype == torch.uint8 and prefetch.dim() == 4

            def front_next(gate):
                self._prefetched = self._front(prefetch, prefetch.shape[0], gate)
            self._after_rec_enqueue = front_next
        try:
            keys = getattr(self, "_page_keys", None)
            texts = self.rec_forward_sources([(pages, quads_per_page)], image_keys=None if keys is None else [keys])[0]
        except BaseException:
            self._after_rec_enqueue = None             # a failed batch must not leave its hook to the next one
            raise
        if self._after_rec_enqueue is not None:        # no line to recognise: the hook was not reached
            self._after_rec_enqueue = None
            front_next(())
        if h["layout_done"] is not None:
            cur.wait_event(h["layout_done"])
        for i in range(P):
            qs = np.asarray(quads_per_page[i], dtype=np.float32).reshape(-1, 4, 2)
            results[i].lines = [(qs[j], t, s) for j, (t, s) in enumerate(texts[i])]
            if h["feats"] is not None:
                results[i].layout_feats = h["feats"][i]
        return results


def render_text_maps(boxes_per_page: Sequence[np.ndarray], page_hw: Tuple[int, int], map_hw: Tuple[int, int],
                     device) -> torch.Tensor:
    """Synthetic DB probability maps for pages whose text-line boxes are known (benchmark helper): 0.9 inside every
    line box shrunk by 0.32 x its height on every side (so that the post-process' unclip, ratio 1.8, grows the region
    back to about the original box), 0.02 elsewhere.  Used because random-weight det maps carry no text to find."""
    H, W = page_hw
    h, w = map_hw
    maps = torch.full((len(boxes_per_page), 1, h, w), 0.02, dtype=torch.float32)
    for i, boxes in enumerate(boxes_per_page):
        for x0, y0, x1, y1 in np.asarray(boxes, dtype=np.float64).reshape(-1, 4):
            bw, bh = x1 - x0, y1 - y0
            d = 0.32 * min(bw, bh)
            xa, xb = int(round((x0 + d) * w / W)), int(round((x1 - d) * w / W))
            ya, yb = int(round((y0 + d) * h / H)), int(round((y1 - d) * h / H))
            if xb > xa and yb > ya:
                maps[i, 0, ya:yb, xa:xb] = 0.9
    return maps.to(device)


class PageUploader:
    """Host -> device upload of a STREAM of page batches.  The reference hands host arrays to every batch
    (`np.array(image)` per page, batch_analyze.py:108-111); on the GPU the 6 MB a page weighs (192 MB per 32-page batch, ~4 ms of
    PCIe) must not sit in front of the networks, so: `n_buffers` device buffers, the copy of batch i + 1 enqueued on a copy stream of
    its own while batch i computes, and an event per batch that the consumer's stream waits on.

        up = PageUploader(device)
        nxt = up.submit(first_batch)                    # numpy [P,H,W,3] u8 or a CPU tensor (pinned: truly asynchronous)
        for batch in rest:
            cur, nxt = nxt, up.submit(batch)            # batch i + 1 travels under batch i's kernels
            results = pipe.run_batch(up.wait(cur))      # the current stream waits for ITS pages only
            up.release(cur)                             # everything that reads batch i is enqueued: its buffer may be recycled

    A buffer is re-used `n_buffers` submits later.  The copy that recycles it must run AFTER the last kernel that reads the old batch:
    `release(ticket)` records that point on the consumer's stream and the recycling `submit` makes the copy stream wait for it.  A
    ticket that was handed out by `wait` and never released is still safe as long as its consumers were ENQUEUED before the
    recycling submit: `submit` then waits for everything enqueued so far on the stream `wait` was called on (coarser than
    `release`: it also waits for later batches' work on that stream).  Pageable sources are staged through a pinned buffer of the
    uploader (one host memcpy); render into `pinned_like(...)` buffers to avoid it."""

    def __init__(self, device: int = 0, n_buffers: int = 2):
        self.tdev = torch.device("cuda", device)
        self.stream = torch.cuda.Stream(device=self.tdev)
        self.n = max(2, n_buffers)
        self._dev: List[Optional[torch.Tensor]] = [None] * self.n
        self._pin: List[Optional[torch.Tensor]] = [None] * self.n
        self._released: List[Optional[torch.cuda.Event]] = [None] * self.n
        self._consumer: List[Optional[torch.cuda.Stream]] = [None] * self.n    # stream `wait` handed buffer k out on, until `release`
        self._i = 0
        self.bytes_uploaded = 0

    @staticmethod
    def pinned_like(shape) -> torch.Tensor:
        """A pinned uint8 host buffer to render pages into (its upload needs no staging copy)."""
        return torch.empty(tuple(shape), dtype=torch.uint8, pin_memory=True)

    def submit(self, pages_host) -> Tuple[torch.Tensor, torch.cuda.Event, int]:
        src = torch.from_numpy(pages_host) if isinstance(pages_host, np.ndarray) else pages_host
        assert not src.is_cuda and src.dtype == torch.uint8 and src.dim() == 4 and src.is_contiguous()
        k = self._i % self.n
        self._i += 1
        if self._dev[k] is None or self._dev[k].numel() < src.numel():
            self._dev[k] = torch.empty(int(src.numel() * 1.1) + 256, dtype=torch.uint8, device=self.tdev)
        dst = self._dev[k][: src.numel()].view(src.shape)
        if self._consumer[k] is not None:                          # handed out and never released: wait for all of that stream's work so far
            rel = torch.cuda.Event()
            rel.record(self._consumer[k])
            self._released[k], self._consumer[k] = rel, None
        if self._released[k] is not None:
            self.stream.wait_event(self._released[k])              # the batch that used this buffer has been consumed (enqueued work done)
        if not src.is_pinned():
            if self._pin[k] is None or self._pin[k].numel() < src.numel():
                self._pin[k] = torch.empty(int(src.numel() * 1.1) + 256, dtype=torch.uint8, pin_memory=True)
            self.stream.synchronize()                              # the staging buffer's previous copy has left the host
            stage = self._pin[k][: src.numel()].view(src.shape)
            stage.copy_(src)
            src = stage
        with torch.cuda.stream(self.stream):
            dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(self.stream)
        self.bytes_uploaded += src.numel()
        return dst, ev, k

    def wait(self, ticket: Tuple[torch.Tensor, torch.cuda.Event, int]) -> torch.Tensor:
        """The device tensor of a submitted batch, valid on the CURRENT stream from here on."""
        dst, ev, k = ticket
        cur = torch.cuda.current_stream()
        cur.wait_event(ev)
        self._consumer[k] = cur
        return dst

    def release(self, ticket) -> None:
        """Call after the work that reads the batch has been enqueued on the current stream: its buffer may be recycled once that
        work is done."""
        _dst, _ev, k = ticket
        rel = torch.cuda.Event()
        rel.record(torch.cuda.current_stream())
        self._released[k], self._consumer[k] = rel, None


class PagePipelinePool:
    """`workers` PagePipelines on one GPU, each fed a contiguous shard of the page batch from its own host thread and HIP
    streams.  A page batch has host stages between its GPU stages (det maps -> DB post-process -> crop descriptors;
    rec indices -> CTC decode): with two shards in flight the GPU runs one shard's networks while the host works on the
    other's.  Results come back in page order, identical to a single pipeline's (pages are independent)."""

    def __init__(self, states: Dict[str, object], device: int = 0, workers: int = 2, **kw):
        self.device = device
        self.pipes = [PagePipeline(states, device, **kw) for _ in range(max(1, workers))]
        self.streams = [torch.cuda.Stream(device=self.pipes[0].tdev) for _ in self.pipes]
        self.pool = ThreadPoolExecutor(max_workers=len(self.pipes))
        self.stats: Dict[str, float] = {}

    @property
    def engines(self) -> List[RdEngine]:
        out = []
        for p in self.pipes:
            out += [p.det, *([p.layout] if p.layout is not None else []), *p.rec_engines, p.rec_tail]
        return out

    def run_batch(self, pages: torch.Tensor, quads_per_page: Optional[Sequence[np.ndarray]] = None,
                  det_maps_override: Optional[torch.Tensor] = None, prefetch: Optional[torch.Tensor] = None,
                  page_keys: Optional[Sequence[int]] = None) -> List[PageResult]:
        """`prefetch`: the next batch's device pages (PagePipeline.run_batch) - every shard's pipeline enqueues its shard of them.
        `page_keys`: PagePipeline.run_batch (global page indices of a page-sharded run; one worker only: the width collective is
        made once per batch by one pipeline)."""
        assert page_keys is None or len(self.pipes) == 1 or all(p.rec_width_sync is None for p in self.pipes)
        if isinstance(pages, np.ndarray) or not pages.is_cuda:      # host pages: one upload for all shards (see PagePipeline.run_batch)
            src = torch.from_numpy(np.ascontiguousarray(pages)) if isinstance(pages, np.ndarray) else pages.contiguous()
            pages = src.to(self.pipes[0].tdev, non_blocking=src.is_pinned())
        P = pages.shape[0]
        n = min(len(self.pipes), max(1, P))
        bounds = [P * k // n for k in range(n + 1)]
        nb = None
        if prefetch is not None and min(len(self.pipes), max(1, prefetch.shape[0])) == n:     # the next call will shard it the same way
            nb = [prefetch.shape[0] * k // n for k in range(n + 1)]
        ready = torch.cuda.Event()
        ready.record()

        def work(k: int):
            lo, hi = bounds[k], bounds[k + 1]
            torch.cuda.set_device(self.device)
            st = self.streams[k]
            with torch.cuda.stream(st):
                st.wait_event(ready)
                res = self.pipes[k].run_batch(pages[lo:hi], None if quads_per_page is None else quads_per_page[lo:hi],
                                              None if det_maps_override is None else det_maps_override[lo:hi],
                                              None if nb is None else prefetch[nb[k]:nb[k + 1]],
                                              None if page_keys is None else page_keys[lo:hi])
                done = torch.cuda.Event()
                done.record(st)
            return res, done

        outs = list(self.pool.map(work, range(n)))
        results: List[PageResult] = []
        for res, done in outs:
            torch.cuda.current_stream().wait_event(done)
            results += res
        self.stats = {}
        for k, p in enumerate(self.pipes[:n]):
            for key, v in p.stats.items():
                self.stats[key] = self.stats.get(key, 0.0) + v
        return results

