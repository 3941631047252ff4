"""OCR branch of the reference's page-batch driver, for GIVEN layout detections.

`BatchAnalyze.__call__` (rapid_doc/backend/pipeline/batch_analyze.py:78-164) runs layout first; its network's neck /
decoder exist only inside an ONNX file that is not available, so this module starts one step later: it takes
the per-page layout detections (the dict schema of `RapidLayoutModel.batch_predict`: `layout_host.to_layout_dets`) and
reproduces what follows for the text regions (and, `recognise_formulas`, for the formula regions), keeping every image on
the GPU:

  `_run_ocr_det_batch` (analyze_utils.py:105-212)
     crop each OCR region with a 50-px white margin, white out the formula boxes inside it, group the crops by language and
     by size rounded up to 64, pad each group with 255, detect, DB post-process, sort / merge the boxes, cut them around the
     formulas, map them back to page coordinates (`get_ocr_result_list`, utils/ocr_utils.py:361-431) -> OcrText spans
  `_run_ocr_rec_postprocess` (analyze_utils.py:216-292)
     recognise every span's line crop, write text / score, demote low-confidence spans

The lines of ALL regions of the page batch are recognised together (`PagePipeline.rec_forward_sources`), pooled page by
page like `_run_ocr_rec_postprocess` does per language; with a pipeline built with rec_mode="strict" the batching is the
reference's (one global argsort, chunks of 6), otherwise `rec_batch_num` is GPU sized.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import layout_host, layout_polygon, ocr_host, table_host
from .engine import preproc_resize_norm_batch

OCR_TEXT, LOW_SCORE_TEXT = 15, 16                 # utils/enum_class.py:103-104
MIN_CONFIDENCE, MIN_WIDTH = 0.5, 3                # utils/ocr_utils.py:9-11
PASTE = 50                                        # analyze_utils.py:131
_SPECIAL = ("（204号", "（20", "（2", "（2号", "（20号", "号", "（204")   # analyze_utils.py:288


def _formula_boxes_in_crop(formulas: Sequence[dict], useful: Sequence[int]) -> List[List[float]]:
    """get_adjusted_mfdetrec_res (utils/ocr_utils.py:320-342): formula boxes in crop coordinates, those outside dropped."""
    px, py, xmin, ymin, _xmax, _ymax, nw, nh = useful
    out = []
    for f in formulas:
        x0, y0, x1, y1 = f["bbox"]
        x0, y0, x1, y1 = x0 - xmin + px, y0 - ymin + py, x1 - xmin + px, y1 - ymin + py
        if x1 < 0 or y1 < 0 or x0 > nw or y0 > nh:
            continue
        out.append([x0, y0, x1, y1])
    return out


def _formulas_in_crop(formulas: Sequence[dict], useful: Sequence[int]):
    """get_adjusted_mfdetrec_res again, keeping the formula each box belongs to (its `latex` travels with it for tables)."""
    px, py, xmin, ymin, _xmax, _ymax, nw, nh = useful
    for f in formulas:
        x0, y0, x1, y1 = f["bbox"]
        x0, y0, x1, y1 = x0 - xmin + px, y0 - ymin + py, x1 - xmin + px, y1 - ymin + py
        if x1 < 0 or y1 < 0 or x0 > nw or y0 > nh:
            continue
        yield f, [x0, y0, x1, y1]


def restore_poly(poly: Sequence[float], angle: str, orig_w: int, orig_h: int) -> List[float]:
    """restore_poly (utils/boxbase.py:328-363): the axis-aligned box of a detection on the upright page -> the same box on the page as
    it came in.  Only (xmin, ymin, xmax, ymax) = poly[0], poly[1], poly[2], poly[5] are read."""
    xmin, ymin, xmax, ymax = poly[0], poly[1], poly[2], poly[5]
    if angle == "0":
        return poly
    if angle == "90":
        nx0, ny0, nx1, ny1 = orig_w - 1 - ymax, xmin, orig_w - 1 - ymin, xmax
    elif angle == "270":
        nx0, ny0, nx1, ny1 = ymin, orig_h - 1 - xmax, ymax, orig_h - 1 - xmin
    elif angle == "180":
        nx0, ny0, nx1, ny1 = orig_w - 1 - xmax, orig_h - 1 - ymax, orig_w - 1 - xmin, orig_h - 1 - ymin
    else:
        raise ValueError(f"unsupported angle: {angle}")
    return [nx0, ny0, nx1, ny0, nx1, ny1, nx0, ny1]


def _int_box(b, h: int, w: int) -> Optional[List[int]]:
    """normalize_to_int_bbox (utils/bbox_utils.py:6-55): floor / ceil, clip to the image, None if empty."""
    x0, y0 = int(np.floor(b[0])), int(np.floor(b[1]))
    x1, y1 = int(np.ceil(b[2])), int(np.ceil(b[3]))
    x0, y0, x1, y1 = max(0, x0), max(0, y0), min(w, x1), min(h, y1)
    return [x0, y0, x1, y1] if x1 > x0 and y1 > y0 else None


def crop_region(pages: torch.Tensor, p: int, det: dict) -> Optional[np.ndarray]:
    """`crop_img(det, page)[0]` with no margin (utils/model_utils.py:90-124): the integer box of the detection on a white canvas - page
    pixels where the box lies on the page, white outside it and outside the detection's polygon, if it has one.  RGB u8 numpy; None
    for an inverted box (the reference's np.ones of a negative size raises)."""
    _P, H, W, _ = pages.shape
    x0, y0, x1, y1 = (int(v) for v in (det["poly"][0], det["poly"][1], det["poly"][4], det["poly"][5]))
    if x1 < x0 or y1 < y0:
        return None
    crop = np.full((y1 - y0, x1 - x0, 3), 255, np.uint8)
    x0c, y0c, x1c, y1c = max(0, x0), max(0, y0), min(W, x1), min(H, y1)
    if x1c > x0c and y1c > y0c:
        part = pages[p, y0c:y1c, x0c:x1c].cpu().numpy()
        if det.get("polygon_points"):
            part = part.copy()
            part[~layout_polygon.polygon_keep_mask(part.shape[:2], det["polygon_points"], x0, y0)] = 255
        crop[y0c - y0: y1c - y0, x0c - x0: x1c - x0] = part
    return crop


def attach_table_images(pages: torch.Tensor, p: int, dets: Sequence[dict]) -> None:
    """Second half of get_res_list_from_layout_res (utils/model_utils.py:181-194): every image region lying inside a table is attached
    to that table's detection as `layout_image_list` = [{uuid, poly, pil_image}] (the table stage pops the field again; with tables
    switched off it stays in the output, as in the reference)."""
    import uuid
    from PIL import Image
    for im, tb in layout_host.images_inside_tables(dets):
        crop = crop_region(pages, p, im)
        if crop is None:
            continue
        tb.setdefault("layout_image_list", []).append({"uuid": str(uuid.uuid4()), "poly": im["poly"], "pil_image": Image.fromarray(crop)})


def recognise_formulas(pages: torch.Tensor, layout_dets_per_page: Sequence[Sequence[dict]], formula_model,
                       expand_px: int = 2, batch_size: int = 16) -> int:
    """Formula branch of BatchAnalyze (batch_analyze.py:258-283): every formula region (category 8 / 13 / 14) is cropped
    - grown by `bbox_expand_px` where no neighbouring layout box is in the way - and handed to
    `formula_model.batch_predict(images, batch_size=...)` (a `formula_host.FormulaRecognizer` or any CustomBaseModel-shaped
    object); a non-empty result is written to the detection's `latex` field IN PLACE.  Returns the number of formulas."""
    P, H, W, _ = pages.shape
    targets, crops = [], []
    for p, dets in enumerate(layout_dets_per_page):
        for d in dets:
            if int(d["category_id"]) not in layout_host.FORMULA_CATEGORY_IDS:
                continue
            c = layout_host.expand_formula_crop(d, dets, (H, W), expand_px)
            x0, y0, x1, y1 = int(c["poly"][0]), int(c["poly"][1]), int(c["poly"][4]), int(c["poly"][5])
            x0, y0, x1, y1 = max(0, x0), max(0, y0), min(W, x1), min(H, y1)
            if x1 <= x0 or y1 <= y0:
                continue
            targets.append(d)
            crop = pages[p, y0:y1, x0:x1].cpu().numpy()                  # RGB crop, no margin (crop_img with paste 0)
            if c.get("polygon_points"):                                  # crop_img whites out what lies outside the polygon
                crop = crop.copy()
                crop[~layout_polygon.polygon_keep_mask(crop.shape[:2], c["polygon_points"], int(c["poly"][0]), int(c["poly"][1]))] = 255
            crops.append(crop)
    if crops:
        for d, res in zip(targets, formula_model.batch_predict(crops, batch_size=batch_size)):
            if res:
                d["latex"] = res
    return len(crops)


class RegionOcr:
    def __init__(self, pipeline, box_thresh: float = 0.3, unclip_ratio: float = 1.8, lang: str = "ch",
                 det_batch_num: Optional[int] = None, det_raw_fn=None):
        """`pipeline`: a `rapiddoc_amd.pipeline.PagePipeline` (its det / rec engines and streams are reused).  box_thresh 0.3 /
        unclip 1.8 are the page-OCR settings of backend/pipeline/model_init.py:73.

        `det_batch_num`: the reference's `Det.rec_batch_num` (analyze_utils.py:118,186): the batch size handed to
        `det_batch_predict` is min(group size, det_batch_num); the engine here takes a group as one batch tensor whatever it
        is, the number only reaches `det_raw_fn`.
        `det_raw_fn(canvases [b,H64,W64,3] u8 RGB, batch_size) -> [raw boxes [n,4,2] per image]` replaces the detector
        (pre-process, network, DB post-process) of a size group - the seam `ocr_model.det_batch_predict` is in the reference
        (analyze_utils.py:187).  Used by tests/test_analyze_trace.py to replay traces of the reference's own driver."""
        self.pipe = pipeline
        self.box_thresh, self.unclip_ratio, self.lang = box_thresh, unclip_ratio, lang
        self.det_batch_num, self.det_raw_fn = det_batch_num, det_raw_fn

    def _pipe_for(self, lang: str):
        """`pipeline` may be one PagePipeline (every page is of language `lang`) or a dict {language: PagePipeline} - the reference
        asks its registry for one OCR model per language (analyze_utils.py:160-166, :246-250)."""
        if isinstance(self.pipe, dict):
            if lang not in self.pipe:
                raise KeyError(f"no OCR pipeline for language {lang!r} (have {sorted(self.pipe)})")
            return self.pipe[lang]
        return self.pipe

    # ------------------------------------------------------------------ det on one size group
    def _detect_group(self, canvases: torch.Tensor, maps_override: Optional[torch.Tensor] = None, pipe=None) -> List[np.ndarray]:
        """canvases [b, H64, W64, 3] u8 RGB on the GPU -> per image the detector's raw boxes [n,4,2] (group image coordinates,
        DB post-process order).  `maps_override` [b,1,dh,dw] replaces the network output as the post-process input (tests and
        benchmarks with random weights, whose maps carry no text); the det forward still runs."""
        pipe = pipe if pipe is not None else self._pipe_for(self.lang)
        b, H, W, _ = canvases.shape
        dh, dw = ocr_host.det_resize_shape(H, W, 960, "max")
        # DetPreProcess: BGR, (x/255 - Det.mean)/Det.std (rapid_ocr.py:61-62,474-536), the whole group in one launch
        x = preproc_resize_norm_batch(canvases, (dh, dw), mean=ocr_host.DET_MEAN, std=ocr_host.DET_STD, interp=1, swap_rb=True)
        maps = pipe.det.det_forward(x)
        if pipe.det.check_range_and_fallback():           # split-fp16 range guard (the pipeline's engines defer it)
            maps = pipe.det.det_forward(x)
        if maps_override is not None:
            assert tuple(maps_override.shape) == tuple(maps.shape)
            maps = maps_override
        # DB post-process with the maps staying in HBM (ocr_host.db_postprocess_device: everything on the device, one copy back)
        res = ocr_host.db_postprocess_device(maps.contiguous(), [(H, W)] * b, thresh=0.3, box_thresh=self.box_thresh,
                                             unclip_ratio=self.unclip_ratio, cache=pipe.db_ws)
        return [boxes.astype(np.float32) for boxes, _scores in res]

    @staticmethod
    def _sort_merge(raw_boxes) -> np.ndarray:
        """analyze_utils.py:193-196: reading-order sort, then same-line merge (nothing for an empty result)."""
        if raw_boxes is None or len(raw_boxes) == 0:
            return np.zeros((0, 4, 2), np.float32)
        q = ocr_host.merge_det_boxes(ocr_host.sorted_boxes(np.asarray(raw_boxes, dtype=np.float32)))
        return np.asarray(q, dtype=np.float32).reshape(-1, 4, 2)

    # ------------------------------------------------------------------ whole batch
    def __call__(self, pages: torch.Tensor, layout_dets_per_page: Sequence[Sequence[dict]], det_maps_fn=None,
                 page_langs: Optional[Sequence[str]] = None, mask_boxes_per_page: Optional[Sequence[Sequence[dict]]] = None,
                 page_keys: Optional[Sequence[int]] = None, all_langs: Optional[Sequence[str]] = None) -> List[List[dict]]:
        """pages [P,H,W,3] u8 RGB (GPU); returns, per page, the layout detections followed by their OcrText spans
        (`layout_res` of the reference after both OCR stages).  `det_maps_fn(regions, (gh, gw), (dh, dw))` may supply the
        det maps of a size group (see `_detect_group`); regions = [(page, region dict, useful_list)].
        `page_langs[p]`: the language of page p (the `lang` of the reference's input tuples); regions are grouped by language first
        and every language's lines are recognised by that language's pipeline in one pooled call.
        `mask_boxes_per_page[p]`: further {'bbox': ...} entries treated like the page's formulas - the reference's `checkbox_res`
        (`single_page_mfdetrec_res + checkbox_res`, analyze_utils.py:133-136).
        Page-sharded runs (the pipeline's `rec_width_sync` set, rapiddoc_amd.dist.GlobalLineWidths): `page_keys[p]` = page p's
        position in the GLOBAL page list - the pooling key of the recogniser's width exchange, which makes the strings and scores
        those of the unsharded batch; `all_langs` = every language of the global batch in one agreed order (default: this
        object's language) - each rank makes one pooled recogniser call per language, also for a language it holds no region of,
        so the exchanges pair up across ranks."""
        assert pages.dtype == torch.uint8 and (pages.is_cuda or self.det_raw_fn is not None)
        P, H, W, _ = pages.shape
        assert page_keys is None or len(page_keys) == P
        pipes = list(self.pipe.values()) if isinstance(self.pipe, dict) else [self.pipe]
        synced = any(getattr(pp, "rec_width_sync", None) is not None for pp in pipes)
        if synced and page_keys is None:
            raise ValueError("rec_width_sync is set on the OCR pipeline: hand over page_keys (the pages' positions in the global page list)")
        out: List[List[dict]] = [list(d) for d in layout_dets_per_page]
        regions = []                                   # (page, region dict, useful_list, formula boxes in crop coords)
        for p, dets in enumerate(layout_dets_per_page):
            ocr_regions, _tables, formulas = layout_host.split_regions(dets)
            if mask_boxes_per_page is not None:
                formulas = list(formulas) + list(mask_boxes_per_page[p])
            for r in ocr_regions:
                useful = layout_host.crop_geometry(r, PASTE, PASTE)
                if useful[6] < 2 * PASTE or useful[7] < 2 * PASTE:        # inverted box: the reference's np.ones would raise
                    continue
                regions.append((p, r, useful, _formula_boxes_in_crop(formulas, useful)))
        if not regions and not synced:
            return out
        langs = [self.lang if page_langs is None else page_langs[p] for p, _r, _u, _f in regions]
        groups = ocr_host.det_buckets([(u[7], u[6]) for _, _, u, _ in regions], langs, det_batch_num=len(regions))
        pending = []                                   # per size group: (unmasked canvases, quads, spans, page of each image)
        for _lang, (gh, gw), members, _bs in groups:
            canv = torch.full((len(members), gh, gw, 3), 255, dtype=torch.uint8, device=pages.device)
            # The reference whites the formula boxes out of a COPY that only the detector sees (`det_image`,
            # analyze_utils.py:136-145); the line crops for the recogniser come from the unmasked region image
            # (`bgr_image`, analyze_utils.py:199 -> ocr_utils.py:361-383).  Two canvases: `canv` unmasked, `det_canv` masked.
            for k, ridx in enumerate(members):
                p, _r, (px, py, x0, y0, x1, y1, _nw, _nh), fboxes = regions[ridx]
                x0c, y0c, x1c, y1c = max(0, x0), max(0, y0), min(W, x1), min(H, y1)      # numpy slicing clips the same way
                if x1c > x0c and y1c > y0c:
                    dst = canv[k, py + (y0c - y0): py + (y0c - y0) + (y1c - y0c), px + (x0c - x0): px + (x0c - x0) + (x1c - x0c)]
                    dst[:] = pages[p, y0c:y1c, x0c:x1c]
                    if _r.get("polygon_points"):       # crop_img (model_utils.py:109-118): outside the region's polygon -> white,
                        keep = layout_polygon.polygon_keep_mask((y1c - y0c, x1c - x0c), _r["polygon_points"], x0, y0)   # for det AND rec
                        dst[~torch.from_numpy(keep).to(dst.device)] = 255
            det_canv = canv.clone() if any(regions[ridx][3] for ridx in members) else canv
            for k, ridx in enumerate(members):
                nh, nw = regions[ridx][2][7], regions[ridx][2][6]
                for fb in regions[ridx][3]:             # _apply_mask_boxes_to_image (analyze_utils.py:82-103)
                    ib = _int_box(fb, nh, nw)
                    if ib:
                        det_canv[k, ib[1]:ib[3], ib[0]:ib[2]] = 255
            override = None
            if det_maps_fn is not None:
                override = det_maps_fn([regions[i][:3] for i in members], (gh, gw), ocr_host.det_resize_shape(gh, gw, 960, "max"))
            if self.det_raw_fn is not None:
                fn = self.det_raw_fn[_lang] if isinstance(self.det_raw_fn, dict) else self.det_raw_fn
                raw = fn(det_canv, min(len(members), self.det_batch_num or len(members)))
            else:
                raw = self._detect_group(det_canv, override, self._pipe_for(_lang))
            boxes_per_img = [self._sort_merge(r) for r in raw]
            spans_per_img: List[List[dict]] = []
            quads_per_img: List[np.ndarray] = []
            for k, ridx in enumerate(members):
                p, r, useful, fboxes = regions[ridx]
                boxes = list(boxes_per_img[k])
                if boxes and fboxes:
                    boxes = ocr_host.update_det_boxes(boxes, [{"bbox": fb} for fb in fboxes])
                px, py, x0, y0 = useful[0], useful[1], useful[2], useful[3]
                spans, quads = [], []
                for q in boxes:
                    q = np.asarray(q, dtype=np.float32).reshape(4, 2)
                    p1, p2, p3, p4 = [list(map(float, pt)) for pt in q]
                    if p3[0] - p1[0] < MIN_WIDTH:
                        continue
                    if ocr_host.quad_is_tilted([p1, p2, p3, p4]):      # utils/ocr_utils.py:392-404
                        xc, yc = sum(pt[0] for pt in (p1, p2, p3, p4)) / 4, sum(pt[1] for pt in (p1, p2, p3, p4)) / 4
                        nh_, nw_ = ((p4[1] - p1[1]) + (p3[1] - p2[1])) / 2, p3[0] - p1[0]
                        p1, p2 = [xc - nw_ / 2, yc - nh_ / 2], [xc + nw_ / 2, yc - nh_ / 2]
                        p3, p4 = [xc + nw_ / 2, yc + nh_ / 2], [xc - nw_ / 2, yc + nh_ / 2]
                    poly = []
                    for pt in (p1, p2, p3, p4):
                        poly += [float(pt[0] - px + x0), float(pt[1] - py + y0)]
                    spans.append({"category_id": OCR_TEXT, "original_label": r.get("original_label"),
                                  "original_order": r.get("original_order", -1), "poly": poly, "score": 1, "text": ""})
                    quads.append(q)                    # the line crop is taken from the UNcorrected box (:381-383)
                spans_per_img.append(spans)
                quads_per_img.append(np.asarray(quads, dtype=np.float32).reshape(-1, 4, 2))
            pending.append((canv, quads_per_img, spans_per_img, [regions[ridx][0] for ridx in members], _lang))
        if not pending and not synced:
            return out
        # rec: every line of the page batch - per language - in ONE pooled call (analyze_utils.py:216-252), ordered page by page
        texts_all: List = [None] * len(pending)
        key_of = (lambda p: p) if page_keys is None else (lambda p: int(page_keys[p]))
        rec_langs = list(dict.fromkeys(e[4] for e in pending))
        if synced:                                     # the agreed language list: every rank makes the same pooled calls
            agreed = list(all_langs) if all_langs is not None else [self.lang]
            missing = [lg for lg in rec_langs if lg not in agreed]
            if missing:
                raise ValueError(f"languages {missing} are not in all_langs {agreed}: a page-sharded run needs the global list")
            rec_langs = agreed
        for lang in rec_langs:
            idx = [i for i, e in enumerate(pending) if e[4] == lang]
            res = self._pipe_for(lang).rec_forward_sources([(pending[i][0], pending[i][1]) for i in idx],
                                                           image_keys=[[key_of(p) for p in pending[i][3]] for i in idx])
            for i, r in zip(idx, res):
                texts_all[i] = r
        for (canv, _q, spans_per_img, pages_of, _lg), texts in zip(pending, texts_all):
            for k, p in enumerate(pages_of):
                for span, (text, score) in zip(spans_per_img[k], texts[k]):
                    span["text"] = text
                    span["score"] = float(f"{score:.3f}")
                    if score < MIN_CONFIDENCE:
                        span["category_id"] = LOW_SCORE_TEXT
                    else:
                        w_, h_ = span["poly"][4] - span["poly"][0], span["poly"][5] - span["poly"][1]
                        if text in _SPECIAL and score < 0.8 and w_ < h_:
                            span["category_id"] = LOW_SCORE_TEXT
                    out[p].append(span)
        return out


def table_crop_rect(det: dict) -> Optional[List[int]]:
    """The table crop of the reference (batch_analyze.py:235-243): the box divided by 5, floor / ceil to integers
    (normalize_to_int_bbox), times 5 again - i.e. snapped OUTWARDS to multiples of 5 px; slicing clips it at the page's far edges.
    None for an empty box (the reference's own unpacking fails on that path; such a region is skipped here)."""
    p = det["poly"]
    b = np.asarray([float(v) / float(5) for v in (p[0], p[1], p[4], p[5])], dtype=np.float64)
    x0, y0, x1, y1 = math.floor(b[0]), math.floor(b[1]), math.ceil(b[2]), math.ceil(b[3])
    if x1 <= x0 or y1 <= y0:
        return None
    return [int(x0 * 5), int(y0 * 5), int(x1 * 5), int(y1 * 5)]


class TableOcr:
    """The reference's own table stage for a `predict`-shaped table model (seam S3, RapidTableModel.predict, rapid_table.py:120):
    `_process_single_table` (analyze_utils.py:295-427) for pages that are OCR-ed (`ocr_enable`; the PDF-text-layer variant and the
    orientation sub-stage are outside SURVEY s8 - a table is taken as upright, which is what the reference does when its
    classifier answers "0").  Per table region, in page / layout order:

      crop (`table_crop_rect`) -> formula boxes of the page in crop coordinates, with their `latex` (get_adjusted_mfdetrec_res,
      return_text) -> detector on the crop with those boxes whited out, box_thresh 0.5 / unclip 1.6, boxes sorted and cut around the
      formulas but NOT merged (`ocr(det, rec=False)`, rapid_ocr.py:257-281) -> every line cropped from the UNmasked image and
      recognised in one call (`_run_table_ocr` :478-540).  `use_word_box` (the reference's default, analyze_utils.py:308): the OCR result
      handed on is one entry per WORD / CJK character - `ocr(det=False, return_word_box=True)` (rapid_ocr.py:282-299) -> per line the
      words with boxes in table coordinates (rapiddoc_amd/word_boxes.py: RapidDoc's patched get_word_info / cal_ocr_word_box and
      calc_word_boxes pinned to the reference, rapidocr's own box arithmetic restated and unpinned), lines whose word list came back
      empty shift the pairing exactly as the reference's zip does (rapid_ocr.py:295); `use_word_box=False`: one entry per line
      -> texts through `normalize_table_ocr_text` -> `table_model.predict(table_img RGB, [boxes, texts,
      scores], fill_image_res, formula boxes, skip_text_in_image, use_img2table, skip_table_orientation=True)` -> the
      `<table>...</table>` part of the answer becomes the region's `html`, and `formula_boxes` = every formula box of the page
      divided by the page's render scale (:405-418).

    The detector / recogniser are the pipeline's GPU engines; `det_raw_fn(canvas [1,h,w,3] u8 RGB, 1) -> [raw boxes [n,4,2]]` and
    `rec_fn(canvas [1,h,w,3], quads [n,4,2]) -> [(text, score)]` (with `use_word_box`: `[(text, score, [(word, conf, box) ...])]`, what
    rapidocr's cal_rec_boxes leaves per line) replace them (tests replaying traces of the reference)."""

    _warned_line_level = False       # the use_word_box downgrade is reported once per process

    def __init__(self, pipeline, det_raw_fn=None, rec_fn=None, skip_text_in_image: bool = True, use_img2table: bool = False,
                 table_formula_enable: bool = True, lang: str = "ch", use_word_box: bool = True):
        self.pipe, self.rec_fn, self.use_word_box = pipeline, rec_fn, use_word_box
        self.det = RegionOcr(pipeline, box_thresh=0.5, unclip_ratio=1.6, lang=lang, det_raw_fn=det_raw_fn)
        self.skip_text_in_image, self.use_img2table, self.table_formula_enable = skip_text_in_image, use_img2table, table_formula_enable

    def ocr_result(self, table: torch.Tensor, adjusted: Optional[List[dict]], maps_override: Optional[torch.Tensor] = None,
                   lang: Optional[str] = None) -> list:
        """table [h,w,3] u8 RGB -> [boxes, texts, scores] (three parallel lists) or [] when nothing was detected.  `maps_override`
        [1,1,dh,dw]: see RegionOcr._detect_group."""
        h, w, _ = table.shape
        canvas = table.unsqueeze(0)
        det_canvas = canvas
        if adjusted:
            det_canvas = canvas.clone()
            for f in adjusted:                          # _apply_mask_boxes_to_image
                ib = _int_box(f["bbox"], h, w)
                if ib:
                    det_canvas[0, ib[1]:ib[3], ib[0]:ib[2]] = 255
        if self.det.det_raw_fn is not None:
            raw = self.det.det_raw_fn(det_canvas, 1)[0]
        else:
            raw = self.det._detect_group(det_canvas.contiguous(), maps_override, self.det._pipe_for(lang or self.det.lang))[0]
        if raw is None or len(raw) == 0:
            return []
        boxes = list(ocr_host.sorted_boxes(np.asarray(raw, dtype=np.float32)))
        if adjusted:
            boxes = ocr_host.update_det_boxes(boxes, adjusted)
        if not boxes:
            return []
        quads = np.asarray(boxes, dtype=np.float32).reshape(-1, 4, 2)
        # word boxes come from the strict two-stage recogniser (per-line kept time steps); a pipeline built in another batching mode
        # serves the table model line-level entries (the reference's use_word_box=False shape) instead of failing in the table stage
        pipe = None if self.rec_fn is not None else self.det._pipe_for(lang or self.det.lang)     # (a custom rec_fn needs no pipeline of that language)
        words_ok = pipe is None or (getattr(pipe, "rec_mode", None) == "strict" and getattr(pipe, "rec_two_stage", False))
        if self.use_word_box and words_ok:
            return self._word_level(canvas, quads, h, w, lang)
        if self.use_word_box and not TableOcr._warned_line_level:
            TableOcr._warned_line_level = True
            import warnings
            warnings.warn("TableOcr: use_word_box=True needs the strict two-stage recogniser (rec_mode='strict'); this pipeline serves the "
                          "table model LINE-level entries (the reference's use_word_box=False input shape)", RuntimeWarning, stacklevel=2)
        if self.rec_fn is not None:
            lines = self.rec_fn(canvas, quads)
        else:
            lines = pipe.rec_forward_sources([(canvas.contiguous(), [quads])], image_keys=[[0]], pooled=False)[0][0]
        return [[q for q in quads], [table_host.normalize_table_ocr_text(t) for t, _s in lines], [s for _t, s in lines]]

    def _word_level(self, canvas: torch.Tensor, quads: np.ndarray, h: int, w: int, lang: Optional[str]) -> list:
        """`_run_table_ocr` with table_use_word_box (analyze_utils.py:478-540): [word boxes, word texts, word confidences]."""
        from . import word_boxes as WB
        if self.rec_fn is not None:
            lines = self.rec_fn(canvas, quads)                     # [(text, score, [(word, conf, box)])]
        else:
            raw = self.det._pipe_for(lang or self.det.lang).rec_forward_sources([(canvas.contiguous(), [quads])], image_keys=[[0]],
                                                                                want_words=True, pooled=False)[0][0]
            # lines the recogniser marked degenerate (no homography: `ws` None) or read nothing in carry no words: they never reach the
            # box arithmetic (whose inverse rotation would divide by a zero side / solve a singular system) and drop out below
            good = [i for i, (t, _s, ws) in enumerate(raw) if ws and t]
            infos = [WB.decode_word_info(raw[i][0], raw[i][2]["cols"], raw[i][2]["confs"], raw[i][2]["n_steps"], raw[i][2]["wh_ratio"],
                                         raw[i][2]["max_wh_ratio"]) for i in good]
            word_good = WB.cal_rec_boxes([raw[i][2]["crop_hw"] for i in good], [quads[i] for i in good], [raw[i][0] for i in good], infos)
            word_lines: List[list] = [[] for _ in raw]
            for i, wl in zip(good, word_good):
                word_lines[i] = wl
            lines = [(t, s, wl) for (t, s, _w), wl in zip(raw, word_lines)]
        origin = WB.calc_word_boxes([wl for _t, _s, wl in lines], h, w)     # (lines without words drop out: rapid_ocr.py:325-326)
        rec_res = list(zip([t for t, _s, _w in lines], [s for _t, s, _w in lines], origin))      # rapid_ocr.py:295 - zip, as it is
        out = []
        for _q, res in zip(quads, rec_res):
            out.extend([wr[2], table_host.normalize_table_ocr_text(wr[0]), wr[1]] for wr in res[2])
        return [list(x) for x in zip(*out)] if out else []

    def __call__(self, pages: torch.Tensor, layout_dets_per_page: Sequence[Sequence[dict]], table_model,
                 page_scales: Optional[Sequence[float]] = None, det_maps_fn=None, table_image_enable: bool = True,
                 page_langs: Optional[Sequence[str]] = None, mask_boxes_per_page: Optional[Sequence[Sequence[dict]]] = None) -> int:
        """Writes `html` (and `formula_boxes` / `img_boxes`) into the table detections IN PLACE; returns the number of tables handed to
        the model.  `table_image_enable`: hand the images that lie inside a table (`layout_image_list`) to the model as
        `fill_image_res` (extract_table_fill_image's layout branch).  `mask_boxes_per_page[p]`: the page's `checkbox_res`, which the
        reference appends to the formulas everywhere in this stage (analyze_utils.py:318-321,411-415); their `checkbox` text travels
        to the table model like a formula's `latex`."""
        n = 0
        for p, dets in enumerate(layout_dets_per_page):
            _ocr, tables, formulas = layout_host.split_regions(dets)
            if mask_boxes_per_page is not None:
                formulas = list(formulas) + list(mask_boxes_per_page[p])
            scale = 1.0 if page_scales is None else page_scales[p]
            for t in tables:
                rect = table_crop_rect(t)
                if rect is None:
                    continue
                x0, y0, x1, y1 = rect
                table = pages[p, y0:y1, x0:x1]
                useful = [0, 0, x0, y0, x1, y1, int(table.shape[1]), int(table.shape[0])]
                adjusted = None
                if self.table_formula_enable:
                    adjusted = []
                    for f, box in _formulas_in_crop(formulas, useful):
                        a = {"bbox": box}
                        if f.get("latex"):
                            a["latex"] = f["latex"]
                        if f.get("checkbox"):
                            a["checkbox"] = f["checkbox"]
                        adjusted.append(a)
                override = None
                if det_maps_fn is not None and table.numel():      # (page, crop rectangle, det input size) -> maps [1,1,dh,dw]
                    override = det_maps_fn(p, (x0, y0, x0 + int(table.shape[1]), y0 + int(table.shape[0])),
                                           ocr_host.det_resize_shape(int(table.shape[0]), int(table.shape[1]), 960, "max"))
                ocr_result = self.ocr_result(table, adjusted, override, None if page_langs is None else page_langs[p]) if table.numel() else []
                fill = layout_host.table_fill_images(t, useful) if table_image_enable else []
                t.pop("layout_image_list", None)
                html_code = table_model.predict(table.cpu().numpy(), ocr_result, fill, adjusted, self.skip_text_in_image, self.use_img2table,
                                                skip_table_orientation=True)
                n += 1
                if html_code and "<table>" in html_code and "</table>" in html_code:
                    t["html"] = html_code[html_code.find("<table>"): html_code.rfind("</table>") + len("</table>")]
                    fboxes = [f["bbox"] for f in formulas if "bbox" in f]
                    if fboxes:
                        t["formula_boxes"] = [[int(c / scale) for c in b] for b in fboxes]
                    iboxes = [f["ori_bbox"] for f in fill if "bbox" in f]
                    if iboxes:
                        t["img_boxes"] = [[int(c / scale) for c in b] for b in iboxes]
        return n


class RegionTextModel:
    """The OCR seam S1 of the reference: an `ocr_config['custom_model']` object
    (`rapid_doc.model.custom.CustomBaseModel.batch_predict(image_list, **kwargs) -> list[str]`, model/custom/__init__.py:4-20),
    called by `BatchAnalyze._run_custom_ocr` (batch_analyze.py:286-331) with the BGR crop of every text REGION and expected
    to return that region's text, lines separated by newlines; the reference wraps each string into one OcrText item.
    Here a region is read the conventional way - det, box sort / merge, rec - on the GPU, and its lines with a
    confidence >= 0.5 are joined in reading order."""

    def __init__(self, pipeline, box_thresh: float = 0.3, unclip_ratio: float = 1.8):
        self._ocr = RegionOcr(pipeline, box_thresh, unclip_ratio)

    def batch_predict(self, image_list: Sequence[np.ndarray], det_maps_fn=None, **kwargs) -> List[str]:
        pipe = self._ocr.pipe
        texts = [""] * len(image_list)
        shapes = [(int(im.shape[0]), int(im.shape[1])) for im in image_list]
        keep = [i for i, (h, w) in enumerate(shapes) if h > 0 and w > 0]
        groups = ocr_host.det_buckets([shapes[i] for i in keep], ["_"] * len(keep), det_batch_num=max(1, len(keep)))
        pending = []
        for _lang, (gh, gw), members, _bs in groups:
            ids = [keep[m] for m in members]
            canv = torch.full((len(ids), gh, gw, 3), 255, dtype=torch.uint8, device=pipe.tdev)
            for k, i in enumerate(ids):
                rgb = np.ascontiguousarray(np.asarray(image_list[i], dtype=np.uint8)[:, :, ::-1])     # BGR -> RGB
                canv[k, : shapes[i][0], : shapes[i][1]] = torch.from_numpy(rgb).to(pipe.tdev)
            override = det_maps_fn(ids, (gh, gw), ocr_host.det_resize_shape(gh, gw, 960, "max")) if det_maps_fn else None
            boxes = [self._ocr._sort_merge(r) for r in self._ocr._detect_group(canv, override)]
            quads = [np.asarray([q for q in b if q[2][0] - q[0][0] >= MIN_WIDTH], dtype=np.float32).reshape(-1, 4, 2) for b in boxes]
            pending.append((canv, quads, ids))
        if pending:
            lines_all = pipe.rec_forward_sources([(c, q) for c, q, _i in pending], image_keys=[ids for _c, _q, ids in pending], pooled=False)
            for (_c, _q, ids), lines in zip(pending, lines_all):
                for k, i in enumerate(ids):
                    texts[i] = "\n".join(t for t, s in lines[k] if s >= MIN_CONFIDENCE and t)
        return texts


class PageAnalyzer:
    """Stage sequencing of `BatchAnalyze.__call__` (rapid_doc/backend/pipeline/batch_analyze.py:78-164) for one page batch
    resident on the GPU:

        1. layout        `layout_model.batch_predict(pages, batch_size)`  (:165-189: + `filter_overlap_boxes`, inline formulas
                         dropped when formulas are disabled or formula_level == 1)
        2. collection    OCR / table / formula regions per page (`split_regions` = get_res_list_from_layout_res)
        3. formulas      `recognise_formulas` (:258-283) when a formula model is given
        4. OCR           det + rec of every text region (`RegionOcr`), or the custom-OCR seam (`RegionTextModel`-shaped object:
                         one string per region, :286-333) when `custom_ocr` is given
        5. tables        `table_model.batch_predict(table crops, fill_image_res_list=...)` for a CustomBaseModel-shaped model (seam
                         S1, :359-379), or `TableOcr` + `table_model.predict(...)` for a RapidTableModel-shaped one (seam S3,
                         `_process_single_table`); the reference's own table NETWORKS are ONNX-only and not built
        6. (rec post-process is part of RegionOcr here: spans get text / score / LowScoreText demotion, analyze_utils.py:216-292)

    Every page is processed independently (`pages[p]` only feeds `out[p]`); the result is the reference's
    `images_layout_res`: per page the filtered layout detections, formula `latex` fields filled in place, followed by the
    OcrText spans, then (7.) `text` written into the seal regions (`_run_seal_ocr`).  Page orientation (0., off by
    default: USE_DOC_ORIENTATION_CLASSIFY) and checkbox detection (off by default) are seams for caller-supplied models, sequenced as
    in the reference.  Not built: the 'txt' det mode (PDF text layer)."""

    def __init__(self, layout_model, pipeline, formula_model=None, table_model=None, custom_ocr=None, layout_batch_size: int = 1,
                 formula_level: int = 0, box_thresh: float = 0.3, unclip_ratio: float = 1.8, formula_batch_size: int = 1,
                 formula_expand_px: int = 2, det_batch_num: Optional[int] = None, det_raw_fn=None, lang: str = "ch",
                 table_det_raw_fn=None, table_rec_fn=None, table_image_enable: bool = True, table_use_word_box: bool = True,
                 seal_model=None, seal_enable: bool = True, checkbox_fn=None, checkbox_enable: bool = False,
                 orientation_model=None, use_doc_orientation_classify: bool = False):
        """Batch sizes default to the reference's (layout_config['batch_num'] / formula_config['batch_num'] = 1,
        batch_analyze.py:66-71); `det_batch_num` / `det_raw_fn`: see RegionOcr."""
        self.layout_model, self.pipe = layout_model, pipeline
        self.formula_model, self.table_model, self.custom_ocr = formula_model, table_model, custom_ocr
        self.layout_batch_size, self.formula_level = layout_batch_size, formula_level
        self.formula_batch_size, self.formula_expand_px = formula_batch_size, formula_expand_px
        self.ocr = RegionOcr(pipeline, box_thresh, unclip_ratio, lang, det_batch_num, det_raw_fn)
        # table_config["use_word_box"], default True (analyze_utils.py:308): word-level OCR entries for the table model
        self.table_ocr = TableOcr(pipeline, det_raw_fn=table_det_raw_fn, rec_fn=table_rec_fn, lang=lang, use_word_box=table_use_word_box)
        self.table_image_enable = table_image_enable          # table_config["table_image_enable"], default True (batch_analyze.py:75)
        # 7. seal OCR (ocr_config["seal_enable"], default True, batch_analyze.py:62,150-151): `seal_model` = the RapidOcrModel-shaped object
        #    the reference obtains with get_atom_model(OCR, is_seal=True) - `.ocr(bgr crop, det=True, rec=True) -> [[(box, (text, score)), ...]]`.
        #    Its seal detector is ONNX-only (not built): pages WITH a seal region and no model to read it fail loudly.
        self.seal_model, self.seal_enable = seal_model, seal_enable
        # checkbox detection (checkbox_config["checkbox_enable"], default False, batch_analyze.py:51,207-219): `checkbox_fn(bgr page) ->
        # [{'bbox': [x0, y0, x1, y1], 'text': ...}]` = utils/checkbox_det_cls.py checkbox_predict (OpenCV morphology on the host; not
        # built - a caller that enables the stage supplies it).  Every hit becomes a CheckBox detection of the page and masks the OCR
        # detector's input like a formula box.
        self.checkbox_fn, self.checkbox_enable = checkbox_fn, checkbox_enable
        if checkbox_enable and checkbox_fn is None:
            raise ValueError("checkbox_enable needs a checkbox_fn (the reference's checkbox_predict is host OpenCV code, not part of this build)")
        # page orientation (USE_DOC_ORIENTATION_CLASSIFY, default off, batch_analyze.py:66-67,113-125,153-161): `orientation_model.predict(
        # rgb page) -> "0" | "90" | "180" | "270"` (the ONNX-only ImgOrientationCls model); pages labelled 90 / 270 are turned upright
        # before the layout model sees them and every detection's `poly` is mapped back afterwards
        self.orientation_model, self.use_doc_orientation_classify = orientation_model, use_doc_orientation_classify
        if use_doc_orientation_classify and orientation_model is None:
            raise ValueError("use_doc_orientation_classify needs an orientation_model (ONNX-only in the reference, not part of this build)")
        self.last_rotate_labels: List[str] = []

    def __call__(self, pages: torch.Tensor, det_maps_fn=None, page_scales: Optional[Sequence[float]] = None,
                 table_det_maps_fn=None, page_langs: Optional[Sequence[str]] = None,
                 page_keys: Optional[Sequence[int]] = None, all_langs: Optional[Sequence[str]] = None) -> List[List[dict]]:
        """`page_keys` / `all_langs`: page-sharded runs, see RegionOcr.__call__ (the OCR stage's pooled recogniser call is the one step
        of the batch whose result depends on the other pages; tables, formulas and layout are per region / per page).
        `page_scales[p]`: the render scale the reference carries with every page (the `scale` of its input tuples); only the
        `formula_boxes` written next to a table's `html` use it (analyze_utils.py:405-418).  Default 1."""
        assert pages.dtype == torch.uint8 and pages.dim() == 4 and (pages.is_cuda or self.ocr.det_raw_fn is not None)
        # 0. page orientation (off by default): classify every page, turn the 90 / 270 ones upright (get_rotate_image, utils/boxbase.py:
        #    312-326: "270" -> cv2.ROTATE_90_CLOCKWISE, "90" -> counter-clockwise; 0 and 180 are left as they are)
        orig_hw_label = []
        if self.use_doc_orientation_classify:
            labels = [self.orientation_model.predict(pages[i].cpu().numpy()) for i in range(pages.shape[0])]
            turned = [lb in ("90", "270") for lb in labels]
            if any(turned) and not all(turned):
                raise NotImplementedError("a batch that mixes upright and sideways pages: the page tensor is one [P,H,W,3] array - "
                                          "hand the two groups over as separate batches")
            orig_hw_label = [(int(pages.shape[1]), int(pages.shape[2]), lb) for lb in labels]
            if all(turned) and labels:
                pages = torch.stack([torch.rot90(pages[i], 1 if lb == "90" else -1, dims=(0, 1)) for i, lb in enumerate(labels)]).contiguous()
            self.last_rotate_labels = list(labels)
        P, H, W, _ = pages.shape
        use_custom = self.custom_ocr is not None
        # 1. layout (+ overlap filter, formula level)
        dets = self.layout_model.batch_predict([pages[i] for i in range(P)], self.layout_batch_size)
        dets = [layout_host.filter_overlap_boxes(d, use_custom) for d in dets]
        if self.formula_model is None or self.formula_level == 1:
            inline = layout_host.CATEGORY_ID["InlineEquation"]
            dets = [[d for d in page if d["category_id"] != inline] for page in dets]
        # a seal region nobody can read must stop the batch HERE, not after the formula / OCR / table stages have run for nothing
        if self.seal_enable and not self._can_read_seals() and any(d.get("original_label") == "seal" for page in dets for d in page):
            raise RuntimeError(self._NO_SEAL_MODEL)
        # 2. region collection: writes the integer `bbox` into the formula detections before anything else touches them, so that the
        #    fields of a detection also appear in the reference's ORDER (bbox, then latex) when the result is serialised
        checkbox_res: Optional[List[List[dict]]] = None
        for p, page in enumerate(dets):
            layout_host.split_regions(page)
            attach_table_images(pages, p, page)
            if self.checkbox_enable:          # after the regions were collected: a checkbox is never an OCR region itself
                if checkbox_res is None:
                    checkbox_res = []
                found = list(self.checkbox_fn(np.ascontiguousarray(pages[p].cpu().numpy()[:, :, ::-1])))
                checkbox_res.append(found)
                for res in found:
                    b = res["bbox"]
                    page.append({"bbox": b, "poly": [b[0], b[1], b[2], b[1], b[2], b[3], b[0], b[3]],
                                 "category_id": layout_host.CHECKBOX_CATEGORY_ID, "checkbox": res["text"], "score": 0.9})
        # 3. formulas
        if self.formula_model is not None:
            recognise_formulas(pages, dets, self.formula_model, self.formula_expand_px, self.formula_batch_size)
        # 4. OCR
        if use_custom:
            # `_run_custom_ocr` (batch_analyze.py:286-333): every text region of the page BATCH in one call, BGR crops without a
            # margin (crop_img: white outside the page and outside the region's polygon), one string per region
            out = [list(d) for d in dets]
            crops, owners = [], []
            for p in range(P):
                regions, _t, _f = layout_host.split_regions(dets[p])
                for r in regions:
                    crop = crop_region(pages, p, r)
                    if crop is None:
                        continue
                    crops.append(np.ascontiguousarray(crop[:, :, ::-1]))          # BGR (:300)
                    owners.append((p, r))
            texts = self.custom_ocr.batch_predict(crops, batch_size=self.ocr.det_batch_num or 1) if crops else []
            for (p, r), text in zip(owners, texts):
                out[p].append({"poly": r["poly"], "category_id": OCR_TEXT, "score": 0.95, "text": text.strip() if text else "",
                               "vl_ocr": True, "original_label": r.get("original_label"), "original_order": r.get("original_order"),
                               "polygon_points": r.get("polygon_points")})
        else:
            out = self.ocr(pages, dets, det_maps_fn=det_maps_fn, page_langs=page_langs, mask_boxes_per_page=checkbox_res,
                           page_keys=page_keys, all_langs=all_langs)
        # 5. tables: one pooled `batch_predict` of a CustomBaseModel-shaped model (seam S1, batch_analyze.py:359-379) or, for a
        #    `predict`-shaped one (RapidTableModel, seam S3), the reference's own table stage with the table OCR on the GPU
        if self.table_model is not None and hasattr(self.table_model, "batch_predict"):
            crops, owners, fills = [], [], []
            for p in range(P):
                _o, tables, _f = layout_host.split_regions(dets[p])
                for t in tables:
                    rect = table_crop_rect(t)                  # snapped outwards to multiples of 5 px (batch_analyze.py:235-243)
                    if rect is not None:
                        crop = pages[p, rect[1]:rect[3], rect[0]:rect[2]].cpu().numpy()
                        useful = [0, 0, rect[0], rect[1], rect[2], rect[3], crop.shape[1], crop.shape[0]]
                        fills.append(layout_host.table_fill_images(t, useful) if self.table_image_enable else [])
                        crops.append(crop)
                        owners.append(t)
            if crops:
                for t, html in zip(owners, self.table_model.batch_predict(crops, fill_image_res_list=fills)):
                    t.pop("layout_image_list", None)
                    if html:
                        t["html"] = html
        elif self.table_model is not None:
            self.table_ocr(pages, dets, self.table_model, page_scales, det_maps_fn=table_det_maps_fn, table_image_enable=self.table_image_enable,
                           page_langs=page_langs, mask_boxes_per_page=checkbox_res)
        if self.seal_enable:
            self._run_seal_ocr(pages, out)
        # back to the coordinates of the pages as they came in (restore_poly, utils/boxbase.py:328-363; batch_analyze.py:153-161:
        # every detection gets `rotate_label`, only 90 / 270 pages have their `poly` mapped - a 180 page keeps its coordinates)
        for p, (h, w, label) in enumerate(orig_hw_label):
            for d in out[p]:
                d["rotate_label"] = label
                if label in ("90", "270"):
                    d["poly"] = restore_poly(d["poly"], label, w, h)
        return out

    _NO_SEAL_MODEL = ("the page holds a seal region and seal OCR is enabled (the reference's default), but no seal_model was "
                      "given: the seal detector / recogniser are ONNX-only and not part of this build - pass a "
                      "RapidOcrModel-shaped `seal_model` or seal_enable=False")

    def _can_read_seals(self) -> bool:
        import inspect
        return self.seal_model is not None or (self.custom_ocr is not None and
                                               "is_seal" in inspect.signature(self.custom_ocr.batch_predict).parameters)

    def _run_seal_ocr(self, pages: torch.Tensor, out: List[List[dict]]) -> int:
        """`BatchAnalyze._run_seal_ocr` (batch_analyze.py:415-470): every region the layout model labelled `seal` is cropped (crop_img, no
        margin, polygon whited out), handed over as BGR and gets `text` = the LIST of the lines read in it - from the custom OCR model when
        its batch_predict takes `is_seal` (one string, split at newlines), else from the seal OCR model (malformed / empty items skipped one
        by one, no score threshold; a crop nothing was read in keeps no `text`).  Returns the number of seal regions."""
        import inspect
        items = []
        for p, page in enumerate(out):
            for d in page:
                if d.get("original_label") == "seal":
                    crop = crop_region(pages, p, d)
                    if crop is None:
                        raise ValueError("seal region with an inverted box")          # (the reference's np.ones raises)
                    items.append((np.ascontiguousarray(crop[:, :, ::-1]), d))
        custom = self.custom_ocr is not None and "is_seal" in inspect.signature(self.custom_ocr.batch_predict).parameters
        for crop_bgr, d in items:
            if custom:
                texts = self.custom_ocr.batch_predict([crop_bgr], is_seal=True)[0].split("\n")
            else:
                if self.seal_model is None:
                    raise RuntimeError(self._NO_SEAL_MODEL)
                res = self.seal_model.ocr(crop_bgr, det=True, rec=True)[0]
                if not res:
                    continue
                texts = []
                for item in res:
                    if not item or len(item) != 2:
                        continue
                    rec = item[1]
                    if not rec or len(rec) < 1:
                        continue
                    if rec[0]:
                        texts.append(rec[0])
            d["text"] = texts
        return len(items)
