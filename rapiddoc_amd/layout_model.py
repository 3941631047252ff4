"""The layout wrapper around a pluggable detector session - host mirror of the three reference layers

    RapidLayoutModel.batch_predict      rapid_doc/model/layout/rapid_layout.py:55-108 (label -> CategoryId, `poly`, score rounding,
                                        inline-formula re-labelling :110-122)
    RapidLayout.__call__                rapid_doc/model/layout/rapid_layout_self/main.py:41-56 (chunks of `batch_size` pages,
                                        default thresholds per model :17-29)
    PPDocLayoutModelHandler.__call__    .../model_handler/pp_doclayout/main.py:14-80 (per-model input size, PPPreProcess,
                                        scale_factor, session call, `[boxes, box_nums(, masks)]` split :88-139, PPPostProcess)

with the pre-process on the GPU (`rd_preproc_resize_norm`, bicubic like cv2.resize(interpolation=2)) and the post-process in
C++ (`rd_layout_postprocess`).  The detector itself enters through the reference's own session contract (seam S2,
rapid_layout_self/inference_engine/base.py:15-57):

    session(image f32[B,3,S,S], scale_factor f32[B,2]) -> [boxes f32[sum n, 6|7|8], box_nums i32[B](, masks)]
    session.characters -> list of label strings

The reference ships that graph only as ONNX (neck / RT-DETR decoder / top-k are not readable source), so this file
is the plumbing of BASELINE.json `configs[0]`: any object honouring the contract plugs in - `SyntheticBoxSession` below for
tests and benchmarks, an onnxruntime session of the real model where that package and file exist.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import layout_host
from .engine import preproc_resize_norm

_TABLES = {k: {int(c): v for c, v in t.items()}
           for k, t in json.loads((Path(__file__).resolve().parent / "data" / "layout_model_tables.json").read_text()).items()}

IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)

# model_type -> (input size S, (mean, std), category family, ordered output, merge-mode table, unclip ratio, default threshold)
# pp_doclayout/main.py:17-29 (sizes, merge modes, unclip), pre_process.py:14-19 (mean / std), rapid_layout_self/main.py:19-29
# and rapid_layout.py:29-34 (default thresholds; S is lowered to 0.2 by the outer wrapper)
_MODELS: Dict[str, dict] = {
    "pp_doclayout_plus_l": dict(S=800, norm=((0, 0, 0), (1, 1, 1)), family="pp_doclayout_plus", ordered=False,
                                merge=_TABLES["PP_DOCLAYOUT_PLUS_L_layout_merge_bboxes_mode"], unclip=[1.0, 1.0],
                                thresh=_TABLES["PP_DOCLAYOUT_PLUS_L_Threshold"]),
    "pp_doclayoutv2": dict(S=800, norm=((0, 0, 0), (1, 1, 1)), family="pp_doclayoutv2", ordered=True,
                           merge=_TABLES["PP_DOCLAYOUTV2_layout_merge_bboxes_mode"], unclip=[1.0, 1.0],
                           thresh=_TABLES["PP_DOCLAYOUTV2_Threshold"]),
    "pp_doclayoutv3": dict(S=800, norm=((0, 0, 0), (1, 1, 1)), family="pp_doclayoutv2", ordered=True,
                           merge=_TABLES["PP_DOCLAYOUTV2_layout_merge_bboxes_mode"], unclip=[1.0, 1.0], thresh=0.3),
    "pp_doclayout_l": dict(S=640, norm=((0, 0, 0), (1, 1, 1)), family="pp_doclayout", ordered=False, merge=None, unclip=None,
                           thresh=_TABLES["PP_DOCLAYOUT_L_Threshold"]),
    "pp_doclayout_m": dict(S=640, norm=(IMAGENET_MEAN, IMAGENET_STD), family="pp_doclayout", ordered=False, merge=None,
                           unclip=None, thresh=0.5),
    "pp_doclayout_s": dict(S=480, norm=(IMAGENET_MEAN, IMAGENET_STD), family="pp_doclayout", ordered=False, merge=None,
                           unclip=None, thresh=0.2),
    # DocLayout-YOLO (model_handler/doc_layout/: letterbox to 1024 x 1024, one page per session call, the graph's own NMS);
    # threshold 0.2 is RapidLayoutModel's default for it (rapid_layout.py:33-35)
    "doclayout_docstructbench": dict(S=1024, yolo=True, family="doclayout_yolo", ordered=False, thresh=0.2),
}
# label -> category id of the DocLayout-YOLO classes: the position in this list, 'isolate_formula' -> 14 (rapid_layout.py:48-50,71-75)
_YOLO_LABELS = ["title", "plain text", "abandon", "figure", "figure_caption", "table", "table_caption", "table_footnote",
                "isolate_formula", "formula_caption", "10", "11", "12", "inline_formula", "isolated_formula", "ocr_text"]
DEFAULT_IGNORE = ("number", "footnote", "header", "header_image", "footer", "footer_image", "aside_text")   # rapid_layout.py:42-43


def _iou(a, b) -> float:
    """rapid_doc/utils/boxbase.py:139-172."""
    xl, yt, xr, yb = max(a[0], b[0]), max(a[1], b[1]), min(a[2], b[2]), min(a[3], b[3])
    if xr < xl or yb < yt:
        return 0.0
    inter = (xr - xl) * (yb - yt)
    a1, a2 = (a[2] - a[0]) * (a[3] - a[1]), (b[2] - b[0]) * (b[3] - b[1])
    if a1 == 0 or a2 == 0:
        return 0
    return inter / float(a1 + a2 - inter)


def split_session_output(pred: Sequence[np.ndarray]) -> List[dict]:
    """`_format_output` for detector outputs (pp_doclayout/main.py:88-139): [boxes, box_nums] or [boxes, box_nums, masks]
    -> one {"boxes"(, "masks")} per image."""
    boxes, nums = np.asarray(pred[0]), np.asarray(pred[1]).reshape(-1)
    out, start = [], 0
    for n in nums.tolist():
        d = {"boxes": np.array(boxes[start:start + n])}
        if len(pred) == 3:
            d["masks"] = np.array(np.asarray(pred[2])[start:start + n])
        out.append(d)
        start += n
    return out


class LayoutModel:
    """`batch_predict(images, batch_size) -> list[list[dict]]` exactly as `RapidLayoutModel.batch_predict` returns it."""

    def __init__(self, session, model_type: str = "pp_doclayoutv3", conf_thresh: Union[None, float, Dict[int, float]] = None,
                 iou_thresh: float = 0.5, markdown_ignore_labels: Sequence[str] = DEFAULT_IGNORE, device: int = 0,
                 layout_shape_mode: str = "auto"):
        """`layout_shape_mode` (typings.py:169, default "auto"): what becomes of the detector's instance masks when its session
        returns them - "rect" ignores them, "poly" / "quad" / "auto" turn them into `polygon_points` (layout_polygon.py)."""
        if model_type not in _MODELS:
            raise ValueError(f"model_type must be one of {sorted(_MODELS)}")
        if layout_shape_mode not in ("rect", "poly", "quad", "auto"):
            raise ValueError("layout_shape_mode must be one of ['rect', 'poly', 'quad', 'auto']")
        self.layout_shape_mode = layout_shape_mode
        self.session, self.model_type, self.m = session, model_type, _MODELS[model_type]
        self.device = torch.device("cuda", device)
        self.labels = list(session.characters)
        self.ignore = tuple(markdown_ignore_labels)
        thr = conf_thresh if conf_thresh else self.m["thresh"]          # `if not cfg.conf_thresh` (rapid_layout_self/main.py:19)
        self.conf_thresh = thr
        if self.m.get("yolo"):
            self.post = None
            return
        self.post = layout_host.LayoutPostProcess(self.labels, thr, iou_thresh, layout_merge_bboxes_mode=self.m["merge"],
                                                  layout_unclip_ratio=self.m["unclip"], scale_size=(self.m["S"], self.m["S"]))

    # ------------------------------------------------------------------ PPPreProcess on the GPU (pre_process.py:22-42)
    def preprocess(self, pages: Sequence[Union[np.ndarray, torch.Tensor]]) -> Tuple[torch.Tensor, np.ndarray]:
        """u8 HWC pages (host arrays or device tensors, RGB as the pipeline hands them over, load_image.py:76-79) ->
        f32 [B,3,S,S] on the device + scale_factor f32[B,2] = [S / H, S / W] (main.py:46-52)."""
        S = self.m["S"]
        mean, std = self.m["norm"]
        x = torch.empty((len(pages), 3, S, S), dtype=torch.float32, device=self.device)
        sf = np.empty((len(pages), 2), np.float32)
        for i, pg in enumerate(pages):
            t = pg if isinstance(pg, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(pg, dtype=np.uint8))
            if t.device.type != "cuda":
                t = t.to(self.device, non_blocking=True)
            preproc_resize_norm(t.contiguous(), (S, S), mean=mean, std=std, interp=2, out=x[i])
            sf[i] = (S / t.shape[0], S / t.shape[1])
        return x, sf

    # ------------------------------------------------------------------ DocLayout-YOLO: letterbox, one page per call
    @staticmethod
    def letterbox_geometry(h: int, w: int, S: int = 1024) -> Tuple[int, int, int, int]:
        """(new_w, new_h, left, top) of LetterBox(new_shape=(S, S), auto=False, center=True) (doc_layout/utils.py:29-67)."""
        r = min(S / h, S / w)
        new_w, new_h = int(round(w * r)), int(round(h * r))
        dw, dh = (S - new_w) / 2, (S - new_h) / 2
        return new_w, new_h, int(round(dw - 0.1)), int(round(dh - 0.1))

    def _resize_linear_u8(self, page: torch.Tensor, new_h: int, new_w: int) -> torch.Tensor:
        """cv2.resize(img, (new_w, new_h), INTER_LINEAR) as float32 [3, new_h, new_w] holding the 8-bit results (GPU kernel)."""
        return preproc_resize_norm(page.contiguous(), (new_h, new_w), scale=1.0, interp=1)

    def preprocess_letterbox(self, page: Union[np.ndarray, torch.Tensor]) -> torch.Tensor:
        """DocLayoutPreProcess (doc_layout/pre_process.py:13-26): letterbox with 114, channel order reversed, / 255 -> float32
        [1, 3, S, S].  The division is the reference's float64 one (uint8 / 255, then float32), done as such on the result of the
        8-bit resize."""
        S = self.m["S"]
        t = page if isinstance(page, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(page, dtype=np.uint8))
        if t.device.type != "cuda" and torch.cuda.is_available():      # (without a GPU the resize below refuses a host tensor)
            t = t.to(self.device, non_blocking=True)
        h, w = int(t.shape[0]), int(t.shape[1])
        new_w, new_h, left, top = self.letterbox_geometry(h, w, S)
        canvas = torch.full((3, S, S), 114.0, dtype=torch.float32, device=t.device)
        if (new_w, new_h) != (w, h):
            canvas[:, top:top + new_h, left:left + new_w] = self._resize_linear_u8(t, new_h, new_w)
        else:
            canvas[:, top:top + new_h, left:left + new_w] = t.permute(2, 0, 1).to(torch.float32)
        return (canvas.flip(0).to(torch.float64) / 255).to(torch.float32).unsqueeze(0).contiguous()

    def _yolo_page(self, page) -> List[dict]:
        """DocLayoutModelHandler.__call__ for one page + DocLayoutPostProcess (doc_layout/main.py:50-66, post_process.py:17-32,
        utils.py:83-130): rows [x1, y1, x2, y2, conf, cls] of the letterboxed input -> page pixels (minus the padding, divided by the
        gain, clipped), labels, then RapidLayoutModel's category ids."""
        S = self.m["S"]
        h, w = int(page.shape[0]), int(page.shape[1])
        x = self.preprocess_letterbox(page)
        wants_device = getattr(self.session, "accepts_device_tensors", False)
        preds = np.asarray(self.session(x if wants_device else x.cpu().numpy())[0])
        rows = np.array(preds[0][preds[0][..., 4] > self.conf_thresh])
        gain = min(S / h, S / w)
        pad = (round((S - w * gain) / 2 - 0.1), round((S - h * gain) / 2 - 0.1))
        rows[..., 0] -= pad[0]
        rows[..., 1] -= pad[1]
        rows[..., 2] -= pad[0]
        rows[..., 3] -= pad[1]
        rows[..., :4] /= gain
        rows[..., [0, 2]] = rows[..., [0, 2]].clip(0, w)
        rows[..., [1, 3]] = rows[..., [1, 3]].clip(0, h)
        dets = []
        for r in rows:
            label = self.labels[int(r[5])]
            x0, y0, x1, y1 = r[0], r[1], r[2], r[3]                     # numpy float32 scalars, like the reference's dicts
            dets.append({"category_id": 14 if label == "isolate_formula" else _YOLO_LABELS.index(label), "original_label": label,
                         "original_order": -1, "poly": [x0, y0, x1, y0, x1, y1, x0, y1], "polygon_points": None,
                         "score": round(float(r[4]), 3)})
        return self._check_inline_formula(dets)

    # ------------------------------------------------------------------ handler: one chunk of pages
    def _chunk(self, pages) -> List[List[dict]]:
        if self.m.get("yolo"):
            return [self._yolo_page(pg) for pg in pages]
        x, sf = self.preprocess(pages)
        wants_device = getattr(self.session, "accepts_device_tensors", False)
        pred = self.session(x if wants_device else x.cpu().numpy(), sf)
        outs = split_session_output(pred)
        res = []
        mode = self.layout_shape_mode
        for pg, o in zip(pages, outs):
            h, w = int(pg.shape[0]), int(pg.shape[1])
            if "masks" not in o:
                mode = "rect"                              # stays "rect" for the rest of the chunk, like the reference (main.py:59-66)
            datas = self.post(o["boxes"], [w, h], o.get("masks"), mode)
            datas = [] if isinstance(datas, np.ndarray) else datas
            dets = layout_host.to_layout_dets(datas, self.m["family"], self.m["ordered"], self.ignore)
            if self.m["family"] != "pp_doclayoutv2":
                dets = self._check_inline_formula(dets)
            res.append(dets)
        return res

    @staticmethod
    def _check_inline_formula(dets: List[dict]) -> List[dict]:
        """rapid_layout.py:110-122: an isolated formula whose box nearly coincides (IoU >= 0.9) with a text box is inline."""
        C = layout_host.CATEGORY_ID
        for d in dets:
            if d["category_id"] == C["InterlineEquation_YOLO"]:
                box = (d["poly"][0], d["poly"][1], d["poly"][4], d["poly"][5])
                for o in dets:
                    if o["category_id"] == C["Text"] and _iou(box, (o["poly"][0], o["poly"][1], o["poly"][4], o["poly"][5])) >= 0.9:
                        d["category_id"] = C["InlineEquation"]
                        break
        return dets

    def batch_predict(self, images: Sequence[Union[np.ndarray, torch.Tensor]], batch_size: int = 1) -> List[List[dict]]:
        out: List[List[dict]] = []
        bs = max(1, int(batch_size))
        for i in range(0, len(images), bs):               # rapid_layout_self/main.py:49-54
            out += self._chunk(images[i:i + bs])
        return out

    def predict(self, image) -> List[dict]:
        return self.batch_predict([image], 1)[0]


class SyntheticYoloSession:
    """Stand-in for a DocLayout-YOLO session: `session(image f32[1,3,S,S]) -> [rows f32[1,N,6]]`, rows = [x1, y1, x2, y2, conf, cls]
    in letterboxed-input pixels, deterministic in (seed, call index).  Some boxes reach into the padding (clipping)."""

    def __init__(self, labels: Sequence[str], n: int = 40, seed: int = 0, size: int = 1024):
        self.characters = list(labels)
        self.n, self.seed, self.size = n, seed, size
        self.calls: List[Tuple[Tuple[int, ...], int]] = []

    def have_key(self, key: str = "character") -> bool:
        return True

    def __call__(self, image, scale_factor=None):
        import zlib
        image = np.asarray(image)
        assert image.shape == (1, 3, self.size, self.size) and image.dtype == np.float32 and scale_factor is None
        self.calls.append((tuple(image.shape), zlib.crc32(np.ascontiguousarray(image).tobytes())))
        rng = np.random.default_rng([self.seed, len(self.calls)])
        x0, y0 = rng.uniform(-20, 0.8 * self.size, self.n), rng.uniform(-20, 0.9 * self.size, self.n)
        bw, bh = rng.uniform(20, 0.5 * self.size, self.n), rng.uniform(10, 0.2 * self.size, self.n)
        rows = np.stack([x0, y0, x0 + bw, y0 + bh, rng.uniform(0.05, 0.99, self.n),
                         rng.integers(0, len(self.characters), self.n).astype(np.float64)], axis=1).astype(np.float32)
        return [rows[None]]


class SyntheticBoxSession:
    """A detector stand-in honouring the session contract: deterministic boxes, a pure function of the page index inside the
    call and of the scale factors (so the post-process sees page-pixel boxes like the real graph's, which rescales by
    1 / scale_factor in-graph, onnxruntime/main.py:61-78).  Plumbing tests / BASELINE `configs[0]` only."""

    def __init__(self, labels: Sequence[str], boxes_per_page: int = 40, ncol: int = 6, seed: int = 0, size: int = 800,
                 twins: Optional[Tuple[str, str, int]] = None, masks: bool = False):
        """`twins` = (label_a, label_b, k): the first k boxes of a page carry label_a and each gets a near-coincident copy
        (2 px smaller all round, IoU > 0.9) labelled label_b - the situation `check_inline_formula` exists for.
        `masks`: a third output like the instance-segmentation detectors give, u8 [sum n, size / 4, size / 4]: per box an ellipse,
        a slanted quadrilateral or the full rectangle drawn in its patch of the mask grid."""
        self.characters = list(labels)
        self.n, self.ncol, self.seed, self.size, self.twins, self.masks = boxes_per_page, ncol, seed, size, twins, masks
        self.calls: List[Tuple[Tuple[int, ...], np.ndarray]] = []

    def have_key(self, key: str = "character") -> bool:
        return True

    def __call__(self, image: np.ndarray, scale_factor: Optional[np.ndarray] = None):
        image = np.asarray(image)
        assert image.ndim == 4 and image.shape[1] == 3 and image.dtype == np.float32
        B = image.shape[0]
        assert scale_factor is not None and scale_factor.shape == (B, 2) and scale_factor.dtype == np.float32
        self.calls.append((tuple(image.shape), scale_factor.copy()))
        rows = []
        for b in range(B):
            rng = np.random.default_rng([self.seed, b, int(image[b].sum() * 0) + B])
            H, W = self.size / scale_factor[b, 0], self.size / scale_factor[b, 1]
            x0 = rng.uniform(0, 0.8 * W, self.n)
            y0 = rng.uniform(0, 0.9 * H, self.n)
            bw = rng.uniform(20, 0.5 * W, self.n)
            bh = rng.uniform(10, 0.15 * H, self.n)
            cols = [rng.integers(0, len(self.characters), self.n).astype(np.float32), rng.uniform(0.05, 0.99, self.n),
                    x0, y0, np.minimum(x0 + bw, W), np.minimum(y0 + bh, H)]
            if self.ncol >= 7:
                cols.append(rng.permutation(self.n).astype(np.float32))      # reading order column of V2 / V3
            if self.ncol == 8:
                cols.append(np.zeros(self.n))
            page = np.stack(cols, axis=1).astype(np.float32)
            if self.twins:
                la, lb, k = self.twins
                page[:k, 0], page[:k, 1] = self.characters.index(la), 0.9 - 0.01 * np.arange(k)     # (distinct: see layout_postprocess.cpp on ties)
                page[:k, 4], page[:k, 5] = np.maximum(page[:k, 4], page[:k, 2] + 200), np.maximum(page[:k, 5], page[:k, 3] + 100)
                twin = page[:k].copy()
                twin[:, 0], twin[:, 1] = self.characters.index(lb), 0.8 - 0.01 * np.arange(k)
                twin[:, 2:6] += np.float32([2, 2, -2, -2])
                page = np.concatenate([page, twin], axis=0)
            rows.append(page)
        out = [np.concatenate(rows, axis=0), np.asarray([len(r) for r in rows], np.int32)]
        if self.masks:
            g = self.size // 4
            grids = []
            for b, page in enumerate(rows):
                sy, sx = scale_factor[b, 0] / 4, scale_factor[b, 1] / 4          # page pixels -> mask grid
                m = np.zeros((len(page), g, g), np.uint8)
                for i, r in enumerate(page):
                    x0, x1 = int(np.floor(r[2] * sx)), min(g, max(int(np.ceil(r[4] * sx)), int(np.floor(r[2] * sx)) + 1))
                    y0, y1 = int(np.floor(r[3] * sy)), min(g, max(int(np.ceil(r[5] * sy)), int(np.floor(r[3] * sy)) + 1))
                    yy, xx = np.mgrid[:max(y1 - y0, 0), :max(x1 - x0, 0)]
                    u, v = (xx + 0.5) / max(x1 - x0, 1), (yy + 0.5) / max(y1 - y0, 1)
                    if i % 3 == 0:
                        shape = ((u - 0.5) / 0.45) ** 2 + ((v - 0.5) / 0.42) ** 2 <= 1
                    elif i % 3 == 1:
                        shape = (u - 0.2 * (v - 0.5) > 0.12) & (u - 0.2 * (v - 0.5) < 0.88) & (v > 0.1) & (v < 0.9)
                    else:
                        shape = np.ones(u.shape, bool)
                    m[i, y0:y1, x0:x1] = shape
                grids.append(m)
            out.append(np.concatenate(grids, axis=0))
        return out
