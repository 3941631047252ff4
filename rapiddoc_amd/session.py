"""Drop-in `InferSession` shims for RapidDoc's engine seam (seam S2).

The reference's OCR sessions are callables `session(np.ndarray NCHW float32) -> np.ndarray`
(rapid_doc/model/ocr/torch.py:171-192) plus `have_key()` / `get_character_list()` (:194-198); rapidocr's
TextDetector / TextRecognizer only ever touch those three members.  These classes provide exactly that surface on
top of the C-ABI, so `INTEGRATION.md`'s two-line patch makes the unchanged pipeline run on the MI355X engine.

`Mi355RecSession.__call__` returns the reference-shaped softmax tensor [B,T,C] (strict drop-in);
`infer_indices` is the fast path that returns only (argmax idx, max prob) per time step - what CTCLabelDecode uses.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from .engine import REC_WANT_SOFTMAX, RdEngine

WeightSrc = Union[str, bytes, Dict[str, np.ndarray]]


class _BaseSession:
    kind = ""

    def __init__(self, weights: WeightSrc, device: int = 0):
        self.engine = RdEngine(self.kind, device).load_weights(weights)
        self.device = torch.device("cuda", device)

    @classmethod
    def from_cfg(cls, cfg) -> "_BaseSession":
        """Build from a rapidocr-style cfg mapping: `model_path` (.safetensors) and `engine_cfg.gpu_id`."""
        model_path = cfg.get("model_path") if hasattr(cfg, "get") else cfg["model_path"]
        eng_cfg = cfg.get("engine_cfg", {}) if hasattr(cfg, "get") else {}
        gpu_id = int(getattr(eng_cfg, "gpu_id", None) or (eng_cfg.get("gpu_id", 0) if hasattr(eng_cfg, "get") else 0))
        if Path(str(model_path)).suffix != ".safetensors":
            raise ValueError("the MI355X engine loads the reference's .safetensors weights (torch.py:93-103)")
        return cls(str(model_path), gpu_id)

    def _to_dev(self, img: np.ndarray) -> torch.Tensor:
        x = torch.from_numpy(np.ascontiguousarray(img, dtype=np.float32))
        return x.to(self.device, non_blocking=True)

    # `copy_out = False`: __call__ returns a VIEW of one of the session's two pinned staging buffers instead of a fresh array - valid
    # until the session has been called twice more.  rapidocr's callers consume a result (DB post-process, CTC argmax) before they
    # call the session again, so the view is safe there and saves a host memcpy of the whole softmax tensor per call; the default
    # keeps the reference's contract (a fresh array per call).
    copy_out = True

    def _to_host(self, t: torch.Tensor) -> np.ndarray:
        """Device tensor -> numpy through pinned staging buffers of the session (two, used in turn, grown on demand): the copy off the
        device runs at the link's rate instead of the pageable-memory path of `.cpu()` (the rec session hands back tens of MB per
        call: softmax [6, T, 18710])."""
        t = t.contiguous()
        n = t.numel()
        if not hasattr(self, "_pins"):
            self._pins, self._pin_i = [None, None], 0
        k = self._pin_i = self._pin_i ^ 1
        if self._pins[k] is None or self._pins[k].numel() < n or self._pins[k].dtype != t.dtype:
            self._pins[k] = torch.empty(int(n * 1.25) + 1024, dtype=t.dtype, pin_memory=True)
        stage = self._pins[k][:n].view(t.shape)
        stage.copy_(t, non_blocking=True)
        torch.cuda.current_stream(self.device).synchronize()
        return stage.numpy().copy() if self.copy_out else stage.numpy()

    # rapidocr InferSession protocol (ocr/torch.py:194-198)
    def have_key(self, key: str = "character") -> bool:
        return False

    def get_character_list(self, key: str = "character") -> List[str]:
        return []


# The split-fp16 range guard (fp32 re-run) is part of every RdEngine forward (engine.RdEngine._guarded, guard="sync").


class Mi355DetSession(_BaseSession):
    """PP-OCRv6 det: [B,3,H,W] -> DB probability map [B,1,H,W] (`maps`, ocr/torch.py:183-184)."""
    kind = "ppocrv6_det"

    def __call__(self, img: np.ndarray) -> np.ndarray:
        x = self._to_dev(img)
        return self._to_host(self.engine.det_forward(x))


class Mi355RecSession(_BaseSession):
    """PP-OCRv6 rec: [B,3,48,W] -> softmax(ctc_logits) [B,T,C] (ocr/torch.py:185-187)."""
    kind = "ppocrv6_rec"

    def __call__(self, img: np.ndarray) -> np.ndarray:
        x = self._to_dev(img)
        return self._to_host(self.engine.rec_forward(x, REC_WANT_SOFTMAX)[2])

    def infer_indices(self, img: Union[np.ndarray, torch.Tensor]) -> Tuple[np.ndarray, np.ndarray]:
        x = img if isinstance(img, torch.Tensor) else self._to_dev(img)
        idx, prob, _ = self.engine.rec_forward(x)
        return idx.cpu().numpy(), prob.cpu().numpy()


class Mi355LayoutBackboneSession(_BaseSession):
    """PPHGNetV2-B4 backbone of PP-DocLayout-L/plus-L/V2/V3: [B,3,S,S] -> 4 NCHW feature maps."""
    kind = "pphgnetv2_b4"

    def __call__(self, img: np.ndarray) -> List[np.ndarray]:
        return [f.cpu().numpy() for f in self.engine.backbone_forward(self._to_dev(img))]


def install_into_rapidocr() -> None:
    """Replace rapidocr's torch engine by the MI355X sessions at the places the reference patches it
    (rapid_doc/model/ocr/ocr_patch.py:95-105): rapidocr >= 3.4.3 keeps `TorchInferSession` in `rapidocr.inference_engine.pytorch.main`
    and re-exports it from the package - both names are replaced, like the reference does; older releases of the pinned range
    (>= 3.4.0) keep it in `rapidocr.inference_engine.torch`.  Needs the `rapidocr` package of the RapidDoc installation."""
    import importlib

    class _Dispatch:
        def __new__(cls, cfg):
            # rapidocr's cfg carries `task_type`; the reference's session itself goes by the weight file's stem, the key of
            # arch_config.yaml ("ch_PP-OCRv6_det_small", rapid_doc/model/ocr/torch.py:70-77): used when task_type is absent
            get = (lambda k: getattr(cfg, k, None)) if not hasattr(cfg, "get") else (lambda k: cfg.get(k, None) or getattr(cfg, k, None))
            task = str(get("task_type") or Path(str(get("model_path") or "")).stem).lower()
            return (Mi355DetSession if "det" in task else Mi355RecSession).from_cfg(cfg)

    try:
        main = importlib.import_module("rapidocr.inference_engine.pytorch.main")
        pkg = importlib.import_module("rapidocr.inference_engine.pytorch")
    except ModuleNotFoundError as e:
        if not str(e.name or "").startswith("rapidocr.inference_engine.pytorch"):
            raise
        main, pkg = importlib.import_module("rapidocr.inference_engine.torch"), None
    main.TorchInferSession = _Dispatch
    if pkg is not None:
        pkg.TorchInferSession = _Dispatch
