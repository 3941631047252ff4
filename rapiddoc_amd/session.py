"""Drop-in `InferSession` shims for RapidDoc's engine seam (seam S2).

The reference's OCR sessions are callables `session(np.ndarray NCHW float32) -> np.ndarray`
(rapid_doc/model/ocr/torch.py:171-192) plus `have_key()` / `get_character_list()` (:194-198); rapidocr's
TextDetector / TextRecognizer only ever touch those three members.  These classes provide exactly that surface on
top of the C-ABI, so `INTEGRATION.md`'s two-line patch makes the unchanged pipeline run on the MI355X engine.

`Mi355RecSession.__call__` returns the reference-shaped softmax tensor [B,T,C] - by default as a `LazySoftmax`: the tensor stays in HBM
and only its per-time-step (argmax, max) cross PCIe, because those two reductions are all rapidocr's CTCLabelDecode asks of it
(rapid_ocr.py:443-449); any other access materialises the exact array.  `lazy_softmax=False` hands out plain ndarrays (every call then
moves 18-60 MB off the device); `infer_indices` is the fast path that returns only (argmax idx, max prob) per time step.
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
        # the session's own stream: every call runs upload -> forward -> download on it and waits for it before it returns (the seam is
        # synchronous: numpy in, numpy out).  Not the legacy default stream - that one cannot be captured, and a forward whose buffers
        # keep their addresses is replayed by the library as one hipGraph launch from its third appearance on.
        self.stream = torch.cuda.Stream(self.device)

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
        """Host array -> the session's own device input buffer (one allocation, grown on demand: the SAME base address call after call,
        which - together with the session's result buffers - lets the library replay a forward of a shape it has seen as one hipGraph
        launch instead of 80-90 kernel launches), through a pinned staging buffer: the array is copied once by the host's cores into
        pinned memory and travels from there at the link's rate; the engine reads it on the current stream behind the copy."""
        a = np.ascontiguousarray(img, dtype=np.float32)
        n = a.size
        if getattr(self, "_in_pin", None) is None or self._in_pin.numel() < n:
            self._in_pin = torch.empty(int(n * 1.25) + 1024, dtype=torch.float32, pin_memory=True)
            self._in_dev = torch.empty(self._in_pin.numel(), dtype=torch.float32, device=self.device)
        stage = self._in_pin[:n].view(a.shape)
        # (numpy's single-threaded memcpy, not torch's copy_: on a many-core host torch splits a 5-MB copy over its whole intra-op pool,
        #  and waking 128 sleeping threads per call measured 5-9 ms against 0.03 ms for the copy itself - profiles/r6_s2_calls.txt)
        np.copyto(stage.numpy(), a)
        x = self._in_dev[:n].view(a.shape)
        x.copy_(stage, non_blocking=True)
        return x

    def _dev_buffer(self, tag: str, shape: tuple, dtype) -> torch.Tensor:
        """A view of the session's persistent device buffer `tag` (grown on demand; stable base address)."""
        n = int(np.prod(shape))
        bufs = self.__dict__.setdefault("_dev_bufs", {})
        b = bufs.get(tag)
        if b is None or b.numel() < n or b.dtype != dtype:
            b = bufs[tag] = torch.empty(int(n * 1.25) + 1024, dtype=dtype, device=self.device)
        return b[:n].view(shape)

    # `copy_out = False`: __call__ returns a VIEW of one of the session's two pinned staging buffers instead of a fresh array - valid
    # until the session has been called twice more.  rapidocr's callers consume a result (DB post-process, CTC argmax) before they
    # call the session again, so the view is safe there and saves a host memcpy of the whole softmax tensor per call; the default
    # keeps the reference's contract (a fresh array per call).
    copy_out = True

    def _to_host(self, t: torch.Tensor, copy_out: Optional[bool] = None) -> np.ndarray:
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
        return stage.numpy().copy() if (self.copy_out if copy_out is None else copy_out) else stage.numpy()

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
        with torch.cuda.stream(self.stream):
            x = self._to_dev(img)
            out = self._dev_buffer("maps", (x.shape[0], 1, x.shape[2], x.shape[3]), torch.float32)
            return self._to_host(self.engine.det_forward(x, out=out))


class LazySoftmax(np.lib.mixins.NDArrayOperatorsMixin):
    """The recogniser's softmax tensor [B,T,C] float32, left in HBM until somebody looks at it.

    rapidocr's `CTCLabelDecode.__call__` reduces the tensor the session returns to `preds.argmax(axis=2)` and `preds.max(axis=2)` and
    never touches it again (the reference's loop: rapid_ocr.py:443-449).  The device computes exactly those two reductions from the
    values it wrote (csrc/kernels_misc.hip row_softmax_kernel: the row's largest written value, and the LOWEST class holding it - numpy's
    tie rule), so `argmax(axis=2)` / `max(axis=2)` answer from 12 bytes per time step instead of 75 KB.  Everything else - `np.asarray`,
    indexing, iteration, arithmetic, any numpy function, any other reduction or axis - first copies the tensor off the device (once,
    through the session's pinned staging buffers) and then behaves as that ndarray: same values, bit for bit, as a session built with
    `lazy_softmax=False` returns.

    It is an array-LIKE (`__array__`, `__array_ufunc__`, `__array_function__`), deliberately not an `np.ndarray` subclass: numpy's C
    code reads a subclass instance's buffer directly (`np.asarray(x)` on a subclass is a base-class view, no Python hook runs), so a
    subclass could not guarantee that what is read has been copied; through `__array__` nothing can see the tensor before it is there.
    `isinstance(x, np.ndarray)` is therefore False - consumers that need the real type call `np.asarray(x)` (or build the session
    with `lazy_softmax=False`)."""

    __array_priority__ = 1000.0

    def __init__(self, session: "_BaseSession", full_dev: torch.Tensor, idx, prob):
        """`full_dev`: the tensor in HBM - possibly a view of a buffer the session re-uses: the session materialises this object before it
        overwrites the buffer (Mi355RecSession._retire).  `idx` / `prob`: the device's argmax / max over the class axis, as host arrays
        (or tensors, copied)."""
        self._session, self._dev = session, full_dev
        self.shape, self.dtype, self.ndim = tuple(full_dev.shape), np.dtype(np.float32), full_dev.dim()
        idx = idx.cpu().numpy() if isinstance(idx, torch.Tensor) else np.asarray(idx)
        prob = prob.cpu().numpy() if isinstance(prob, torch.Tensor) else np.asarray(prob)
        self._idx = idx.astype(np.intp)                            # numpy's argmax dtype (a copy: the source may be a staging buffer)
        self._prob = prob.astype(np.float32, copy=True)
        self._host: Optional[np.ndarray] = None

    # ---- the two reductions the CTC decode performs
    def _last_axis(self, axis, out, kw) -> bool:
        return self._host is None and axis is not None and int(axis) in (self.ndim - 1, -1) and out is None and not kw

    def argmax(self, axis=None, out=None, **kw):
        if self._last_axis(axis, out, kw):
            return self._idx.copy()
        return self.materialize().argmax(axis=axis, out=out, **kw)

    def max(self, axis=None, out=None, **kw):
        if self._last_axis(axis, out, kw):
            return self._prob.copy()
        return self.materialize().max(axis=axis, out=out, **kw)

    # ---- everything else: the exact tensor
    @property
    def materialized(self) -> bool:
        return self._host is not None

    def materialize(self) -> np.ndarray:
        if self._host is None:
            self._host = self._session._to_host(self._dev, copy_out=True)      # (its own memory: the object may outlive the staging buffer)
            self._dev = None
            self._session.softmax_materialized += 1
        return self._host

    def __array__(self, dtype=None, copy=None):
        a = self.materialize()
        return a if dtype is None or np.dtype(dtype) == a.dtype else a.astype(dtype)

    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        inputs = tuple(x.materialize() if isinstance(x, LazySoftmax) else x for x in inputs)
        if "out" in kwargs:
            kwargs["out"] = tuple(x.materialize() if isinstance(x, LazySoftmax) else x for x in kwargs["out"])
        return getattr(ufunc, method)(*inputs, **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        if func in (np.argmax, np.max, np.amax) and args and args[0] is self:          # np.argmax(preds, axis=2) == preds.argmax(axis=2)
            return (self.argmax if func is np.argmax else self.max)(*args[1:], **kwargs)

        def conv(v):
            if isinstance(v, LazySoftmax):
                return v.materialize()
            if isinstance(v, (list, tuple)):
                return type(v)(conv(e) for e in v)
            return v
        return func(*conv(tuple(args)), **{k: conv(v) for k, v in kwargs.items()})

    def __getitem__(self, key):
        return self.materialize()[key]

    def __setitem__(self, key, value):
        self.materialize()[key] = value

    def __len__(self) -> int:
        return self.shape[0]

    def __iter__(self):
        return iter(self.materialize())

    def __getattr__(self, name):
        # any ndarray member not spelled out above (sum, mean, reshape, tobytes, T, flags, ...): the materialised array's
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.materialize(), name)

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    @property
    def nbytes(self) -> int:
        return self.size * 4

    def __repr__(self) -> str:
        return "LazySoftmax(shape=%s, %s)" % (self.shape, "materialized" if self.materialized else "on device")


class Mi355RecSession(_BaseSession):
    """PP-OCRv6 rec: [B,3,48,W] -> softmax(ctc_logits) [B,T,C] (ocr/torch.py:185-187).

    `lazy_softmax` (default True): the result is a `LazySoftmax` - an array-like that holds the tensor in HBM, answers the CTC decode's
    `argmax(axis=2)` / `max(axis=2)` from the device's own reductions of it, and turns into the exact ndarray on any other access.
    False: a plain ndarray per call (the round-1..5 behaviour; `copy_out` then picks between a fresh array and a pinned view)."""
    kind = "ppocrv6_rec"

    def __init__(self, weights: WeightSrc, device: int = 0, lazy_softmax: bool = True):
        super().__init__(weights, device)
        self.lazy_softmax = lazy_softmax
        self.softmax_materialized = 0          # LazySoftmax results that were copied off the device after all
        self._last_lazy = [None, None]         # weak references to the LazySoftmax living in either softmax buffer
        self._turn = 0
        self.host_ms = {"stage_in": 0.0, "forward_and_wait": 0.0, "calls": 0}      # host clock per phase of the lazy path, summed over calls

    def _retire(self, k: int) -> None:
        """Softmax buffer `k` is about to be overwritten: a LazySoftmax that still lives in it takes its copy first.  There are TWO
        buffers, used in turn, because the reference's loop holds call i's `preds` while it makes call i + 1 (`preds = session(...)`,
        rapid_ocr.py:443, rebinds the name only when the call returns): by call i + 2 the object is gone, so in the reference's use this
        never copies."""
        ref = self._last_lazy[k]
        prev = ref() if ref is not None else None
        if prev is not None and not prev.materialized:
            prev.materialize()
        self._last_lazy[k] = None

    def __call__(self, img: np.ndarray):
        with torch.cuda.stream(self.stream):
            return self._call(img)

    def _call(self, img: np.ndarray):
        import weakref
        import time
        t0 = time.perf_counter()
        k = self._turn = self._turn ^ 1
        self._retire(k)
        x = self._to_dev(img)
        t1 = time.perf_counter()
        B, T = x.shape[0], self.engine._l.rd_rec_seq_len(x.shape[3])
        idx = self._dev_buffer("idx", (B, T), torch.int32)
        prob = self._dev_buffer("prob", (B, T), torch.float32)
        full = self._dev_buffer("softmax%d" % k, (B, T, self.engine.num_classes), torch.float32)
        if not self.lazy_softmax:
            self.engine.rec_forward(x, REC_WANT_SOFTMAX, out=(idx, prob, full))
            return self._to_host(full)
        # (argmax, max) ride to the host behind the forward, in front of the range guard's synchronisation: one sync per call
        n = B * T
        if getattr(self, "_stat_pin", None) is None or self._stat_pin.numel() < 2 * n:
            self._stat_pin = torch.empty(2 * n + 4096, dtype=torch.int32, pin_memory=True)
        pin_i, pin_p = self._stat_pin[:n].view(B, T), self._stat_pin[n:2 * n].view(B, T).view(torch.float32)

        def copy_stats():
            pin_i.copy_(idx, non_blocking=True)
            pin_p.copy_(prob, non_blocking=True)
        self.engine.rec_forward(x, REC_WANT_SOFTMAX, out=(idx, prob, full), after_launch=copy_stats)
        if self.engine.guard != "sync" or self.engine.precision == "fp32":      # (no guard synchronisation happened)
            torch.cuda.current_stream(self.device).synchronize()
        lazy = LazySoftmax(self, full, pin_i.numpy(), pin_p.numpy())
        self._last_lazy[k] = weakref.ref(lazy)
        t2 = time.perf_counter()
        self.host_ms["stage_in"] += (t1 - t0) * 1e3        # previous result retired, input copied to pinned memory, upload enqueued
        self.host_ms["forward_and_wait"] += (t2 - t1) * 1e3
        self.host_ms["calls"] += 1
        return lazy

    def infer_indices(self, img: Union[np.ndarray, torch.Tensor]) -> Tuple[np.ndarray, np.ndarray]:
        if isinstance(img, torch.Tensor):
            idx, prob, _ = self.engine.rec_forward(img)
            return idx.cpu().numpy(), prob.cpu().numpy()
        with torch.cuda.stream(self.stream):
            idx, prob, _ = self.engine.rec_forward(self._to_dev(img))
            return idx.cpu().numpy(), prob.cpu().numpy()


class Mi355LayoutBackboneSession(_BaseSession):
    """PPHGNetV2-B4 backbone of PP-DocLayout-L/plus-L/V2/V3: [B,3,S,S] -> 4 NCHW feature maps."""
    kind = "pphgnetv2_b4"

    def __call__(self, img: np.ndarray) -> List[np.ndarray]:
        with torch.cuda.stream(self.stream):
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
