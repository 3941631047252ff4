"""Thin Python handle over the C-ABI: PyTorch-ROCm tensors provide device memory and the stream, the
library does all the arithmetic."""
from __future__ import annotations

import ctypes as C
import json
import os
from typing import Dict, List, Optional, Tuple, Union

import numpy as np
import torch

from . import _lib, weights as W

REC_UNFUSED_CTC, REC_WANT_SOFTMAX, REC_WANT_LOGITS = 1, 2, 4
KINDS = ("ppocrv6_det", "ppocrv6_rec", "pphgnetv2_b4", "pphgnetv2_b6_formula", "ppformulanet_head")


def rec_line_table(widths, first_tokens) -> np.ndarray:
    """int32 [B, 4] line table of rd_rec_backbone_forward_lines (include/rapiddoc_mi355.h): (w, width after stem1, width after stem3 =
    the blocks' width, first token); a line of padded width w yields (w4 // 2) = rd_rec_seq_len(w) tokens."""
    w = np.asarray(widths, dtype=np.int64).reshape(-1)
    w2 = (w - 1) // 2 + 1
    w4 = (w2 - 1) // 2 + 1
    return np.ascontiguousarray(np.stack([w, w2, w4, np.asarray(first_tokens, dtype=np.int64).reshape(-1)], axis=1).astype(np.int32))


def ragged_tables(line_lengths: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
    """(seg int32 [n][2] = (first token, tokens) per line, tokinfo int32 [n_tokens] = position | tokens << 16)."""
    lens = np.asarray(line_lengths, dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(lens)[:-1]])
    seg = np.stack([off, lens], axis=1).astype(np.int32)
    pos = np.arange(int(lens.sum()), dtype=np.int64) - np.repeat(off, lens)
    tokinfo = (pos | (np.repeat(lens, lens) << 16)).astype(np.int32)
    return seg, tokinfo


TAIL_LINE_BUCKET, TAIL_T_BUCKET, TAIL_TOKEN_BUCKET = 32, 8, 256


def pad_tail_lengths(line_lengths) -> Tuple[np.ndarray, int]:
    """Line lengths of one `rec_tail_forward` call padded with DUMMY lines so that the plan key of the call - (lines, longest
    line, tokens) - falls on a coarse grid (multiples of 32 lines / 8 tokens / 256 tokens).  On real documents almost every
    group of rec batches has a new total token count; an exact key would build a new plan (and possibly grow the workspace,
    with a stream sync) per call.  Returns (padded lengths = the real lines followed by the dummy ones, padded longest line).
    The dummy tokens must be finite (zero) in the token buffer; their outputs sit behind the real ones and are ignored."""
    lens = np.asarray(line_lengths, dtype=np.int64)
    n, tok = len(lens), int(lens.sum())
    T = -(-int(lens.max()) // TAIL_T_BUCKET) * TAIL_T_BUCKET
    d = (n // TAIL_LINE_BUCKET + 1) * TAIL_LINE_BUCKET - n          # 1 .. 32 dummy lines
    while True:
        pad = -(-(tok + d) // TAIL_TOKEN_BUCKET) * TAIL_TOKEN_BUCKET - tok      # >= d: every dummy line gets >= 1 token
        if pad <= d * T:                                                      # and none is longer than the longest line
            break
        d += TAIL_LINE_BUCKET
    base, extra = divmod(pad, d)
    dummy = np.full(d, base, dtype=np.int64)
    dummy[:extra] += 1
    return np.concatenate([lens, dummy]), T


class EngineError(RuntimeError):
    pass


def _stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


class RdEngine:
    """One network on one GPU (`rd_handle`).

    Range guard of the split-fp16 kernels (include/rapiddoc_mi355.h, rd_range_status) lives HERE, once, for every
    forward of every network kind: with `guard="sync"` (default) a forward that raised the range flag is repeated in
    native fp32 before it returns (one stream synchronisation per call); `guard="deferred"` is for callers that keep
    several forwards in flight (PagePipeline) and promise to call `check_range_and_fallback()` before they use the
    results; `guard="off"` leaves the flag to the caller (tests)."""

    def __init__(self, kind: str, device: int = 0, guard: str = "sync", reuse_outputs: bool = False):
        """`reuse_outputs`: the output tensors of a forward are cached per shape and handed out again by the next call with
        that shape (a later call overwrites what an earlier one returned).  For callers that consume a result before they
        issue the next forward of the same shape (PagePipeline): stable addresses are what lets the library replay a forward
        as one hipGraph launch instead of 30-90 kernel launches (csrc/engine.cpp, Engine::run)."""
        if kind not in KINDS:
            raise ValueError(f"kind must be one of {KINDS}")
        if not torch.cuda.is_available():
            raise EngineError("no ROCm device visible: rapiddoc_amd runs on MI355X only (no CPU fallback)")
        self._l = _lib.load()
        self.kind, self.device = kind, device
        self._h = self._l.rd_create(device, kind.encode())
        if not self._h:
            raise EngineError(self._l.rd_create_error().decode())
        self._tdev = torch.device("cuda", device)
        self._profiling = False
        if guard not in ("sync", "deferred", "off"):
            raise ValueError("guard must be 'sync', 'deferred' or 'off'")
        self.guard = guard
        self.range_fallbacks = 0
        self.reuse_outputs = reuse_outputs
        self._out_cache: Dict[tuple, torch.Tensor] = {}
        self.precision = os.environ.get("RD_PRECISION", "auto")   # the library reads the same variable in rd_create
        self.profile_log: List[dict] = []

    def close(self):
        if getattr(self, "_h", None):
            self._l.rd_destroy(self._h)
            self._h = None

    __del__ = close

    def _chk(self, rc: int):
        if rc != 0:
            raise EngineError(self._l.rd_last_error(self._h).decode())

    # ------------------------------------------------------------------ weights
    def load_weights(self, src: Union[bytes, str, Dict[str, np.ndarray]]):
        """`src`: path to / bytes of a .safetensors file, or a name->ndarray state dict."""
        if isinstance(src, dict):
            blob = W.to_safetensors_bytes({k: np.asarray(v) for k, v in src.items()}, skip_int=True)
        elif isinstance(src, (bytes, bytearray)):
            blob = bytes(src)
        else:
            blob = open(src, "rb").read()
        buf = C.create_string_buffer(blob, len(blob))
        self._chk(self._l.rd_load_weights(self._h, buf, len(blob)))
        return self

    @property
    def num_classes(self) -> int:
        return self._l.rd_rec_num_classes(self._h)

    def workspace_bytes(self, B: int, H: int, W_: int, flags: int = 0) -> int:
        n = C.c_size_t(0)
        self._chk(self._l.rd_query_workspace(self._h, B, H, W_, flags, C.byref(n)))
        return n.value

    # ------------------------------------------------------------------ forwards
    def _out(self, tag: str, shape: tuple, dtype, device) -> torch.Tensor:
        if not self.reuse_outputs:
            return torch.empty(shape, dtype=dtype, device=device)
        key = (tag, tuple(shape), dtype)
        t = self._out_cache.get(key)
        if t is None:
            if len(self._out_cache) > 64:
                self._out_cache.clear()
            t = self._out_cache[key] = torch.empty(shape, dtype=dtype, device=device)
        return t

    def _prep(self, x: torch.Tensor) -> torch.Tensor:
        if x.device.type != "cuda":
            x = x.to(self._tdev, non_blocking=True)
        if x.dtype != torch.float32 or not x.is_contiguous():
            x = x.contiguous().float()
        return x

    def det_forward(self, x: torch.Tensor, out: Optional[torch.Tensor] = None, after_launch=None) -> torch.Tensor:
        """`out` / `after_launch`: as in `rec_forward` (a caller-owned [B,1,H,W] float32 result tensor; work enqueued behind the launch)."""
        x = self._prep(x)
        B, Cc, H, W_ = x.shape
        if out is None:
            out = self._out("det", (B, 1, H, W_), torch.float32, x.device)
        elif out.shape != (B, 1, H, W_) or out.dtype != torch.float32 or not out.is_contiguous():
            raise EngineError("det_forward: `out` does not match the forward's shape")

        def launch():
            self._chk(self._l.rd_det_forward(self._h, x.data_ptr(), B, H, W_, out.data_ptr(), None, 0, _stream_ptr()))
            self._log()
            if after_launch is not None:
                after_launch()
        self._guarded(launch)
        return out

    def rec_forward(self, x: torch.Tensor, flags: int = 0, out: Optional[tuple] = None,
                    after_launch=None) -> Tuple[torch.Tensor, torch.Tensor, Optional[torch.Tensor]]:
        """`out`: (idx int32 [B,T], prob float32 [B,T], full float32 [B,T,C] or None) contiguous device tensors to write into - a caller
        that hands over the SAME addresses call after call (session.Mi355RecSession) lets the library replay the forward as one
        hipGraph launch.  `after_launch()`: enqueued behind every (re)launch, in front of the range guard's synchronisation (e.g. the
        caller's device -> host copies, so that one synchronisation serves both)."""
        x = self._prep(x)
        B, Cc, H, W_ = x.shape
        if H != 48:
            raise EngineError("rec input height must be 48")
        T = self._l.rd_rec_seq_len(W_)
        want_full = bool(flags & (REC_WANT_SOFTMAX | REC_WANT_LOGITS))
        if out is not None:
            idx, prob, full = out
            ok = (idx.shape == (B, T) and idx.dtype == torch.int32 and idx.is_contiguous() and prob.shape == (B, T)
                  and prob.dtype == torch.float32 and prob.is_contiguous()
                  and (not want_full or (full is not None and full.shape == (B, T, self.num_classes) and full.dtype == torch.float32
                                         and full.is_contiguous())))
            if not ok:
                raise EngineError("rec_forward: `out` tensors do not match the forward's shapes")
        else:
            idx = torch.empty((B, T), dtype=torch.int32, device=x.device)
            prob = torch.empty((B, T), dtype=torch.float32, device=x.device)
            full = torch.empty((B, T, self.num_classes), dtype=torch.float32, device=x.device) if want_full else None

        def launch():
            self._chk(self._l.rd_rec_forward(self._h, x.data_ptr(), B, W_, idx.data_ptr(), prob.data_ptr(),
                                             full.data_ptr() if want_full else None, flags, None, 0, _stream_ptr()))
            self._log()
            if after_launch is not None:
                after_launch()
        self._guarded(launch)
        return idx, prob, (full if want_full else None)

    # two-stage form of the recogniser (include/rapiddoc_mi355.h: rd_rec_backbone_forward / rd_rec_tail_forward)
    @property
    def rec_token_dim(self) -> int:
        return self._l.rd_rec_token_dim(self._h)

    def rec_backbone_forward(self, x: torch.Tensor, tokens_out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [B,3,48,W] -> pooled backbone tokens [B, T, dim]; `tokens_out`: a contiguous float32 view to write them into."""
        x = self._prep(x)
        B, Cc, H, W_ = x.shape
        if H != 48:
            raise EngineError("rec input height must be 48")
        T = self._l.rd_rec_seq_len(W_)
        if tokens_out is None:
            tokens_out = torch.empty((B, T, self.rec_token_dim), dtype=torch.float32, device=x.device)
        if tokens_out.numel() != B * T * self.rec_token_dim or not tokens_out.is_contiguous() or tokens_out.dtype != torch.float32:
            raise EngineError("tokens_out must be a contiguous float32 tensor of B*T*dim elements")

        def launch():
            self._chk(self._l.rd_rec_backbone_forward(self._h, x.data_ptr(), B, W_, tokens_out.data_ptr(), None, 0, _stream_ptr()))
            self._log()
        self._guarded(launch)
        return tokens_out

    def rec_backbone_forward_lines(self, x: torch.Tensor, line_tab: torch.Tensor, tokens_out: torch.Tensor) -> torch.Tensor:
        """The backbone stage over lines of DIFFERENT reference padded widths (rd_rec_backbone_forward_lines): x [B,3,48,W] with
        line b in columns [0, w_b), `line_tab` = `rec_line_table(widths, first tokens)` on the device, `tokens_out` the token buffer
        the table's offsets refer to.  Line b's tokens == rec_backbone_forward(x[b:b+1, :, :, :w_b])."""
        x = self._prep(x)
        B, Cc, H, W_ = x.shape
        if H != 48:
            raise EngineError("rec input height must be 48")
        if line_tab.dtype != torch.int32 or not line_tab.is_cuda or line_tab.numel() != 4 * B or not line_tab.is_contiguous():
            raise EngineError("line_tab must be a contiguous int32 device tensor [B, 4]")
        if not tokens_out.is_contiguous() or tokens_out.dtype != torch.float32:
            raise EngineError("tokens_out must be a contiguous float32 tensor")

        def launch():
            self._chk(self._l.rd_rec_backbone_forward_lines(self._h, x.data_ptr(), B, W_, line_tab.data_ptr(), tokens_out.data_ptr(), None, 0,
                                                            _stream_ptr()))
            self._log()
        self._guarded(launch)
        return tokens_out

    def rec_tail_tables(self, line_lengths, device, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """The two int32 tables rd_rec_tail_forward wants, as one device tensor [2 * n_lines + n_tokens] (seg, then tokinfo).
        `out`: a device int32 buffer at least that long to fill instead of a new tensor (a view of its head is returned)."""
        seg_h, tokinfo_h = ragged_tables(np.asarray(line_lengths, dtype=np.int64))
        host = torch.from_numpy(np.concatenate([seg_h.reshape(-1), tokinfo_h])).pin_memory()
        if out is None:
            return host.to(device, non_blocking=True)
        view = out[: host.numel()]
        view.copy_(host, non_blocking=True)
        self._keep_host = getattr(self, "_keep_host", [])[-15:] + [host]      # the pinned source must outlive the asynchronous copy
        return view

    def rec_tail_forward(self, tokens: torch.Tensor, line_lengths, tables: Optional[torch.Tensor] = None,
                         max_tokens: Optional[int] = None, out: Optional[Tuple[torch.Tensor, torch.Tensor]] = None) -> Tuple[torch.Tensor, torch.Tensor]:
        """tokens [n_tokens, dim] = the text lines back to back, `line_lengths` their token counts -> (idx, prob) [n_tokens].
        `tables`: rec_tail_tables(line_lengths) uploaded earlier (e.g. before the backbones were enqueued).  `max_tokens`: an
        upper bound of the longest line used as the plan key instead of the exact maximum (see `pad_tail_lengths`).  `out`:
        (idx int32 [n_tokens], prob float32 [n_tokens]) to write into."""
        lens = np.asarray(line_lengths, dtype=np.int64)
        n_tokens = int(lens.sum())
        longest = int(lens.max()) if max_tokens is None else int(max_tokens)
        if longest < int(lens.max()):
            raise EngineError("rec tail: max_tokens below the longest line")
        if tokens.numel() != n_tokens * self.rec_token_dim or lens.min() < 1 or lens.max() >= 32768:
            raise EngineError("rec tail: token count / line lengths mismatch")
        dev = tokens.device
        if tables is None:
            tables = self.rec_tail_tables(lens, dev)
        seg, tokinfo = tables[: 2 * len(lens)], tables[2 * len(lens):]
        if out is not None:
            idx, prob = out
            assert idx.numel() == n_tokens and prob.numel() == n_tokens and idx.dtype == torch.int32 and prob.dtype == torch.float32
        else:
            idx = torch.empty((n_tokens,), dtype=torch.int32, device=dev)
            prob = torch.empty((n_tokens,), dtype=torch.float32, device=dev)

        def launch():
            self._chk(self._l.rd_rec_tail_forward(self._h, tokens.data_ptr(), n_tokens, len(lens), longest, seg.data_ptr(),
                                                  tokinfo.data_ptr(), idx.data_ptr(), prob.data_ptr(), None, 0, _stream_ptr()))
            self._log()
        self._guarded(launch)
        self._keep = (tables,)          # the tables must outlive the asynchronous launch
        return idx, prob

    def backbone_forward(self, x: torch.Tensor) -> List[torch.Tensor]:
        x = self._prep(x)
        B, Cc, H, W_ = x.shape
        chans = (128, 512, 1024, 2048)
        feats = [self._out("feat%d" % s, (B, c, H // s, W_ // s), torch.float32, x.device)
                 for c, s in zip(chans, (4, 8, 16, 32))]
        arr = (C.c_void_p * 4)(*[f.data_ptr() for f in feats])

        def launch():
            self._chk(self._l.rd_backbone_forward(self._h, x.data_ptr(), B, H, W_, arr, None, 0, _stream_ptr()))
            self._log()
        self._guarded(launch)
        return feats

    def formula_encoder_forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [B,1|3,H,W] -> encoder states [B, (H/32)*(W/32), 2048] (PPHGNetV2_B6_Formula.last_hidden_state)."""
        x = self._prep(x)
        B, Cc, H, W_ = x.shape
        out = torch.empty((B, (H // 32) * (W_ // 32), 2048), dtype=torch.float32, device=x.device)

        def launch():
            self._chk(self._l.rd_formula_encoder_forward(self._h, x.data_ptr(), B, Cc, H, W_, out.data_ptr(), None, 0, _stream_ptr()))
            self._log()
        self._guarded(launch)
        return out

    def formula_decode(self, enc: torch.Tensor, max_new_tokens: int) -> torch.Tensor:
        """enc [B,S,2048] -> token ids [B,L] int64 exactly as PPFormulaNet_Head.forward returns them."""
        enc = self._prep(enc)
        B, S, _ = enc.shape
        ids = torch.empty((B, max_new_tokens + 1), dtype=torch.int64, device=enc.device)
        n = C.c_int32(0)
        self._chk(self._l.rd_formula_decode(self._h, enc.data_ptr(), B, S, max_new_tokens, ids.data_ptr(), C.byref(n), _stream_ptr()))
        return ids[:, : n.value]

    @property
    def formula_max_new_tokens(self) -> int:
        return self._l.rd_formula_max_new_tokens(self._h)

    # ------------------------------------------------------------------ precision (include/rapiddoc_mi355.h)
    def set_precision(self, mode: str):
        """'auto' (default: split-fp16 channel mixers, fp32 MFMA elsewhere), 'fp32' (native fp32 MFMA only) or 'h3'."""
        self._chk(self._l.rd_set_precision(self._h, mode.encode()))
        self.precision = mode
        return self

    def _guarded(self, launch) -> None:
        """Run `launch` (enqueue one forward); in "sync" mode repeat it in native fp32 if a split kernel flagged an
        operand outside the fp16 range - a result is never silently wrong, whichever entry point produced it."""
        launch()
        if self.guard == "sync" and self.precision != "fp32" and self.range_overflow():
            self.set_precision("fp32")
            self.range_fallbacks += 1
            launch()

    def check_range_and_fallback(self) -> bool:
        """Deferred guard: True when forwards since the last check overflowed; the handle is then switched to fp32 and
        the CALLER must repeat those forwards."""
        if self.precision != "fp32" and self.range_overflow():
            self.set_precision("fp32")
            self.range_fallbacks += 1
            return True
        return False

    def range_overflow(self) -> bool:
        """True when a split-fp16 kernel met an operand outside the fp16 range since the last call (synchronises the
        current stream, clears the flag).  The results of those calls are invalid: switch to 'fp32' and repeat them."""
        rc = self._l.rd_range_status(self._h, _stream_ptr())
        if rc < 0:
            raise EngineError(self._l.rd_last_error(self._h).decode())
        return rc == 1

    # ------------------------------------------------------------------ profiling
    def plan_stats(self) -> dict:
        """{plans_built, graph_captures, graph_replays} of this handle since it was created (rd_plan_stats)."""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._chk(self._l.rd_plan_stats(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return {"plans_built": int(a.value), "graph_captures": int(b.value), "graph_replays": int(c.value)}

    def set_profiling(self, on: bool):
        self._chk(self._l.rd_set_profiling(self._h, 1 if on else 0))
        self._profiling = bool(on)

    def _log(self):
        if self._profiling:
            self.profile_log.extend(self.profile())

    def profile(self) -> List[dict]:
        return json.loads(self._l.rd_profile_json(self._h).decode())


def preproc_resize_norm(img_u8_hwc: torch.Tensor, out_hw: Tuple[int, int], mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0),
                        scale: float = 1.0 / 255.0, interp: int = 2, swap_rb: bool = False,
                        out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """u8 HWC device image -> normalised CHW float32 (PPPreProcess, pp_doclayout/pre_process.py:22-42)."""
    lib = _lib.load()
    assert img_u8_hwc.is_cuda and img_u8_hwc.dtype == torch.uint8 and img_u8_hwc.is_contiguous()
    H, W_, ch = img_u8_hwc.shape
    assert ch == 3
    OH, OW = out_hw
    if out is None:
        out = torch.empty((3, OH, OW), dtype=torch.float32, device=img_u8_hwc.device)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    rc = lib.rd_preproc_resize_norm(img_u8_hwc.device.index or 0, img_u8_hwc.data_ptr(), H, W_, OH, OW, m, s, scale,
                                    interp, 1 if swap_rb else 0, out.data_ptr(), _stream_ptr())
    if rc != 0:
        raise EngineError("rd_preproc_resize_norm failed")
    return out


def preproc_resize_norm_batch(imgs_u8_nhwc: torch.Tensor, out_hw: Tuple[int, int], mean=(0.0, 0.0, 0.0), std=(1.0, 1.0, 1.0),
                              scale: float = 1.0 / 255.0, interp: int = 2, swap_rb: bool = False,
                              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """`preproc_resize_norm` for a whole [P,H,W,3] u8 array in one launch -> [P,3,OH,OW] float32."""
    lib = _lib.load()
    assert imgs_u8_nhwc.is_cuda and imgs_u8_nhwc.dtype == torch.uint8 and imgs_u8_nhwc.is_contiguous() and imgs_u8_nhwc.dim() == 4
    P, H, W_, ch = imgs_u8_nhwc.shape
    assert ch == 3
    OH, OW = out_hw
    if out is None:
        out = torch.empty((P, 3, OH, OW), dtype=torch.float32, device=imgs_u8_nhwc.device)
    assert out.is_contiguous() and tuple(out.shape) == (P, 3, OH, OW)
    m = (C.c_float * 3)(*mean)
    s = (C.c_float * 3)(*std)
    rc = lib.rd_preproc_resize_norm_batch(imgs_u8_nhwc.device.index or 0, imgs_u8_nhwc.data_ptr(), P, H, W_, OH, OW, m, s, scale,
                                          interp, 1 if swap_rb else 0, out.data_ptr(), _stream_ptr())
    if rc != 0:
        raise EngineError("rd_preproc_resize_norm_batch failed")
    return out
