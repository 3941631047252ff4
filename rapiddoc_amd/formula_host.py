"""Host side of the formula stage (PP-FormulaNet_plus): image pre-processing and the batch_predict-shaped driver.

Reference: rapid_doc/model/formula/rapid_formula_self/model_handler/pp_formulanet_plus/pre_process.py:12-256
  UniMERNetImgDecode  crop the margin (normalised grey < 200), PIL bilinear resize so the short side is 384, thumbnail to
                      fit 384x384, centre-pad with 0                              (:39-163)
  UniMERNetTestTransform  (x/255 - 0.7931) / 0.1738, grey = .299 R + .587 G + .114 B of the *RGB-ordered* array read as BGR
                          by cv2.cvtColor(COLOR_BGR2GRAY), i.e. .114 c0 + .587 c1 + .299 c2      (:188-208)
  LatexImageFormat    pad to a multiple of 16 with 1, keep one channel -> [1,1,H,W]                (:229-246)
PIL does the resampling exactly like the reference (PIL is what the reference calls); the three cv2 calls
(findNonZero, boundingRect, cvtColor) are restated in numpy - cv2 is not installed here, so those are parity-unpinned.
The driver mirrors RapidFormulaModel.batch_predict (rapid_formula_model.py:34-41 -> rapid_formula_self/main.py:28-41):
chunks of `batch_size`, one [B,1,384,384] tensor per chunk -> encoder -> greedy decoder -> token ids.
"""
from __future__ import annotations

import math
from typing import Callable, List, Optional, Sequence

import numpy as np
from PIL import Image, ImageOps

INPUT_SIZE = (384, 384)
MEAN, STD = 0.7931, 0.1738


def _crop_margin(img: Image.Image) -> Image.Image:
    data = np.array(img.convert("L")).astype(np.uint8)
    mx, mn = data.max(), data.min()
    if mx == mn:
        return img
    norm = (data - mn) / (mx - mn) * 255
    ys, xs = np.nonzero(norm < 200)
    if len(xs) == 0:      # cv2.boundingRect(None) -> (0, 0, 0, 0): an empty crop, which the caller treats as failure
        return img.crop((0, 0, 0, 0))
    a, b = int(xs.min()), int(ys.min())
    return img.crop((a, b, int(xs.max()) + 1, int(ys.max()) + 1))


def decode_image(img: np.ndarray, input_size=INPUT_SIZE) -> Optional[np.ndarray]:
    """UniMERNetImgDecode.img_decode: uint8 HxWx3 (or HxW) -> uint8 384x384x3, or None for an empty image."""
    pil = _crop_margin(Image.fromarray(img).convert("RGB"))
    if pil.height == 0 or pil.width == 0:
        return None
    w, h = pil.size
    short, long_ = (w, h) if w <= h else (h, w)
    new_short = min(input_size)
    new_long = int(new_short * long_ / short)
    new_w, new_h = (new_short, new_long) if w <= h else (new_long, new_short)
    pil = pil.resize((new_w, new_h), resample=2)            # PIL bilinear
    pil.thumbnail((input_size[1], input_size[0]))
    dw, dh = input_size[1] - pil.width, input_size[0] - pil.height
    pw, ph = dw // 2, dh // 2
    return np.array(ImageOps.expand(pil, (pw, ph, dw - pw, dh - ph)))


def to_network_input(img384: np.ndarray) -> np.ndarray:
    """UniMERNetTestTransform + LatexImageFormat: uint8 HxWx3 -> float32 [1,1,H16,W16]."""
    x = (img384.astype("float32") * float(1 / 255.0) - np.float32(MEAN)) / np.float32(STD)
    grey = (np.float32(0.114) * x[..., 0] + np.float32(0.587) * x[..., 1] + np.float32(0.299) * x[..., 2]).astype(np.float32)
    h, w = grey.shape
    H, W_ = math.ceil(h / 16) * 16, math.ceil(w / 16) * 16
    grey = np.pad(grey, ((0, H - h), (0, W_ - w)), constant_values=(1, 1))
    return grey[None, None]


def preprocess(imgs: Sequence[np.ndarray]) -> List[Optional[np.ndarray]]:
    out = []
    for im in imgs:
        d = decode_image(im)
        out.append(None if d is None else to_network_input(d))
    return out


def make_token_decoder(tokenizer_json, fix_text="auto") -> Callable[[List[int]], str]:
    """ids -> LaTeX string exactly as UniMERNetDecode.token2str produces it after its EOS cut
    (pp_formulanet_plus/post_process.py:92-94 builds `tokenizers.Tokenizer.from_buffer(fast_tokenizer_file JSON)`, :277-296
    `tokenizer.decode(ids, skip_special_tokens=True)`, :350-381 LaTeX fix-ups then `ftfy.fix_text`).

    tokenizer_json: path to / str of / dict of the `fast_tokenizer_file` JSON the reference downloads with the model
    (`PP-FormulaNet_plus-M_inference.yml` -> PostProcess.character_dict).  fix_text: a callable, None, or "auto" =
    `ftfy.fix_text` when that package is importable (it is not part of this container; the reference imports it lazily)."""
    import json
    import os
    from tokenizers import Tokenizer
    from .latex_post import latex_postprocess
    if isinstance(tokenizer_json, dict):
        tok = Tokenizer.from_str(json.dumps(tokenizer_json))
    elif isinstance(tokenizer_json, (str, os.PathLike)) and os.path.exists(str(tokenizer_json)):
        tok = Tokenizer.from_file(str(tokenizer_json))
    else:
        tok = Tokenizer.from_str(str(tokenizer_json))
    if fix_text == "auto":
        try:
            from ftfy import fix_text as _ft   # noqa: WPS433
            fix_text = _ft
        except ImportError:
            fix_text = None

    def decode(ids: List[int]) -> str:
        text = latex_postprocess(tok.decode([int(i) for i in ids], skip_special_tokens=True))
        return fix_text(text) if fix_text else text
    return decode


class FormulaRecognizer:
    """`batch_predict(image_list, batch_size) -> list[str]` like rapid_doc.model.custom.CustomBaseModel / RapidFormulaModel
    when a tokenizer is given (`tokenizer_json=`, see `make_token_decoder`).

    Returns one entry per input image: the decoded string when `token_decoder` (ids -> str, e.g. the reference's
    UniMERNetDecode.token2str with its downloaded tokenizer) is given, otherwise the list of generated token ids with the
    start token, everything from EOS on, and padding removed.  Alternatively pass `bpe_decode` (ids -> raw string: only the
    tokenizer, whose JSON the reference downloads) and, optionally, `fix_text` (ftfy's): the LaTeX fix-ups in between are
    this package's own (`rapiddoc_amd.latex_post`, pinned to the reference's)."""

    def __init__(self, weights, device: int = 0, max_new_tokens: Optional[int] = None,
                 token_decoder: Optional[Callable[[List[int]], str]] = None,
                 bpe_decode: Optional[Callable[[List[int]], str]] = None,
                 fix_text: Optional[Callable[[str], str]] = None, tokenizer_json=None):
        import torch  # noqa: F401  (device memory + stream only)
        from . import weights as W
        from .engine import RdEngine
        if isinstance(weights, (str, bytes)):
            blob = weights if isinstance(weights, bytes) else open(weights, "rb").read()
            state = W.strip_model_prefix(W.from_safetensors_bytes(blob))
        else:
            state = dict(weights)
        self.encoder = RdEngine("pphgnetv2_b6_formula", device).load_weights({k: v for k, v in state.items() if k.startswith("backbone.")})
        self.decoder = RdEngine("ppformulanet_head", device).load_weights({k: v for k, v in state.items() if k.startswith("head.")})
        self.max_new_tokens = max_new_tokens or self.decoder.formula_max_new_tokens
        if token_decoder is None and tokenizer_json is not None:
            token_decoder = make_token_decoder(tokenizer_json, fix_text if fix_text is not None else "auto")
        if token_decoder is None and bpe_decode is not None:
            from .latex_post import latex_postprocess

            def token_decoder(ids, _bpe=bpe_decode, _fix=fix_text):   # UniMERNetDecode.token2str after the EOS cut
                text = latex_postprocess(_bpe(ids))
                return _fix(text) if _fix else text
        self.token_decoder = token_decoder

    def batch_predict(self, image_list: Sequence[np.ndarray], batch_size: int = 16, **kwargs) -> list:
        import torch
        results: list = [None] * len(image_list)
        inputs = preprocess(image_list)
        valid = [i for i, x in enumerate(inputs) if x is not None]
        for beg in range(0, len(valid), batch_size):
            chunk = valid[beg: beg + batch_size]
            x = torch.from_numpy(np.concatenate([inputs[i] for i in chunk], axis=0)).cuda(self.encoder.device)
            ids = self.decoder.formula_decode(self.encoder.formula_encoder_forward(x), self.max_new_tokens).cpu().numpy()
            for row, i in zip(ids, chunk):
                toks = row[1:].tolist()
                if 2 in toks:
                    toks = toks[: toks.index(2)]          # UniMERNetDecode cuts at EOS (post_process.py:277-296)
                results[i] = self.token_decoder(toks) if self.token_decoder else toks
        return [r if r is not None else ([] if self.token_decoder is None else "") for r in results]
