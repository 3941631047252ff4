"""CPU: formula pre-processing geometry and constants (rapiddoc_amd/formula_host.py) - known answers derived from the
reference's definitions (pp_formulanet_plus/pre_process.py:39-246)."""
import numpy as np

from rapiddoc_amd import formula_host as F


def _img(h, w, box, ink=100):
    im = np.full((h, w, 3), 255, np.uint8)
    y0, x0, y1, x1 = box
    im[y0:y1, x0:x1] = ink
    im[y0, x0] = 0          # one darker pixel so that the min/max normalisation has a range
    return im


def test_crop_resize_pad_geometry():
    # a 40x200 black bar inside a 300x500 white canvas: margin cropped to the bar, short side -> 384 would overflow the
    # width, so thumbnail() shrinks to width 384, height round(40*384/200)=77 (PIL thumbnail rounding), centred
    out = F.decode_image(_img(300, 500, (100, 150, 140, 350)))
    assert out.shape == (384, 384, 3)
    rows = np.nonzero((out[..., 0] > 50).any(axis=1))[0]      # the grey (100) bar; padding is 0 (ImageOps.expand default)
    cols = np.nonzero((out[..., 0] > 50).any(axis=0))[0]
    assert cols[0] == 0 and cols[-1] == 383
    assert 74 <= len(rows) <= 78 and abs((rows[0] + rows[-1]) / 2 - 191.5) <= 1.0
    assert out[0, 0, 0] == 0 and out[-1, -1, 0] == 0


def test_uniform_image_is_not_cropped_and_empty_is_none():
    out = F.decode_image(np.full((50, 80, 3), 200, np.uint8))
    assert out.shape == (384, 384, 3)


def test_normalisation_and_latex_format():
    im = np.zeros((40, 50, 3), np.uint8)
    im[..., 0], im[..., 1], im[..., 2] = 10, 100, 250
    x = F.to_network_input(im)
    assert x.shape == (1, 1, 48, 64) and x.dtype == np.float32
    n = lambda v: (np.float32(v) * np.float32(1 / 255.0) - np.float32(0.7931)) / np.float32(0.1738)
    exp = np.float32(0.114) * n(10) + np.float32(0.587) * n(100) + np.float32(0.299) * n(250)
    assert abs(float(x[0, 0, 0, 0]) - float(exp)) < 1e-6
    assert (x[0, 0, 40:, :] == 1).all() and (x[0, 0, :, 50:] == 1).all()


def test_preprocess_batch_shapes():
    xs = F.preprocess([_img(100, 300, (30, 40, 70, 260)), _img(400, 100, (10, 10, 390, 90))])
    assert all(x.shape == (1, 1, 384, 384) for x in xs)


def test_token_decoder_from_tokenizer_json(tmp_path):
    """a16: ids -> str through `tokenizers` (the library the reference's UniMERNetDecode wraps) + the LaTeX fix-ups.  The
    shipped tokenizer JSON is download-only, so a small BPE tokenizer of the same kind (byte-level, <s>/<pad>/</s>/<unk> =
    0/1/2/3) is built here; the oracle for the decode stage is the library itself."""
    import json
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers
    from rapiddoc_amd.latex_post import latex_postprocess
    vocab = {"<s>": 0, "<pad>": 1, "</s>": 2, "<unk>": 3}
    for t in ["\\", "f", "r", "a", "c", "{", "}", "x", "y", "^", "2", "\u0120", "\\f", "\\fr", "\\fra", "\\frac", "\u0120{", "\u0120}",
              "l", "e", "t", "i", "g", "h", "\\l", "\\le", "\\lef", "\\left", "(", ")", "\u0120(", "\u0120)"]:
        vocab.setdefault(t, len(vocab))
    merges = [("\\", "f"), ("\\f", "r"), ("\\fr", "a"), ("\\fra", "c"), ("\u0120", "{"), ("\u0120", "}"), ("\\", "l"), ("\\l", "e"),
              ("\\le", "f"), ("\\lef", "t"), ("\u0120", "("), ("\u0120", ")")]
    tok = Tokenizer(models.BPE(vocab=vocab, merges=merges, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    tok.add_special_tokens(["<s>", "<pad>", "</s>", "<unk>"])
    path = tmp_path / "tokenizer.json"
    tok.save(str(path))
    text = "\\frac {x} {y}^2"
    ids = [0] + tok.encode(text).ids + [2]
    for src in (str(path), json.loads(path.read_text()), path.read_text()):
        dec = F.make_token_decoder(src, fix_text=None)
        assert dec(ids) == latex_postprocess(tok.decode(ids, skip_special_tokens=True)) == latex_postprocess(text)
    upper = F.make_token_decoder(str(path), fix_text=str.upper)      # the ftfy hook is applied last
    assert upper(ids) == latex_postprocess(text).upper()
    # an unbalanced \\left( is repaired by the LaTeX stage between tokenizer and fix_text (post_process.py:350-381)
    ids2 = [0] + tok.encode("\\left (x").ids + [2]
    assert F.make_token_decoder(str(path), None)(ids2) == latex_postprocess(tok.decode(ids2, skip_special_tokens=True))


# ---- pinned to the reference's own PPPreProcess (tests/golden/make_golden_formula_pre.py: pre_process.py run unmodified, PIL for the
# resampling, its four cv2 calls stood in for by their definitions) ---------------------------------------------------------------
import json as _json
import sys as _sys
import zlib as _zlib
from pathlib import Path as _Path

import pytest as _pytest

_GOLD = _Path(__file__).parent / "golden"
_PRE = _json.loads((_GOLD / "formula_pre.json").read_text())["cases"]


@_pytest.mark.parametrize("case", _PRE, ids=lambda c: f"{c['seed']}-{c['hw'][0]}x{c['hw'][1]}")
def test_preprocess_matches_the_reference_byte_for_byte(case):
    _sys.path.insert(0, str(_GOLD))
    import make_golden_formula_pre as M               # only its seeded image generator (no reference import at module level)
    img = M.make_image(case["seed"], *case["hw"])
    decoded = F.decode_image(img)
    assert list(decoded.shape) == case["decoded_shape"]
    assert _zlib.crc32(np.ascontiguousarray(decoded).tobytes()) == case["decoded_crc32"]
    x = F.preprocess([img])[0]
    assert list(x.shape) == case["input_shape"] and str(x.dtype) == case["input_dtype"]
    assert _zlib.crc32(np.ascontiguousarray(x).tobytes()) == case["input_crc32"]
