"""Weight containers for the MI355X engine.

The reference ships PP-OCRv6 weights as ``.safetensors`` files and loads them with
``safetensors.torch.load_file`` + ``load_state_dict`` (reference
``rapid_doc/model/ocr/torch.py:93-110``: a leading ``model.`` prefix is stripped).  The engine's C-ABI
``rd_load_weights`` takes exactly that byte image, so real weights drop in unchanged.

No real weights exist in the build container (``.MISSING_LARGE_BLOBS``), so tests and benchmarks use
*synthetic* weights generated deterministically from a manifest of (name, shape) pairs that was captured
from the reference's own ``state_dict()`` (``tests/golden/manifest_*.json``).  The generator is
numpy-only so the same bytes are produced in the build container (where golden outputs are minted with
the reference definitions) and on the GPU box.
"""
from __future__ import annotations

import json
import struct
import zlib
from pathlib import Path
from typing import Dict, Iterable, List, Tuple

import numpy as np

Manifest = List[Tuple[str, Tuple[int, ...], str]]

_NORM_TOKENS = (".normalization.", ".norm.", ".bn.", "layer_norm", ".norm1.", ".norm2.")


def load_manifest(path) -> Manifest:
    raw = json.loads(Path(path).read_text())
    return [(n, tuple(s), d) for n, s, d in raw]


def _is_norm(name: str) -> bool:
    if any(tok in name for tok in _NORM_TOKENS):
        return True
    # `head.encoder.norm.weight` style (LightSVTR final LayerNorm)
    stem = name.rsplit(".", 1)[0]
    return stem.endswith(".norm") or stem.endswith("norm")


def synth_tensor(name: str, shape: Tuple[int, ...], dtype: str, seed: int) -> np.ndarray:
    """One synthetic tensor, a pure function of (name, shape, seed)."""
    rng = np.random.default_rng([seed, zlib.crc32(name.encode())])
    leaf = name.rsplit(".", 1)[-1]
    if leaf == "num_batches_tracked":
        return np.zeros(shape, dtype=np.int64)
    if leaf == "running_mean":
        return rng.normal(0.0, 0.1, shape).astype(np.float32)
    if leaf == "running_var":
        return rng.uniform(0.5, 1.5, shape).astype(np.float32)
    if len(shape) <= 1 and leaf == "weight" and _is_norm(name):
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if leaf == "bias":
        return rng.normal(0.0, 0.05, shape).astype(np.float32)
    if leaf in ("scale",):
        return rng.uniform(0.8, 1.2, shape).astype(np.float32)
    if leaf == "weight" and len(shape) >= 2:
        fan_in = int(np.prod(shape[1:]))
        # ConvTranspose2d stores [Cin, Cout, kh, kw]; its fan-in is Cin.
        if "conv_up" in name or "conv_final" in name:
            fan_in = shape[0]
        # gain 1.6 (most convs here are linear or followed by a residual add); the second point-wise conv
        # of a residual mixer is damped so that 20+ stacked blocks keep activations O(1..10).
        std = (1.6 / max(fan_in, 1)) ** 0.5
        if ".channel_conv2." in name or ".aggregation_excitation_conv." in name or ".mlp.fc2." in name:
            std *= 0.5
        if name == "head.head.weight":  # CTC classifier: spread the logits so argmax varies over time
            std *= 6.0
        return rng.normal(0.0, std, shape).astype(np.float32)
    return rng.normal(0.0, 0.05, shape).astype(np.float32)


def synth_state_dict(manifest: Manifest, seed: int = 0) -> Dict[str, np.ndarray]:
    return {name: synth_tensor(name, shape, dtype, seed) for name, shape, dtype in manifest}


def checksum(state: Dict[str, np.ndarray]) -> float:
    """Order-independent float64 checksum used to pin the generator across machines."""
    tot = 0.0
    for name in sorted(state):
        tot += float(np.asarray(state[name], dtype=np.float64).sum())
    return tot


_ST_DTYPES = {"float32": "F32", "int64": "I64", "float16": "F16", "int32": "I32", "uint8": "U8"}
_ST_NP = {v: k for k, v in _ST_DTYPES.items()}


def to_safetensors_bytes(state: Dict[str, np.ndarray], skip_int: bool = False) -> bytes:
    """Serialise to the safetensors byte image (8-byte LE header length, JSON header, raw data)."""
    header = {}
    chunks = []
    off = 0
    for name, arr in state.items():
        arr = np.asarray(arr, order="C")          # (np.ascontiguousarray would turn a 0-d tensor - num_batches_tracked - into shape [1])
        if skip_int and arr.dtype.kind in "iu":
            continue
        raw = arr.tobytes()
        header[name] = {
            "dtype": _ST_DTYPES[str(arr.dtype)],
            "shape": list(arr.shape),
            "data_offsets": [off, off + len(raw)],
        }
        chunks.append(raw)
        off += len(raw)
    hjson = json.dumps(header, separators=(",", ":")).encode()
    pad = (8 - len(hjson) % 8) % 8
    hjson += b" " * pad
    return struct.pack("<Q", len(hjson)) + hjson + b"".join(chunks)


def from_safetensors_bytes(blob: bytes) -> Dict[str, np.ndarray]:
    (hlen,) = struct.unpack("<Q", blob[:8])
    header = json.loads(blob[8 : 8 + hlen])
    base = 8 + hlen
    out = {}
    for name, meta in header.items():
        if name == "__metadata__":
            continue
        b, e = meta["data_offsets"]
        arr = np.frombuffer(blob[base + b : base + e], dtype=_ST_NP[meta["dtype"]]).reshape(meta["shape"])
        out[name] = arr
    return out


def strip_model_prefix(state: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Reference ``torch.py:105-110``: drop a leading ``model.`` from every key."""
    if any(k.startswith("model.") for k in state):
        return {k[len("model."):] if k.startswith("model.") else k: v for k, v in state.items()}
    return state
