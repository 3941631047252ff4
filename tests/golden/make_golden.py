#!/usr/bin/env python3
"""Mint golden vectors by importing the REFERENCE's own nn.Module definitions.

Runs only in the build container (needs /root/reference, which never travels to the GPU box).
What is committed next to this script is data only: weight-name manifests captured from the reference
``state_dict()`` and input/output tensors.  Weights are synthetic (``rapiddoc_amd.weights``), because the
shipped ``.safetensors``/``.onnx`` blobs are absent (``.MISSING_LARGE_BLOBS``).

    python tests/golden/make_golden.py            # rewrites tests/golden/*.json|*.npz

Reference entry points exercised:
  rapid_doc/model/ocr/ppocrv6_pytorch/modeling/architectures/base_model.py:9-106  (det, rec)
  rapid_doc/model/formula/rapid_formula_self/networks/backbones/rec_pphgnetv2.py:1445-1477 (B4 det=True)
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch
import yaml

HERE = Path(__file__).resolve().parent
ROOT = HERE.parents[1]
sys.path.insert(0, str(ROOT))
REF = Path("/root/reference")

from rapiddoc_amd import weights as W  # noqa: E402
from oracle import nets as O  # noqa: E402

SEED = 0


def make_input(shape, seed, kind):
    rng = np.random.default_rng(seed)
    if kind == "normal":
        return rng.standard_normal(shape).astype(np.float32)
    if kind == "unit":
        return rng.uniform(0.0, 1.0, shape).astype(np.float32)
    return rng.uniform(-1.0, 1.0, shape).astype(np.float32)


def manifest_of(model):
    return [[k, list(v.shape), str(v.dtype).replace("torch.", "")] for k, v in model.state_dict().items()]


def load_synth(model, manifest):
    man = [(n, tuple(s), d) for n, s, d in manifest]
    state = W.synth_state_dict(man, SEED)
    model.load_state_dict({k: torch.from_numpy(v) for k, v in state.items()}, strict=True)
    model.eval()
    return state


def maxdiff(a, b):
    return float((a - b).abs().max())


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    sys.path.insert(0, str(REF / "rapid_doc/model/ocr"))
    from ppocrv6_pytorch.modeling.architectures.base_model import BaseModel

    arch = yaml.safe_load(open(REF / "rapid_doc/resources/arch_config.yaml"))
    summary = {}

    # ---------------- det ----------------
    det = BaseModel(arch["ch_PP-OCRv6_det_small"])
    man = manifest_of(det)
    (HERE / "manifest_ppocrv6_det.json").write_text(json.dumps(man))
    state = load_synth(det, man)
    summary["det_checksum"] = W.checksum(state)
    tstate = O.as_torch_state(state)
    for tag, shape in (("64x96", (1, 3, 64, 96)), ("b2_96x160", (2, 3, 96, 160))):
        x = make_input(shape, 100 + len(tag), "normal")
        xt = torch.from_numpy(x)
        with torch.no_grad():
            feats = det.backbone(xt)
            neck = det.neck(feats)
            maps = det.head(neck)["maps"]
            mine = O.det_forward(tstate, xt, return_all=True)
        d = max(maxdiff(maps, mine["maps"]), maxdiff(neck, mine["neck"]),
                *[maxdiff(a, b) for a, b in zip(feats, mine["feats"])])
        print(f"det {tag}: oracle-vs-reference max|diff| = {d:.3e}; maps range {float(maps.min()):.3f}..{float(maps.max()):.3f}")
        assert d < 2e-5, d
        np.savez_compressed(
            HERE / f"det_seed0_{tag}.npz", x=x, maps=maps.numpy(), neck=neck.numpy(),
            **{f"feat{i}": f.numpy() for i, f in enumerate(feats)})

    # ---------------- rec ----------------
    rec = BaseModel(arch["ch_PP-OCRv6_small_rec_infer"])
    man = manifest_of(rec)
    (HERE / "manifest_ppocrv6_rec.json").write_text(json.dumps(man))
    state = load_synth(rec, man)
    summary["rec_checksum"] = W.checksum(state)
    tstate = O.as_torch_state(state)
    for tag, shape in (("b2_w320", (2, 3, 48, 320)), ("b1_w96", (1, 3, 48, 96)), ("b3_w640", (3, 3, 48, 640))):
        x = make_input(shape, 200 + len(tag) + shape[3], "pm1")
        xt = torch.from_numpy(x)
        with torch.no_grad():
            bb = rec.backbone(xt)
            out = rec.head(bb)
            logits = out["ctc_logits"]
            neck = rec.head.encoder(bb)
            mine = O.rec_forward(tstate, xt, return_all=True)
        d = max(maxdiff(logits, mine["logits"]), maxdiff(bb, mine["backbone"]), maxdiff(neck, mine["neck"]))
        print(f"rec {tag}: oracle-vs-reference max|diff| = {d:.3e}; logits range {float(logits.min()):.3f}..{float(logits.max()):.3f}")
        assert d < 5e-5, d
        prob = torch.softmax(logits, dim=2)  # what the reference session returns (ocr/torch.py:186-187)
        p, idx = prob.max(dim=2)
        top2 = torch.topk(logits, 2, dim=2).values
        np.savez_compressed(
            HERE / f"rec_seed0_{tag}.npz", x=x, backbone=bb.numpy(), neck=neck.numpy(),
            idx=idx.numpy().astype(np.int32), prob=p.numpy(), top2gap=(top2[..., 0] - top2[..., 1]).numpy(),
            logits_sub=logits[:, :, ::61].contiguous().numpy(),
            logits_t0=logits[:, 0, :].contiguous().numpy())

    # ---------------- PPHGNetV2-B4 (layout backbone) ----------------
    sys.path.insert(0, str(REF / "rapid_doc/model/formula/rapid_formula_self"))
    from networks.backbones.rec_pphgnetv2 import PPHGNetV2_B4

    b4 = PPHGNetV2_B4(det=True)
    man = manifest_of(b4)
    (HERE / "manifest_pphgnetv2_b4.json").write_text(json.dumps(man))
    state = load_synth(b4, man)
    summary["b4_checksum"] = W.checksum(state)
    tstate = O.as_torch_state(state)
    for tag, shape in (("64x96", (1, 3, 64, 96)),):
        x = make_input(shape, 300, "unit")
        xt = torch.from_numpy(x)
        with torch.no_grad():
            feats = b4(xt)
            mine = O.pphgnetv2_features(tstate, xt)
        d = max(maxdiff(a, b) / max(1.0, float(a.abs().max())) for a, b in zip(feats, mine))
        print(f"b4 {tag}: oracle-vs-reference max rel diff = {d:.3e}; feat absmax {[float(f.abs().max()) for f in feats]}")
        assert d < 2e-5, d
        np.savez_compressed(HERE / f"b4_seed0_{tag}.npz", x=x, **{f"feat{i}": f.numpy() for i, f in enumerate(feats)})

    # ---------------- host box helpers (pure python in the reference; cv2 stubbed, it is not called by them) ----------
    import importlib.util
    import types
    sys.modules.setdefault("cv2", types.ModuleType("cv2"))
    spec = importlib.util.spec_from_file_location("ref_ocr_utils", REF / "rapid_doc/utils/ocr_utils.py")
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    for seed in range(5):
        rng = np.random.default_rng(1000 + seed)
        quads = []
        for _ in range(40):
            x0, y0 = rng.integers(0, 900), rng.integers(0, 60) * 14
            w, h = rng.integers(20, 400), rng.integers(12, 40)
            q = np.array([[x0, y0], [x0 + w, y0], [x0 + w, y0 + h], [x0, y0 + h]], dtype=np.float32)
            if rng.random() < 0.15:   # a tilted quad
                q[1, 1] += h * 1.5; q[2, 1] += h * 1.5
            quads.append(q)
        formulas = [{"bbox": [int(a), int(b), int(a + c), int(b + d)]} for a, b, c, d in
                    zip(rng.integers(0, 900, 12), rng.integers(0, 60, 12) * 14, rng.integers(20, 120, 12), rng.integers(12, 40, 12))]
        out = {
            "quads": [q.tolist() for q in quads], "formulas": formulas,
            "sorted": [np.asarray(b).tolist() for b in ref.sorted_boxes(np.array(quads))],
            "merged": [np.asarray(b).tolist() for b in ref.merge_det_boxes([q.copy() for q in quads])],
            "updated": [np.asarray(b).tolist() for b in ref.update_det_boxes([q.copy() for q in quads], formulas)],
            "is_angle": [bool(ref.calculate_is_angle(q)) for q in quads],
        }
        (HERE / f"boxes_seed{seed}.json").write_text(json.dumps(out))
        print(f"boxes seed {seed}: {len(quads)} quads -> merged {len(out['merged'])}, updated {len(out['updated'])}")

    (HERE / "summary.json").write_text(json.dumps(summary, indent=1))
    print(summary)


if __name__ == "__main__":
    main()
