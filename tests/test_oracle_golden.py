"""CPU: the oracle (oracle/nets.py) against golden vectors minted from the reference's own modules
(tests/golden/make_golden.py).  Pins both the restatement and the synthetic-weight generator."""
import json

import numpy as np
import pytest
import torch

from oracle import nets as O
from rapiddoc_amd import weights as W

TOL = 2e-5


def _state(golden_dir, name):
    man = W.load_manifest(golden_dir / f"manifest_{name}.json")
    st = W.synth_state_dict(man, 0)
    return st, O.as_torch_state(st)


def test_weight_generator_is_pinned(golden_dir):
    summ = json.loads((golden_dir / "summary.json").read_text())
    for name, key in (("ppocrv6_det", "det_checksum"), ("ppocrv6_rec", "rec_checksum"),
                      ("pphgnetv2_b4", "b4_checksum")):
        st, _ = _state(golden_dir, name)
        assert abs(W.checksum(st) - summ[key]) < 1e-6 * max(1.0, abs(summ[key]))


@pytest.mark.parametrize("tag", ["64x96", "b2_96x160"])
def test_det_oracle_matches_reference_golden(golden_dir, tag):
    _, st = _state(golden_dir, "ppocrv6_det")
    g = np.load(golden_dir / f"det_seed0_{tag}.npz")
    out = O.det_forward(st, torch.from_numpy(g["x"]), return_all=True)
    for i, f in enumerate(out["feats"]):
        assert np.abs(f.numpy() - g[f"feat{i}"]).max() < TOL
    assert np.abs(out["neck"].numpy() - g["neck"]).max() < TOL * 10
    assert np.abs(out["maps"].numpy() - g["maps"]).max() < TOL


@pytest.mark.parametrize("tag", ["b2_w320", "b1_w96", "b3_w640"])
def test_rec_oracle_matches_reference_golden(golden_dir, tag):
    _, st = _state(golden_dir, "ppocrv6_rec")
    g = np.load(golden_dir / f"rec_seed0_{tag}.npz")
    out = O.rec_forward(st, torch.from_numpy(g["x"]), return_all=True)
    assert np.abs(out["backbone"].numpy() - g["backbone"]).max() < TOL
    assert np.abs(out["neck"].numpy() - g["neck"]).max() < TOL * 5
    lg = out["logits"]
    assert np.abs(lg[:, :, ::61].numpy() - g["logits_sub"]).max() < 2e-4
    assert np.abs(lg[:, 0, :].numpy() - g["logits_t0"]).max() < 2e-4
    idx, p = O.ctc_greedy_stats(lg)
    safe = g["top2gap"] > 1e-3
    assert (idx.numpy() == g["idx"])[safe].all()
    assert np.abs(p.numpy() - g["prob"]).max() < 1e-5


def test_b4_oracle_matches_reference_golden(golden_dir):
    _, st = _state(golden_dir, "pphgnetv2_b4")
    g = np.load(golden_dir / "b4_seed0_64x96.npz")
    feats = O.pphgnetv2_features(st, torch.from_numpy(g["x"]))
    for i, f in enumerate(feats):
        assert np.abs(f.numpy() - g[f"feat{i}"]).max() < TOL


@pytest.mark.parametrize("tag", ["b2_c1_64x96", "b1_c3_96x64"])
def test_formula_encoder_oracle_matches_reference_golden(golden_dir, tag):
    _, st = _state(golden_dir, "pphgnetv2_b6_formula")
    g = np.load(golden_dir / f"b6_seed0_{tag}.npz")
    enc = O.formula_encoder_forward(st, torch.from_numpy(g["x"]))
    assert np.abs(enc.numpy() - g["enc"]).max() < TOL


@pytest.mark.parametrize("tag,max_new", [("dec_a", 16), ("dec_b", 24)])
def test_formula_decoder_oracle_matches_reference_golden(golden_dir, tag, max_new):
    """oracle/formula.py vs token ids produced by the reference PPFormulaNet_Head (generate_export)."""
    from oracle import formula as OF
    g = np.load(golden_dir / f"formula_seed0_{tag}.npz")
    st = W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_ppformulanet_head_{tag}.json"), 0)
    st["head.decoder.lm_head.weight"][2] *= float(g["eos_gain"])
    ids = OF.formula_decode(O.as_torch_state(st), torch.from_numpy(g["enc"]), max_new)
    assert ids.shape == g["ids"].shape and (ids.numpy() == g["ids"]).all()
