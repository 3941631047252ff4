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


def formula_long_case(golden_dir):
    """(state dict, encoder states, golden npz) of the 300-token decode fixture (tests/golden/make_golden_r3.py): the encoder
    states are regenerated from their seed (2.4 MB of noise is not worth committing) and checked against the stored crc."""
    import zlib
    g = np.load(golden_dir / "formula_seed0_dec_long.npz")
    st = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppformulanet_head_dec_long.json"), 0)
    st["head.decoder.lm_head.weight"] = st["head.decoder.lm_head.weight"] * np.float32(g["logit_gain"])
    st["head.decoder.lm_head.weight"][2] *= np.float32(g["eos_gain"])
    enc = (np.random.default_rng(int(g["enc_seed"])).standard_normal(tuple(int(v) for v in g["enc_shape"])) * 3.0).astype(np.float32)
    assert zlib.crc32(enc.tobytes()) == int(g["enc_crc32"]), "numpy's Generator stream changed: re-mint the fixture"
    return st, enc, g


def live_steps(ids):
    """[B, L-1] bool: step t produced a real token (the sequence had not emitted EOS before it)."""
    ended = np.cumsum(ids[:, 1:] == 2, axis=1) - (ids[:, 1:] == 2)      # EOS seen strictly before this step
    return ended == 0


def test_formula_oracle_long_decode_teacher_forced(golden_dir):
    """The oracle fed the reference's own 300-token ids reproduces, step by step, the token the reference chose next
    (one teacher-forced pass; the step-by-step greedy run of the oracle equalled the reference when the fixture was minted)."""
    from oracle import formula as OF
    st, enc, g = formula_long_case(golden_dir)
    ids = g["ids"]
    assert ids.shape == (2, 301)
    with torch.no_grad():
        lg = OF.teacher_forced_logits(O.as_torch_state(st), torch.from_numpy(enc), torch.from_numpy(ids))
    live = live_steps(ids)
    top2 = torch.topk(lg, 2, dim=-1).values
    gaps = (top2[..., 0] - top2[..., 1]).numpy()
    assert np.abs(gaps - g["top2gap"])[live].max() < 2e-3
    assert (lg.argmax(-1).numpy() == ids[:, 1:])[live & (gaps > 1e-2)].all()
    assert live[0].all() and live[1].sum() == 76          # sequence 0 runs to the limit, sequence 1 emits EOS as its 76th token
    assert ids[1, 76] == 2 and (ids[1, 77:] == 1).all()    # and is padded afterwards
