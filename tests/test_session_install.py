"""CPU: `install_into_rapidocr` replaces rapidocr's torch session class where the reference's own patch does
(rapid_doc/model/ocr/ocr_patch.py:95-105), for both module layouts of the pinned rapidocr range; a session is then built per task."""
import sys
import types

import pytest

from rapiddoc_amd import session as S


def _fake_rapidocr(monkeypatch, new_layout: bool):
    mods = {"rapidocr": types.ModuleType("rapidocr"), "rapidocr.inference_engine": types.ModuleType("rapidocr.inference_engine")}
    for m in mods.values():
        m.__path__ = []
    if new_layout:
        pkg = types.ModuleType("rapidocr.inference_engine.pytorch")
        pkg.__path__ = []
        main = types.ModuleType("rapidocr.inference_engine.pytorch.main")
        pkg.TorchInferSession = main.TorchInferSession = object
        mods["rapidocr.inference_engine.pytorch"], mods["rapidocr.inference_engine.pytorch.main"] = pkg, main
    else:
        old = types.ModuleType("rapidocr.inference_engine.torch")
        old.TorchInferSession = object
        mods["rapidocr.inference_engine.torch"] = old
    for k, v in mods.items():
        monkeypatch.setitem(sys.modules, k, v)
    return mods


@pytest.mark.parametrize("new_layout", [True, False])
def test_install_replaces_the_session_class_and_dispatches_by_task(monkeypatch, new_layout):
    mods = _fake_rapidocr(monkeypatch, new_layout)
    built = []
    monkeypatch.setattr(S.Mi355DetSession, "from_cfg", classmethod(lambda cls, cfg: built.append(("det", cfg)) or "det-session"))
    monkeypatch.setattr(S.Mi355RecSession, "from_cfg", classmethod(lambda cls, cfg: built.append(("rec", cfg)) or "rec-session"))
    S.install_into_rapidocr()
    holders = ([mods["rapidocr.inference_engine.pytorch"], mods["rapidocr.inference_engine.pytorch.main"]] if new_layout
               else [mods["rapidocr.inference_engine.torch"]])
    assert all(h.TorchInferSession is not object for h in holders) and len({h.TorchInferSession for h in holders}) == 1
    cls = holders[0].TorchInferSession
    assert cls({"task_type": "TaskType.DET", "model_path": "x.safetensors"}) == "det-session"
    assert cls(types.SimpleNamespace(task_type="rec", model_path="y.safetensors")) == "rec-session"
    assert cls({"model_path": "/w/ch_PP-OCRv6_det_small.safetensors"}) == "det-session"          # no task_type: the file's stem decides
    assert cls({"model_path": "/w/ch_PP-OCRv6_rec_small.safetensors"}) == "rec-session"
    assert [b[0] for b in built] == ["det", "rec", "det", "rec"]


def test_install_without_rapidocr_fails_loudly(monkeypatch):
    for k in [k for k in sys.modules if k == "rapidocr" or k.startswith("rapidocr.")]:
        monkeypatch.delitem(sys.modules, k)
    with pytest.raises(ModuleNotFoundError):
        S.install_into_rapidocr()


def test_safetensors_writer_and_reader_agree_with_the_safetensors_package(golden_dir):
    """The byte image the tests write for `from_cfg` (weights.to_safetensors_bytes) is what the installed `safetensors` package - the
    one the reference loads its weights with (torch.py:93-103) - reads and writes: same tensors both ways."""
    import numpy as np
    safetensors_numpy = __import__("pytest").importorskip("safetensors.numpy")
    from rapiddoc_amd import weights as W
    state = W.synth_state_dict(W.load_manifest(golden_dir / "manifest_ppocrv6_det.json"), 0)
    state = {"model." + k: v for k, v in list(state.items())[:40]}
    state["i64"] = np.arange(5, dtype=np.int64)
    state["f16"] = np.linspace(-1, 1, 7).astype(np.float16)
    blob = W.to_safetensors_bytes(state)
    theirs = safetensors_numpy.load(blob)
    assert set(theirs) == set(state)
    for k, v in state.items():
        assert theirs[k].dtype == v.dtype and theirs[k].shape == v.shape and np.array_equal(theirs[k], v), k
    ours = W.from_safetensors_bytes(safetensors_numpy.save(state))
    assert set(ours) == set(state)
    for k, v in state.items():
        assert ours[k].dtype == v.dtype and np.array_equal(ours[k], v), k
