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
