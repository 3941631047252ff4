"""CPU: the C-ABI library builds, loads and exports every symbol include/rapiddoc_mi355.h declares; the
product path fails loudly (no CPU fallback) when there is no GPU."""
import ctypes as C
import os
import re
from pathlib import Path

import pytest
import torch

ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def lib():
    from rapiddoc_amd import build as rd_build
    rd_build.build(verbose=False)
    from rapiddoc_amd import _lib
    return _lib.load()


def test_header_symbols_are_exported(lib):
    header = (ROOT / "include" / "rapiddoc_mi355.h").read_text()
    declared = set(re.findall(r"\b(rd_[a-z0-9_]+)\s*\(", header))
    declared -= {"rd_handle", "rd_crop_desc", "rd_text_box", "rd_layout_post_cfg"}
    assert len(declared) >= 14
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    from rapiddoc_amd import _lib
    assert declared == set(_lib.SYMBOLS), (declared ^ set(_lib.SYMBOLS))


def test_version_and_seq_len(lib):
    assert b"gfx950" in lib.rd_version()
    from rapiddoc_amd import ocr_host
    for w in (15, 16, 17, 96, 320, 321, 327, 1157, 2112):
        assert lib.rd_rec_seq_len(w) == ocr_host.rec_seq_len(w)
    assert lib.rd_rec_seq_len(320) == 40


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_no_gpu_fails_loudly(lib):
    h = lib.rd_create(0, b"ppocrv6_det")
    assert not h
    assert b"no HIP device" in lib.rd_create_error()
    from rapiddoc_amd.engine import EngineError, RdEngine
    with pytest.raises(EngineError):
        RdEngine("ppocrv6_det")


def test_crop_desc_layout_matches_header():
    from rapiddoc_amd.pipeline import CROP_DTYPE, CropDesc
    assert C.sizeof(CropDesc) == 60 == CROP_DTYPE.itemsize  # 15 x 4-byte fields, see rd_crop_desc


def test_product_package_never_imports_oracle():
    for py in (ROOT / "rapiddoc_amd").rglob("*.py"):
        src = py.read_text()
        assert "import oracle" not in src and "from oracle" not in src, py


def test_importing_the_package_defaults_the_hardware_queue_count_and_respects_an_explicit_one():
    """`import rapiddoc_amd` gives every HIP stream its own hardware queue unless the environment says otherwise (with
    the runtime's default of four, streams that share a queue serialise - 7-15 % of a step)."""
    import subprocess
    import sys
    root = str(Path(__file__).resolve().parents[1])
    code = "import os, sys; sys.path.insert(0, %r); import rapiddoc_amd; print(os.environ['GPU_MAX_HW_QUEUES'])" % root
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "16"
    env["GPU_MAX_HW_QUEUES"] = "4"
    assert subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True).stdout.strip() == "4"
    # the override is a side effect on the embedding process: it is logged once (INFO, logger "rapiddoc_amd") and can be asked for
    code2 = ("import logging, os, sys; logging.basicConfig(level=logging.INFO, stream=sys.stdout, format='%%(name)s|%%(message)s'); "
             "sys.path.insert(0, %r); import rapiddoc_amd; print(rapiddoc_amd.HW_QUEUES_SET_BY_IMPORT)" % root)
    del env["GPU_MAX_HW_QUEUES"]
    out = subprocess.run([sys.executable, "-c", code2], env=env, capture_output=True, text=True).stdout
    assert out.count("rapiddoc_amd|rapiddoc_amd: GPU_MAX_HW_QUEUES was unset") == 1 and out.strip().endswith("True")
    env["GPU_MAX_HW_QUEUES"] = "8"
    out = subprocess.run([sys.executable, "-c", code2], env=env, capture_output=True, text=True).stdout
    assert "GPU_MAX_HW_QUEUES" not in out and out.strip().endswith("False")
