"""Build recipe for librapiddoc_mi355.so (hipcc, gfx950 only, in-tree so the .so travels to the GPU box)."""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OUT = PKG / "librapiddoc_mi355.so"
SOURCES = ["kernels_conv.hip", "kernels_conv_h3.hip", "kernels_conv_direct_h3.hip", "kernels_conv3x3_h1.hip", "kernels_stem34.hip", "kernels_rtdetr.hip", "kernels_conv_stream_h3.hip", "kernels_gemm_h3_dma.hip", "kernels_gemm_h1.hip", "kernels_misc.hip", "kernels_dw_lds.hip", "kernels_stem_fused.hip", "kernels_image.hip", "kernels_dbpost.hip", "kernels_attention_h3.hip", "kernels_ctc.hip", "kernels_mixer.hip", "kernels_mixer_h3.hip", "kernels_mixer_ws.hip", "kernels_mixer_res.hip", "formula_decoder.hip", "engine.cpp", "models.cpp", "api.cpp", "db_postprocess.cpp", "layout_postprocess.cpp", "polygon_ops.cpp", "rec_chunks.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-x", "hip"]
# Per-file code-generation flags (tests/test_isa_resources.py and tools/isa_mix.py compile with the same ones: `extra_flags_for`).
# kernels_mixer_ws.hip: hipcc -O3 SLP-packs adjacent scalar fp32 chains into v_pk_*_f32, and packed fp32 VALU does NOT issue under another
# wavefront's MFMAs on gfx950 while plain VALU does (profiles/r4_probe_mfma_valu_wall.txt): the ws mixer's GELU is written scalar
# (RD_GELU_SCALAR) and must stay scalar for its GELU-in-the-middle step order to pay.
# kernels_gemm_h1.hip: the activation split next to the MFMAs must stay on plain VALU (v_cvt_pk_f16_f32 + v_fma_mix), same reason.
FILE_FLAGS = {"kernels_mixer_ws.hip": ["-fno-slp-vectorize", "-DRD_GELU_SCALAR"], "kernels_gemm_h1.hip": ["-fno-slp-vectorize"],
              "kernels_stem_fused.hip": ["-fno-slp-vectorize"]}


def extra_flags_for(src: str) -> list:
    return list(FILE_FLAGS.get(Path(src).name, []))


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    raise RuntimeError("hipcc not found (ROCm 7.x required)")


def needs_build() -> bool:
    if not OUT.exists():
        return True
    t = OUT.stat().st_mtime
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "rapiddoc_mi355.h", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force: bool = False, verbose: bool = True, extra_flags: tuple = (), out: Path = None) -> Path:
    """`extra_flags` / `out`: developer A/B builds of the same sources into another file (tools/ab_variants.py); the product is
    always the default call."""
    variant = out is not None
    if not variant and not force and not needs_build():
        return OUT
    hipcc = _hipcc()
    objdir = PKG / "build" / (Path(out).stem if variant else "")
    objdir.mkdir(parents=True, exist_ok=True)
    OUT_ = Path(out) if variant else OUT

    def compile_one(src: str) -> Path:
        obj = objdir / (src + ".o")
        cmd = [hipcc, *FLAGS, *extra_flags_for(src), *extra_flags, "-c", str(CSRC / src), "-o", str(obj)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed for {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    tmp = OUT_.with_name(OUT_.name + f".tmp{os.getpid()}")      # link aside, then rename: readers never see a partial file
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", *map(str, objs), "-o", str(tmp)]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        tmp.unlink(missing_ok=True)
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    os.replace(tmp, OUT_)
    if verbose:
        print(f"built {OUT_} ({OUT_.stat().st_size/1e6:.1f} MB)")
    return OUT_


if __name__ == "__main__":
    build(force="--force" in sys.argv)
