"""CPU: no hot-path kernel may spill registers (commit 27b771f put 23 spilled VGPRs and a 96-byte private segment
into the second-largest GEMM and nothing noticed).  Every HIP source is compiled to gfx950 assembly (`hipcc -S`, no GPU needed)
and the `.amdhsa` metadata of every kernel the engine dispatches on the hot path is checked: zero spilled VGPRs, no private
(scratch) segment.  Kernels reachable only through the developer / ablation entry points are listed explicitly."""
import re
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "rapiddoc_amd" / "csrc"

# kernels allowed to spill: not dispatched by the engine (microbenchmark ablations, opt-in experiments), with the reason
ALLOWED = {
    # template arguments <C, GATED, KEEPX, ABL, PF>
    r"lc_mixer_ws_kernelILi192ELb[01]ELb1E": "KEEPX instantiations (fp32 X tile kept in registers): debug entry only, C = 192 spills by design",
    r"lc_mixer_ws_kernelILi192ELb[01]ELb[01]ELi(1|2|4|8|12|13|16|17|25|27|28|29|64|72|88)ELb[01]E": "ablation instantiations of the ws mixer (tools/microbench.py)",
    r"gemm_h3_dma16_kernelILi(1|2|4|8|10)E": "ablation instantiations of the 16-wavefront GEMM (tools/mb_gemm_abl.py, RD_GEMM_DBG)",
    r"dwconv_tiled_kernelILi3ELi3ELi1ELi8ELi[1-7]E": "ablation instantiations of the depthwise 3x3 (RD_DW_DBG)",
    r"gemm_h1_kernelILi(1|2|8|16|32|34|64)E": "ablation instantiations of the single-accumulator GEMM (tools/mb_gemm_h1.py, RD_GEMM1_DBG)",
    r"gemm_h1_kernelILi0ELb1E": "the single-accumulator GEMM with its DMA pieces between the MFMA groups (RD_GEMM1_IL=1, A/B only): 8 spilled "
                                "registers at the tile switch",
    r"lc_mixer_h3_kernel.*Li192E": "round-1 C = 192 mixer: superseded by the ws kernel, kept for A/B (RD_MIXER_WS=0)",
    r"db_(regions|finish)_kernel": "no spill: local arrays (4-corner boxes, hull scratch) of the geometry code shared with the host path "
                                   "(csrc/db_geom.h), indexed at run time; one thread per text-line candidate, ~50 candidates per page",
    r"ctc_collapse_kernel": "no spill: 32 bytes of CALL STACK for the recursive numpy-pairwise-sum restatement (one thread per line)",
}


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    pytest.skip("hipcc not available")


def _kernel_table(src: Path):
    from rapiddoc_amd.build import extra_flags_for          # the per-file flags the library is built with
    out = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *extra_flags_for(src.name), "-x", "hip", "-S", "--cuda-device-only",
                          f"-I{CSRC}", str(src), "-o", "-"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    rows = []
    for blk in re.split(r"\n\s+- \.agpr_count:", out.stdout)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        get = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))
        rows.append((name.group(1), get("vgpr_count"), get("vgpr_spill_count"), get("private_segment_fixed_size")))
    return rows


@pytest.fixture(scope="module")
def tables():
    srcs = sorted(CSRC.glob("*.hip"))
    with ThreadPoolExecutor(max_workers=8) as ex:
        return dict(zip([s.name for s in srcs], ex.map(_kernel_table, srcs)))


def test_every_hip_source_has_kernels(tables):
    assert len(tables) >= 12
    assert sum(len(v) for v in tables.values()) >= 60


def test_no_hot_kernel_spills(tables):
    bad = []
    for src, rows in tables.items():
        for name, vgprs, spills, private in rows:
            if "ctc_collapse_kernel" in name or "db_regions_kernel" in name or "db_finish_kernel" in name:
                assert spills == 0
            if any(re.search(pat, name) for pat in ALLOWED):
                continue
            if spills or private:
                bad.append((src, name, vgprs, spills, private))
    assert not bad, "kernels with spilled VGPRs / scratch:\n" + "\n".join(map(str, bad))


def test_known_register_budgets(tables):
    """The two LDS-DMA GEMMs: the 8-wavefront kernel must fit the 256 registers of two wavefronts per SIMD, the 16-wavefront one
    the 128 of four."""
    rows = {n: (v, s, p) for n, v, s, p in tables["kernels_gemm_h3_dma.hip"]}
    k8 = next(v for n, v in rows.items() if "gemm_h3_dma_kernelILb0E" in n)
    k16 = next(v for n, v in rows.items() if "gemm_h3_dma16_kernelILi0E" in n)
    assert k8[0] <= 256 and k8[1:] == (0, 0)
    assert k16[0] <= 128 and k16[1:] == (0, 0)
    # the single-accumulator GEMM lives on TWO workgroups of four wavefronts per CU: 256 registers, nothing spilled
    h1 = {n: (v, s, p) for n, v, s, p in tables["kernels_gemm_h1.hip"]}
    k1 = next(v for n, v in h1.items() if "gemm_h1_kernelILi0ELb0E" in n)
    assert k1[0] <= 256 and k1[1:] == (0, 0)
    # the LDS-DMA-staged depthwise 3x3 lives on occupancy (nothing persistent: one workgroup's DMA under another's arithmetic): four
    # workgroups per CU need <= 128 registers, five <= 96 (left alone the compiler sank every FMA behind the loop: 138, three per SIMD)
    dw_all = {n: (v, s_, p_) for n, v, s_, p_ in tables["kernels_dw_lds.hip"]}
    dw = {n: v for n, v in dw_all.items() if "dwconv3x3_lds_kernel" in n}
    kxk = {n: v for n, v in dw_all.items() if "dwconv_kxk_lds_kernel" in n}
    assert len(dw) == 5 and len(kxk) == 2 and all(v[0] <= 104 and v[1:] == (0, 0) for v in kxk.values()), kxk
    for n, (v, s_, p_) in dw.items():
        assert v <= 104 and (s_, p_) == (0, 0), (n, v, s_, p_)
    assert next(v for n, v in dw.items() if "ILi16ELi6E" in n)[0] <= 96
    # the fused projection + attention launches of the formula decode: 256 threads, no spill
    fd = {n: (v, s_, p_) for n, v, s_, p_ in tables["formula_decoder.hip"]}
    fused = [v for n, v in fd.items() if "dec_attn_fused_kernel" in n]
    assert len(fused) == 2 and all(v[0] <= 160 and v[1:] == (0, 0) for v in fused), fused


def test_inline_asm_register_loads_are_not_touched_in_flight():
    """The prefetching ws mixer loads its X registers with inline assembly (the compiler's own wait insertion cannot express "loads
    landed, younger stores still in flight").  The compiler therefore does not know those registers are in flight: between such a
    load and the `s_waitcnt vmcnt` that ends its window no instruction may name one of them (a register copy at a block boundary
    would silently move garbage), and no block boundary may fall into a window.  Checked on the ISA of every instantiation."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_inflight", ROOT / "tools" / "check_inflight.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from rapiddoc_amd.build import extra_flags_for
    out = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *extra_flags_for("kernels_mixer_ws.hip"), "-x", "hip", "-S",
                          "--cuda-device-only", f"-I{CSRC}", str(CSRC / "kernels_mixer_ws.hip"), "-o", "-"], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.count("global_load_dwordx4") >= 48          # the PF instantiations are there
    hazards = mod.check(out.stdout, ["ELb1EEEvNS_11MixerParams"])      # <..., PF = true>: the instantiations with inline-asm loads
    assert not hazards, hazards[:10]


def test_no_serialised_load_runs_in_the_kernels_fixed_for_it():
    """`x = ok ? p[i] : 0` compiles to `global_load ; s_waitcnt vmcnt(0) ; v_cndmask` per element: a run of conditional loads is a run
    of serial memory round trips.  Round 3 found and removed such runs in the fused stem (image-patch prefetch), the
    skinny GEMM of the formula decode steps, the streaming small-K conv and the MFMA attention's staging; this keeps them out: in
    those kernels no four loads in a row may each be followed by a full vmcnt(0) wait before the next load is issued."""
    bad = []
    for fname, kernels in (("kernels_stem_fused.hip", ("stem_fused_kernel",)), ("kernels_conv.hip", ("skinny2_gemm_kernel",)),
                           ("kernels_conv_stream_h3.hip", ("conv_stream_h3_kernel",)), ("kernels_attention_h3.hip", ("attention_h3_kernel",))):
        out = subprocess.run([_hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-x", "hip", "-S", "--cuda-device-only",
                              f"-I{CSRC}", str(CSRC / fname), "-o", "-"], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr[-2000:]
        seen = 0
        for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\n\s*\.end_amdhsa_kernel", out.stdout, re.S | re.M):
            name, body = m.group(1), m.group(2)
            if not any(k in name for k in kernels):
                continue
            seen += 1
            ev = []
            for line in body.split("\n"):
                ls = line.strip()
                if ls.startswith(("global_load_dword", "buffer_load_dword")) and "lds" not in ls:
                    ev.append("L")
                elif ls.startswith("s_waitcnt") and "vmcnt(0)" in ls:
                    ev.append("w")
                elif ls.startswith(("v_mfma", "s_barrier", "global_store", "ds_write", "ds_read")):
                    ev.append("x")
            runs = re.findall(r"(?:Lw){4,}", "".join(ev))
            if runs:
                bad.append((name, [len(r) // 2 for r in runs]))
        assert seen, fname
    # (the 8-row / 4-column / 4-load instantiation of the skinny GEMM keeps one run of five: its 64 weight registers plus 32
    #  accumulators leave the compiler no room to keep all sixteen loads in flight)
    bad = [b for b in bad if not ("skinny2_gemm_kernelILi8ELi4ELi4E" in b[0] and max(b[1]) <= 5)]
    assert not bad, bad


def test_inflight_checker_flags_a_hazard():
    """The checker itself: a register copy of an in-flight register and a block boundary inside a load window must be reported, a clean
    window must not (so that the test above cannot pass vacuously)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_inflight", ROOT / "tools" / "check_inflight.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kernel = lambda body: "k1:                                     ; @k1\n" + body + "\n\t.end_amdhsa_kernel\n"
    clean = kernel("\tglobal_load_dwordx4 v[8:11], v[2:3], off\n\tv_add_u32_e32 v1, v0, v0\n\ts_waitcnt vmcnt(0)\n\tv_mov_b32_e32 v20, v8")
    assert mod.check(clean, ["k1"]) == []
    copied = kernel("\tglobal_load_dwordx4 v[8:11], v[2:3], off\n\tv_mov_b32_e32 v20, v9\n\ts_waitcnt vmcnt(0)")
    assert len(mod.check(copied, ["k1"])) == 1
    split = kernel("\tglobal_load_dwordx4 v[8:11], v[2:3], off\n.LBB0_3:\n\ts_waitcnt vmcnt(0)")
    assert any("block boundary" in h[2] for h in mod.check(split, ["k1"]))


def test_isa_mix_finds_the_inner_loops_of_the_dominant_kernels():
    """tools/isa_mix.py (the static issue budget quoted in DESIGN.md s3c): the prefetching ws mixer's tile loop holds 72 MFMAs of the
    16x16x32 shape and no vector-memory stores or plain loads (only LDS-DMA requests), the 16-wavefront GEMM's K loop 12 of the 32x32x16."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("isa_mix", ROOT / "tools" / "isa_mix.py")
    mix = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mix)
    _hipcc()
    for src, pat, shape, n_mfma in (("kernels_mixer_ws.hip", r"lc_mixer_ws_kernelILi192ELb0ELb0ELi0ELb1E", "f32_16x16x32_f16", 72),
                                    ("kernels_gemm_h3_dma.hip", r"gemm_h3_dma16_kernelILi0E", "f32_32x32x16_f16", 12),
                                    ("kernels_gemm_h1.hip", r"gemm_h1_kernelILi0ELb0E", "f32_32x32x16_f16", 48)):
        bodies, _meta = mix.kernel_bodies(mix.asm_of(src))
        name = min((n for n in bodies if re.search(pat, n)), key=len)
        loops = mix.innermost_mfma_loops(bodies[name])
        assert len(loops) == 1
        c = mix.classify(bodies[name][loops[0][0]: loops[0][1] + 1])
        assert c["mfma"] == {shape: n_mfma}
        assert c["vmem_ld"] == 0 and c["vmem_st"] == 0 and c["vmem_ld_lds"] > 0 and c["barrier"] >= 1
