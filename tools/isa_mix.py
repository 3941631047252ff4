#!/usr/bin/env python3
"""Static instruction mix and register / LDS budget of the hot kernels, from `hipcc -S` (no GPU needed).

    python tools/isa_mix.py > profiles/r3_isa_mix.txt

For every kernel the bench's per-kernel table names (profiles/r3_bench_per_kernel.csv, top of the list) the gfx950 assembly of the
instantiation the engine dispatches is split into instruction classes: MFMA (by shape), LDS reads / writes (with bytes), vector memory
loads / stores (LDS-DMA loads apart), VALU (packed and transcendental apart), SALU, waits and barriers - whole-kernel static counts,
i.e. loop bodies count once however often they run, and code on both sides of a branch counts whether or not it runs - and, apart, the same classes for every innermost loop that holds MFMAs (a backward
branch to an earlier label), where the per-MFMA ratios mean something: KB of LDS reads per wavefront and VALU issue slots per MFMA (by
the additive model of DESIGN.md s3c a SIMD's VALU cycles add to its MFMA cycles; a 16x16x32 MFMA issues in 16 cycles, a 32x32x16 in 32,
a plain or packed VALU instruction in 4, a transcendental in 16)."""
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "rapiddoc_amd" / "csrc"

# (label of the per-kernel table, source, regex over the mangled name of the dispatched instantiation)
HOT = [
    ("lc_mixer_ws_kernel<192> (prefetching form)", "kernels_mixer_ws.hip", r"lc_mixer_ws_kernelILi192ELb0ELb0ELi384ELb1E"),
    ("gemm_h3_dma16_kernel", "kernels_gemm_h3_dma.hip", r"gemm_h3_dma16_kernelILi0E"),
    ("gemm_h3_dma_kernel", "kernels_gemm_h3_dma.hip", r"gemm_h3_dma_kernelILb0E"),
    ("dwconv3x3 (tiled, 8 wide)", "kernels_misc.hip", r"dwconv_tiled_kernelILi3ELi3ELi1ELi8ELi0E"),
    ("lc_mixer_res_kernel<96>", "kernels_mixer_res.hip", r"lc_mixer_res_kernelILi96E"),
    ("stem_fused_kernel<48>", "kernels_stem_fused.hip", r"stem_fused_kernelILi48E"),
    ("conv_direct_h3_kernel", "kernels_conv_direct_h3.hip", r"conv_direct_h3_kernel"),
    ("conv_igemm_h3_kernel<256x64>", "kernels_conv_h3.hip", r"conv_igemm_h3_kernelILi256ELi64E"),
    ("ctc_head_h3_kernel", "kernels_ctc.hip", r"ctc_head_h3_kernelILi0E"),
    ("conv_stream_h3_kernel", "kernels_conv_stream_h3.hip", r"conv_stream_h3_kernel"),
    ("attention_h3_kernel<15>", "kernels_attention_h3.hip", r"attention_h3_kernelILi15E"),
    ("se_fc_kernel", "kernels_misc.hip", r"se_fc_kernel"),
]
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
LDS_BYTES = {"b8": 1, "u8": 1, "i8": 1, "b16": 2, "u16": 2, "i16": 2, "b32": 4, "b64": 8, "b96": 12, "b128": 16}


def hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and Path(cand).exists():
            return cand
    sys.exit("hipcc not found")


def asm_of(src: str) -> str:
    sys.path.insert(0, str(ROOT))
    from rapiddoc_amd.build import extra_flags_for          # the per-file flags the library is built with
    out = subprocess.run([hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", *extra_flags_for(src), "-x", "hip", "-S", "--cuda-device-only",
                          f"-I{CSRC}", str(CSRC / src), "-o", "-"], capture_output=True, text=True)
    if out.returncode != 0:
        sys.exit(out.stderr[-3000:])
    return out.stdout


def kernel_bodies(asm: str):
    """{mangled name: [instruction lines]} and {mangled name: metadata block}."""
    bodies, cur, name = {}, None, None
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):\s*(;.*)?$", line)
        if m:
            name, cur = m.group(1), []
            bodies[name] = cur
            continue
        if cur is not None:
            t = line.strip()
            if t.startswith(".end_amdhsa_kernel") or t.startswith(".section") or t.startswith(".Lfunc_end"):
                cur = None
                continue
            if re.match(r"^\.LBB\w+:", t):
                cur.append("@" + t.split(":")[0])                 # a branch target inside the kernel
            elif t and not t.startswith((".", ";", "//")) and not t.endswith(":"):
                cur.append(t.split(";")[0].strip())
    meta = {}
    for blk in re.split(r"\n\s+- \.agpr_count:", asm)[1:]:
        n = re.search(r"\.name:\s+(\S+)", blk)
        if n:
            meta[n.group(1)] = "\n  - .agpr_count:" + blk
    return bodies, meta


def lds_bytes(op: str) -> int:
    n = 2 if "read2" in op or "write2" in op else 1
    for suf, b in LDS_BYTES.items():
        if op.endswith("_" + suf) or f"_{suf}_" in op:
            return n * b
    return 0


def innermost_mfma_loops(lines):
    """[(first, last)] index ranges of the loops (a backward branch to an earlier label) that hold MFMAs and contain no smaller loop
    that does."""
    where = {ln[1:]: i for i, ln in enumerate(lines) if ln.startswith("@")}
    loops = []
    for i, ln in enumerate(lines):
        m = re.match(r"^s_c?branch\w*\s+(\.LBB\w+)", ln)
        if m and m.group(1) in where and where[m.group(1)] < i:
            j = where[m.group(1)]
            if any(x.startswith("v_mfma") for x in lines[j:i]):
                loops.append((j, i))
    return [a for a in loops if not any(b != a and a[0] <= b[0] and b[1] <= a[1] for b in loops)]


def classify(lines):
    lines = [ln for ln in lines if not ln.startswith("@")]
    c = {"mfma": {}, "lds_r": 0, "lds_r_bytes": 0, "lds_w": 0, "lds_w_bytes": 0, "vmem_ld": 0, "vmem_ld_lds": 0, "vmem_st": 0,
         "valu": 0, "valu_pk": 0, "valu_trans": 0, "salu": 0, "smem": 0, "waitcnt": 0, "barrier": 0, "readlane": 0, "other": 0, "total": len(lines)}
    for ln in lines:
        op = ln.split()[0]
        if op.startswith("v_mfma") or op.startswith("v_smfmac"):
            shape = re.sub(r"^v_s?mfmac?_", "", op)
            c["mfma"][shape] = c["mfma"].get(shape, 0) + 1
        elif op.startswith("ds_"):
            if "read" in op or "load" in op or "bpermute" in op or "swizzle" in op:
                c["lds_r"] += 1
                c["lds_r_bytes"] += lds_bytes(op)
            else:
                c["lds_w"] += 1
                c["lds_w_bytes"] += lds_bytes(op)
        elif op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
            if " lds" in ln or op.startswith("global_load_lds"):
                c["vmem_ld_lds"] += 1
            else:
                c["vmem_ld"] += 1
        elif op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store", "global_atomic", "buffer_atomic")):
            c["vmem_st"] += 1
        elif op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
            c["readlane"] += 1
        elif op.startswith("v_"):
            if op.startswith(TRANS):
                c["valu_trans"] += 1
            elif op.startswith("v_pk_"):
                c["valu_pk"] += 1
            else:
                c["valu"] += 1
        elif op == "s_waitcnt" or op.startswith("s_wait"):
            c["waitcnt"] += 1
        elif op.startswith("s_barrier"):
            c["barrier"] += 1
        elif op.startswith(("s_load", "s_buffer_load")):
            c["smem"] += 1
        elif op.startswith("s_"):
            c["salu"] += 1
        else:
            c["other"] += 1
    return c


def main():
    srcs = sorted({s for _l, s, _r in HOT})
    with ThreadPoolExecutor(max_workers=8) as ex:
        asm = dict(zip(srcs, ex.map(asm_of, srcs)))
    parsed = {s: kernel_bodies(a) for s, a in asm.items()}
    print("Static instruction mix of the hot kernels (gfx950, hipcc -O3; whole-kernel counts: a loop body counts once)")
    print("columns: VGPR/AGPR/SGPR, spilled VGPRs, LDS bytes (static), then instruction classes\n")
    for label, src, pat in HOT:
        bodies, meta = parsed[src]
        names = [n for n in bodies if re.search(pat, n)]
        if not names:
            print(f"{label}: no instantiation matches {pat}")
            continue
        name = min(names, key=len)
        m = meta.get(name, "")
        get = lambda k: (re.search(rf"\.{k}:\s+(\d+)", m) or [None, "?"])[1]          # noqa: E731
        c = classify(bodies[name])
        n_mfma = sum(c["mfma"].values())
        print(f"{label}\n  {name}")
        print(f"  registers  vgpr {get('vgpr_count')}  agpr {get('agpr_count')}  sgpr {get('sgpr_count')}  spilled vgpr {get('vgpr_spill_count')}"
              f"  scratch {get('private_segment_fixed_size')} B  static LDS {get('group_segment_fixed_size')} B  wavefronts/workgroup "
              f"{int(get('max_flat_workgroup_size')) // 64 if get('max_flat_workgroup_size') != '?' else '?'}")
        print(f"  instructions {c['total']}:  MFMA {n_mfma} {dict(sorted(c['mfma'].items()))}")
        print(f"    LDS reads {c['lds_r']} ({c['lds_r_bytes']} B/lane)  LDS writes {c['lds_w']} ({c['lds_w_bytes']} B/lane)  "
              f"vmem loads {c['vmem_ld']} (+{c['vmem_ld_lds']} LDS-DMA)  vmem stores {c['vmem_st']}")
        print(f"    VALU {c['valu']}  packed {c['valu_pk']}  transcendental {c['valu_trans']}  lane moves {c['readlane']}  SALU {c['salu']}  "
              f"scalar loads {c['smem']}  waits {c['waitcnt']}  barriers {c['barrier']}  other {c['other']}")
        for a, b in innermost_mfma_loops(bodies[name]):
            lc = classify(bodies[name][a:b + 1])
            lm = sum(lc["mfma"].values())
            print(f"    inner loop of {lc['total']} instructions: MFMA {lm}  LDS reads {lc['lds_r']} ({lc['lds_r_bytes']} B/lane)  "
                  f"vmem loads {lc['vmem_ld']} (+{lc['vmem_ld_lds']} DMA)  stores {lc['vmem_st']}  VALU {lc['valu']} + packed {lc['valu_pk']} + "
                  f"transcendental {lc['valu_trans']}  SALU {lc['salu']}  waits {lc['waitcnt']}  barriers {lc['barrier']}"
                  f"   -> per MFMA {lc['lds_r_bytes'] * 64 / lm / 1024:.2f} KB LDS, {(lc['valu'] + lc['valu_pk'] + 4 * lc['valu_trans']) / lm:.2f} VALU slots")
            # additive model, one workgroup per CU: per SIMD the MFMA and VALU issue cycles of its wavefronts add up; the CU's LDS
            # delivers up to 256 B per clock (ds_read_b128: 64 lanes x 16 B in 4 LDS cycles, MI355X_MICROARCH.md LDS table)
            waves = int(get('max_flat_workgroup_size')) // 64 if get('max_flat_workgroup_size') != '?' else 8
            per_simd = max(1, waves // 4)
            mf = sum(n * (32 if k.startswith("f32_32x32") else 16 if k.startswith("f32_16x16x32") else 8) for k, n in lc["mfma"].items())
            va = 4 * (lc["valu"] + lc["valu_pk"]) + 16 * lc["valu_trans"]
            lds = (lc["lds_r_bytes"] + lc["lds_w_bytes"]) * 64 * waves / 256
            print(f"       model per trip ({waves} wavefronts, {per_simd} per SIMD): MFMA {per_simd * mf} + VALU {per_simd * va} = {per_simd * (mf + va)} "
                  f"SIMD cycles, LDS {lds:.0f} CU cycles  -> MFMA share <= {per_simd * mf / max(per_simd * (mf + va), lds):.2f}")
        print()


if __name__ == "__main__":
    main()
