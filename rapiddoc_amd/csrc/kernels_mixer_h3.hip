// Fused PPLCNetV4 channel mixer on the fp16 matrix cores with hi/lo operand splitting ("h3", see kernels_conv_h3.hip
// for the arithmetic: x = hi + lo*2^-11, 3 x v_mfma_f32_32x32x16_f16 per product, fp32 accumulate, error ~5e-7).
// Same structure as kernels_mixer.hip - a 128-pixel X tile resident in LDS, weights streamed in hidden chunks of 32,
// GEMM1 computed transposed so that its accumulator registers feed GEMM2 directly - with two changes:
//   * the X tile and the weight chunks live in LDS as (hi, lo) fp16 pairs (same bytes as fp32); X is split once per tile,
//     the weights once at load time, only the 16 hidden values per lane are split per chunk (after the GELU);
//   * for the 32x32x16 MFMA the A operand holds 8 consecutive k per lane.  The C/D registers of the transposed GEMM1
//     hold hidden index 8*(r>>2) + 4*(lane>>5) + (r&3); W2's columns are therefore PERMUTED inside every 32-wide hidden
//     chunk at weight-preparation time (position 16*(a>>1) + 8*b + 4*(a&1) + c for hidden 8a + 4b + c) so that one
//     ds_read_b128 still yields the B fragment that matches registers 8s .. 8s+7.
// Tried and dropped: a chunk-level software pipeline (GELU + split of chunk j in one scheduling region with the MFMAs of
// GEMM1 for chunk j+1): it needs a second 32-register hidden accumulator, which at C = 192 (192 + 32 output / hidden
// accumulators) spills 49-65 VGPRs (351 us instead of 194) and at C = 96 costs the second workgroup per CU (143 vs 118 us).
// Tried and dropped: a "wide" variant for C = 384 / 192 (64-pixel tile, hidden chunks of 16, the four wavefronts as 2 pixel
// groups x 2 output-channel halves so that two workgroups fit a CU).  GEMM1 and the GELU are then computed twice; it
// measured 107 TF/s at C = 384 (the unfused pair of split-fp16 GEMMs reaches ~170) and 104 at C = 192 (this kernel: 160).
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "rd_device.h"

namespace rd {


static constexpr int HX_BM = 128;
static constexpr int HX_HC = 32;

// DBG: compile-time ablation bits for tools/microbench.py (1 no GELU, 2 no weight streaming, 4 skip GEMM1, 8 skip GEMM2);
// run-time switches here would cut the chunk body into basic blocks the scheduler cannot move instructions across
template <int C, int DBG>
__global__ void __launch_bounds__(256, C <= 96 ? 2 : 1) lc_mixer_h3_kernel(MixerParams p) {
    constexpr int XS = C + 8;              // LDS row stride (halfs) of the X tile: conflict-free b128 reads
    constexpr int NTT = (C + 31) / 32;
    constexpr int W2ROWS = NTT * 32;
    constexpr int KS1 = C / 16;            // k-steps of GEMM1
    // Weight chunks arrive by LDS-DMA (global_load_lds_dwordx4: no staging VGPRs, no ds_write, no address VALU in the
    // loop).  A DMA writes wave-uniform base + lane*16, so the chunk images are UNPADDED and bank conflicts are avoided by
    // XOR-swizzling the 16-byte chunk index with the row on the SOURCE side (W1: below; W2, rows of 32 halfs:
    // chunk ^ ((row >> 2) & 3)) - conflict-free for the ds_read_b128 lane groups.
    // W1 rows are 2C bytes: consecutive rows start 32 / 16 / 8 dwords... apart modulo the 64 banks, which fixes how many
    // low chunk bits may be XORed (the swizzled chunk must stay inside its row: 24 / 12 / 6 chunks) and which row bits
    // must drive them: C = 192: ^((row >> 1) & 7), C = 96: ^((row >> 2) & 3), C = 48: ^((row >> 3) & 1).
    constexpr int SW1_SHIFT = C == 192 ? 1 : C == 96 ? 2 : 3, SW1_MASK = C == 192 ? 7 : C == 96 ? 3 : 1;
    constexpr int NI = C / 8;              // DMA instructions per chunk and matrix: (2 planes x 32 x C halfs) / 1 KB
    constexpr int NIW = (NI + 3) / 4;      // ... per wavefront
    extern __shared__ __attribute__((aligned(16))) _Float16 smemh[];
    _Float16* Xh = smemh;                      // [128][XS]
    _Float16* Xl = Xh + HX_BM * XS;
    _Float16* W1h = Xl + HX_BM * XS;           // [32][C]   (lo plane follows)
    _Float16* W1l = W1h + HX_HC * C;
    _Float16* W2h = W1l + HX_HC * C;           // [W2ROWS][32]: only rows < C are ever written by the DMA
    _Float16* W2l = W2h + W2ROWS * HX_HC;
    float* B1s = reinterpret_cast<float*>(W2l + W2ROWS * HX_HC);       // [2C] hidden biases

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.x * HX_BM;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const _Float16* w1h_g = reinterpret_cast<const _Float16*>(p.w1h);
    const _Float16* w1l_g = reinterpret_cast<const _Float16*>(p.w1l);
    const _Float16* w2h_g = reinterpret_cast<const _Float16*>(p.w2h);
    const _Float16* w2l_g = reinterpret_cast<const _Float16*>(p.w2l);

    // ---- X tile: gate, split once, store as hi / lo
    float amax = 0.f;   // range guard: largest |operand| this thread split (checked once at the end)
    {
        // (loads are unconditional from a clamped row: a predicated load makes the compiler wait per load; rows past M
        //  hold a copy of the last row and are never stored)
        constexpr int QPR = C / 4;
        constexpr int XIT = HX_BM * QPR / 256;
        // two copies of the loop: a run-time `if (gate)` inside it is a branch per iteration, which keeps the compiler from
        // batching the tile loads of different iterations (seen in the ISA: one s_cbranch per load)
        auto load_tile = [&](auto gated) {
#pragma unroll 12
            for (int it = 0; it < XIT; ++it) {
                const int i = tid + 256 * it;
                const int r = i / QPR, q = i - r * QPR;
                const int m = min(m0 + r, p.M - 1);
                f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.xld + 4 * q);
                if (decltype(gated)::value) v *= *reinterpret_cast<const f32x4*>(p.gate + (size_t)(m / p.HW) * C + 4 * q);
                f16x4 hi, lo;
                rd_split4(v, hi, lo);
#pragma unroll
                for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[e]));
                *reinterpret_cast<f16x4*>(&Xh[r * XS + 4 * q]) = hi;
                *reinterpret_cast<f16x4*>(&Xl[r * XS + 4 * q]) = lo;
            }
        };
        // (at C = 192 the kernel sits at its register ceiling and the batched loads cost more than the branches: 207 vs 192 us)
        if (C == 192) {
#pragma unroll 12
            for (int it = 0; it < XIT; ++it) {
                const int i = tid + 256 * it;
                const int r = i / QPR, q = i - r * QPR;
                const int m = min(m0 + r, p.M - 1);
                f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.xld + 4 * q);
                if (p.gate) v *= *reinterpret_cast<const f32x4*>(p.gate + (size_t)(m / p.HW) * C + 4 * q);
                f16x4 hi, lo;
                rd_split4(v, hi, lo);
#pragma unroll
                for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[e]));
                *reinterpret_cast<f16x4*>(&Xh[r * XS + 4 * q]) = hi;
                *reinterpret_cast<f16x4*>(&Xl[r * XS + 4 * q]) = lo;
            }
        } else if (p.gate) {
            load_tile(std::true_type{});
        } else {
            load_tile(std::false_type{});
        }
    }
    // rows [C, W2ROWS) of both W2 planes (C = 48 only) are read by the last n-tile and never written by the DMA
    if (C % 32 != 0)
        for (int i = tid; i < (W2ROWS - C) * HX_HC; i += 256) W2h[C * HX_HC + i] = W2l[C * HX_HC + i] = (_Float16)0.f;
    for (int i = tid; i < 2 * C / 4; i += 256) *reinterpret_cast<f32x4*>(&B1s[4 * i]) = *reinterpret_cast<const f32x4*>(p.b1 + 4 * i);

    // ---- DMA source addressing (per lane, chunk-independent part).  Instruction q of a matrix covers LDS bytes
    // [q*1024, +1024) of its [hi plane | lo plane] image; wavefront w issues q = w, w + 4, ...
    unsigned char* lds_bytes = reinterpret_cast<unsigned char*>(smemh);
    const unsigned w1_base = (unsigned)((W1h - smemh) * 2), w2h_base = (unsigned)((W2h - smemh) * 2), w2l_base = (unsigned)((W2l - smemh) * 2);
    const _Float16* src1[NIW];
    const _Float16* src2[NIW];
    unsigned dst1[NIW], dst2[NIW];
    const int wv = __builtin_amdgcn_readfirstlane(wave);
#pragma unroll
    for (int u = 0; u < NIW; ++u) {
        const int q = wv + 4 * u;
        const int plane = q / (NI / 2), o = (q % (NI / 2)) * 1024 + lane * 16;
        {   // W1 chunk image: 32 rows x 2C bytes per plane
            const int row = o / (2 * C), cp = (o % (2 * C)) >> 4;
            const int c = cp ^ ((row >> SW1_SHIFT) & SW1_MASK);
            src1[u] = (plane ? w1l_g : w1h_g) + (size_t)row * C + 8 * c;
            dst1[u] = w1_base + (unsigned)plane * (HX_HC * C * 2) + (unsigned)(q % (NI / 2)) * 1024u;
        }
        {   // W2 chunk image: C rows x 64 bytes per plane
            const int row = o >> 6, cp = (o & 63) >> 4;
            const int c = cp ^ ((row >> 2) & 3);
            src2[u] = (plane ? w2l_g : w2h_g) + (size_t)row * (2 * C) + 8 * c;
            dst2[u] = (plane ? w2l_base : w2h_base) + (unsigned)(q % (NI / 2)) * 1024u;
        }
    }
    auto dma = [&](const _Float16* src, unsigned dst) {
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(lds_bytes + dst), 16, 0, 0);
    };
    auto issue_w1 = [&](int j) {
#pragma unroll
        for (int u = 0; u < NIW; ++u)
            if (wv + 4 * u < NI) dma(src1[u] + (size_t)j * HX_HC * C, dst1[u]);
    };
    auto issue_w2 = [&](int j) {
#pragma unroll
        for (int u = 0; u < NIW; ++u)
            if (wv + 4 * u < NI) dma(src2[u] + j * HX_HC, dst2[u]);
    };
    issue_w1(0);
    issue_w2(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // X tile, biases, chunk 0 of both matrices

    f32x16 y1[NTT], y2[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) y1[n][r] = y2[n][r] = 0.f;

    const int xo = (wave * 32 + l31) * XS + 8 * lhi;   // this lane's pixel row (B operand of GEMM1)
    const int f1 = (l31 >> SW1_SHIFT) & SW1_MASK, f2 = (l31 >> 2) & 3;   // swizzle keys of this lane's weight rows
    const int w1row = l31 * C;                            // hidden row (A operand of GEMM1), halfs
    const int w2row = l31 * HX_HC;                        // output-channel row (B operand of GEMM2), halfs
    auto w1_at = [&](int ks) { return w1row + (((2 * ks + lhi) ^ f1) << 3); };
    auto w2_at = [&](int n, int s2) { return n * 32 * HX_HC + w2row + (((2 * s2 + lhi) ^ f2) << 3); };
    constexpr int NCHUNK = 2 * C / HX_HC;
    for (int j = 0; j < NCHUNK; ++j) {
        const bool more = (DBG & 2) ? false : j + 1 < NCHUNK;
        f32x4 b1v[4];   // this chunk's hidden biases
#pragma unroll
        for (int g = 0; g < 4; ++g) b1v[g] = *reinterpret_cast<const f32x4*>(&B1s[j * HX_HC + g * 8 + 4 * lhi]);
        // GEMM1 (transposed): Ht = W1c . X^T
        f32x16 h1, h2;
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[r] = h2[r] = 0.f;
        // one wavefront per SIMD (LDS-limited): nothing hides a ds_read latency, so the fragments run through a
        // 3-deep register ring and the scheduler is told to interleave the reads of step g+2 with the MFMAs of step g
        if (!(DBG & 4)) {
            f16x8 ah[3], al[3], bh[3], bl[3];
#pragma unroll
            for (int g = 0; g < 2 && g < KS1; ++g) {
                ah[g] = *reinterpret_cast<const f16x8*>(&W1h[w1_at(g)]);
                bh[g] = *reinterpret_cast<const f16x8*>(&Xh[xo + g * 16]);
                bl[g] = *reinterpret_cast<const f16x8*>(&Xl[xo + g * 16]);
                al[g] = *reinterpret_cast<const f16x8*>(&W1l[w1_at(g)]);
            }
#pragma unroll
            for (int g = 0; g < KS1; ++g) {
                const int c = g % 3, nx = (g + 2) % 3;
                if (g + 2 < KS1) {
                    ah[nx] = *reinterpret_cast<const f16x8*>(&W1h[w1_at(g + 2)]);
                    bh[nx] = *reinterpret_cast<const f16x8*>(&Xh[xo + (g + 2) * 16]);
                    bl[nx] = *reinterpret_cast<const f16x8*>(&Xl[xo + (g + 2) * 16]);
                    al[nx] = *reinterpret_cast<const f16x8*>(&W1l[w1_at(g + 2)]);
                }
                h1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bh[c], h1, 0, 0, 0);
                h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[c], bl[c], h2, 0, 0, 0);
                h2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[c], bh[c], h2, 0, 0, 0);
                if (g + 2 < KS1) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
        }
        // every wavefront is done with W1 chunk j (and W2 chunk j, requested one barrier ago, has landed): request W1
        // chunk j+1 - it lands under the GELU and GEMM2
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (more) issue_w1(j + 1);
        // bias + GELU in fp32, then split the 16 hidden values of this lane into the A fragments of GEMM2:
        // register r (hidden 8*(r>>2) + 4*lhi + (r&3)) is element r&7 of k-step r>>3
        f16x8 hh[2], hl[2];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = b1v[g];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = g * 4 + e;
                const float pre = fmaf(h2[r], 1.f / 2048.f, h1[r]) + bv[e];
                const float v = (DBG & 1) ? pre : rd_gelu(pre);
                _Float16 a, b;
                rd_split(v, a, b);
                amax = fmaxf(amax, fabsf(v));
                hh[r >> 3][r & 7] = a;
                hl[r >> 3][r & 7] = b;
            }
        }
        // GEMM2: Y += H . W2c^T (K = 32, two k-steps); W2's hidden columns were permuted at load time to match
        if (!(DBG & 8)) {
            constexpr int NP = 2 * NTT;   // (k-step, n-tile) pairs, same 3-deep fragment ring
            f16x8 bh[3], bl[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bh[i] = *reinterpret_cast<const f16x8*>(&W2h[w2_at(i % NTT, i / NTT)]);
                bl[i] = *reinterpret_cast<const f16x8*>(&W2l[w2_at(i % NTT, i / NTT)]);
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int c = i % 3, nx = (i + 2) % 3, s = i / NTT, n = i % NTT;
                if (i + 2 < NP) {
                    bh[nx] = *reinterpret_cast<const f16x8*>(&W2h[w2_at((i + 2) % NTT, (i + 2) / NTT)]);
                    bl[nx] = *reinterpret_cast<const f16x8*>(&W2l[w2_at((i + 2) % NTT, (i + 2) / NTT)]);
                }
                y1[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh[s], bh[c], y1[n], 0, 0, 0);
                y2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hh[s], bl[c], y2[n], 0, 0, 0);
                y2[n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(hl[s], bh[c], y2[n], 0, 0, 0);
                if (i + 2 < NP) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
        }
        // every wavefront is done with W2 chunk j, and W1 chunk j+1 has landed: request W2 chunk j+1 (lands under GEMM1)
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (more) issue_w2(j + 1);
    }
    // ---- epilogue: + b2 + residual (re-read from global, so it stays exact fp32; unconditional clamped loads), strided store.
    // Two alternatives measured slower: rebuilding the residual from the (hi, lo) tile in LDS (2-byte LDS reads, +10 %)
    // and parking the result rows in LDS to stream them out as float4 rows (+8 %); seeding the accumulators with X . I
    // (two MFMAs per 16 channels against a per-lane identity fragment, no re-read at all: +9 %, and 22-bit residuals).
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        const int co = n * 32 + l31;
        if (co >= C) continue;
        const float bv = p.b2[co];
        float res[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = min(m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi, p.M - 1);
            res[r] = p.x[(size_t)m * p.xld + co];
            if (p.gate) res[r] *= p.gate[(size_t)(m / p.HW) * C + co];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const float o = fmaf(y2[n][r], 1.f / 2048.f, y1[n][r]) + bv + res[r];
            if (!(fabsf(o) < INFINITY)) amax = INFINITY;   // NaN operands are invisible to the fmaxf chains but reach the output
            if (m < p.M) __builtin_nontemporal_store(o, &p.y[(size_t)m * p.yld + co]);
        }
    }
    if (!(amax < 65504.f) && p.range_flag) rd_raise_flag(p.range_flag);
}


template <int C, int DBG>
static void launch_mixer_h3_c(const MixerParams& p, hipStream_t s) {
    const size_t sh = (size_t)(2 * HX_BM * (C + 8) + 2 * HX_HC * C + 2 * ((C + 31) / 32 * 32) * HX_HC) * sizeof(_Float16) + 2 * C * sizeof(float);
    static unsigned long long lds_ok = 0;
    rd_allow_dynamic_lds((const void*)lc_mixer_h3_kernel<C, DBG>, sh, lds_ok);
    hipLaunchKernelGGL((lc_mixer_h3_kernel<C, DBG>), dim3((p.M + HX_BM - 1) / HX_BM), dim3(256), sh, s, p);
}

void launch_mixer_fused_h3(const MixerParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    switch (p.C) {
        case 48: launch_mixer_h3_c<48, 0>(p, s); break;
        case 96: launch_mixer_h3_c<96, 0>(p, s); break;
        case 192:
            switch (p.dbg) {   // ablation variants: microbenchmark only
                case 0: launch_mixer_h3_c<192, 0>(p, s); break;
                case 1: launch_mixer_h3_c<192, 1>(p, s); break;
                case 2: launch_mixer_h3_c<192, 2>(p, s); break;
                case 4: launch_mixer_h3_c<192, 4>(p, s); break;
                case 8: launch_mixer_h3_c<192, 8>(p, s); break;
                case 10: launch_mixer_h3_c<192, 10>(p, s); break;
                case 11: launch_mixer_h3_c<192, 11>(p, s); break;
                default: launch_mixer_h3_c<192, 15>(p, s); break;
            }
            break;
        default: break;
    }
}

// host: split (and for W2 permute) the mixer weights.  w1 [2C][C], w2 [C][2C] fp32 (BN folded)
void prepare_mixer_weights_h3(const float* w1, const float* w2, int C, std::vector<uint16_t>& w1h, std::vector<uint16_t>& w1l,
                              std::vector<uint16_t>& w2h, std::vector<uint16_t>& w2l) {
    auto put = [](float v, uint16_t& hb, uint16_t& lb) {
        const _Float16 h = (_Float16)v;
        const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
        __builtin_memcpy(&hb, &h, 2);
        __builtin_memcpy(&lb, &l, 2);
    };
    const int H2 = 2 * C;
    w1h.assign((size_t)H2 * C, 0); w1l.assign((size_t)H2 * C, 0);
    w2h.assign((size_t)C * H2, 0); w2l.assign((size_t)C * H2, 0);
    for (size_t i = 0; i < (size_t)H2 * C; ++i) put(w1[i], w1h[i], w1l[i]);
    for (int n = 0; n < C; ++n)
        for (int hid = 0; hid < H2; ++hid) {
            const int chunk = hid / 32, q = hid % 32;
            const int a = q >> 3, b = (q >> 2) & 1, c = q & 3;
            const int pos = 16 * (a >> 1) + 8 * b + 4 * (a & 1) + c;
            put(w2[(size_t)n * H2 + hid], w2h[(size_t)n * H2 + chunk * 32 + pos], w2l[(size_t)n * H2 + chunk * 32 + pos]);
        }
}

}  // namespace rd
