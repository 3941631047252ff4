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
// Tried and dropped: a "wide" variant for C = 384 / 192 (64-pixel tile, hidden chunks of 16, the four wavefronts as 2 pixel
// groups x 2 output-channel halves so that two workgroups fit a CU).  GEMM1 and the GELU are then computed twice; it
// measured 107 TF/s at C = 384 (the unfused pair of split-fp16 GEMMs reaches ~170) and 104 at C = 192 (this kernel: 160).
#include <cstdlib>
#include <vector>

#include "rd_kernels.h"

namespace rd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

static constexpr int HX_BM = 128;
static constexpr int HX_HC = 32;

__device__ __forceinline__ float hx_gelu(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erfz = 1.f - poly * t * __expf(-z * z);
    return 0.5f * v * (1.f + copysignf(erfz, v));
}
__device__ __forceinline__ void hx_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)__builtin_fmaf((float)hi, -2048.f, v * 2048.f);   // exact; one v_fma_mix*_f16 (see kernels_conv_h3.hip)
}

template <int C>
__global__ void __launch_bounds__(256, C <= 96 ? 2 : 1) lc_mixer_h3_kernel(MixerParams p) {
    constexpr int XS = C + 8;              // LDS row stride (halfs) of X and the W1 chunk: conflict-free b128 reads
    constexpr int WS = HX_HC + 8;          // W2 chunk row stride (halfs)
    constexpr int NTT = (C + 31) / 32;
    constexpr int W2ROWS = NTT * 32;
    constexpr int KS1 = C / 16;            // k-steps of GEMM1
    constexpr int WQ = C * 4;              // 16-byte pieces per weight chunk component (W1 hi: 32 x C halfs = C*4 pieces)
    constexpr int WL = (4 * WQ + 255) / 256;   // pieces per thread: W1 hi, W1 lo, W2 hi, W2 lo
    extern __shared__ __attribute__((aligned(16))) _Float16 smemh[];
    _Float16* Xh = smemh;                      // [128][XS]
    _Float16* Xl = Xh + HX_BM * XS;
    _Float16* W1h = Xl + HX_BM * XS;           // [32][XS]
    _Float16* W1l = W1h + HX_HC * XS;
    _Float16* W2h = W1l + HX_HC * XS;          // [W2ROWS][WS]
    _Float16* W2l = W2h + W2ROWS * WS;

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
#pragma unroll 12
        for (int it = 0; it < XIT; ++it) {
            const int i = tid + 256 * it;
            const int r = i / QPR, q = i - r * QPR;
            const int m = min(m0 + r, p.M - 1);
            f32x4 v = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.xld + 4 * q);
            if (p.gate) v *= *reinterpret_cast<const f32x4*>(p.gate + (size_t)(m / p.HW) * C + 4 * q);
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, b;
                hx_split(v[e], a, b);
                hi[e] = a;
                lo[e] = b;
                amax = fmaxf(amax, fabsf(v[e]));
            }
            *reinterpret_cast<f16x4*>(&Xh[r * XS + 4 * q]) = hi;
            *reinterpret_cast<f16x4*>(&Xl[r * XS + 4 * q]) = lo;
        }
        if (C % 32 != 0)
            for (int i = tid; i < (W2ROWS - C) * WS; i += 256) W2h[C * WS + i] = W2l[C * WS + i] = (_Float16)0.f;
    }
    // weight chunk j: W1 rows [32j, 32j+32) x C  and  W2 rows [0, C) x (permuted) hidden [32j, 32j+32)
    u32x4 wreg[WL];
    auto piece = [&](int t, int j, const _Float16*& src, _Float16*& dst) {
        // t in [0, 4*WQ): 0..WQ W1 hi, WQ..2WQ W1 lo, then W2 hi, W2 lo
        const int comp = t / WQ, i = t - comp * WQ;
        if (comp < 2) {
            const int r = i / (C / 8), c8 = i - r * (C / 8);
            src = (comp == 0 ? w1h_g : w1l_g) + (size_t)(j * HX_HC + r) * C + 8 * c8;
            dst = (comp == 0 ? W1h : W1l) + r * XS + 8 * c8;
        } else {
            const int r = i >> 2, c8 = i & 3;
            src = (comp == 2 ? w2h_g : w2l_g) + (size_t)r * (2 * C) + j * HX_HC + 8 * c8;
            dst = (comp == 2 ? W2h : W2l) + r * WS + 8 * c8;
        }
    };
    auto load_w = [&](int j) {
#pragma unroll
        for (int u = 0; u < WL; ++u) {
            const int t = tid + 256 * u;
            if (t < 4 * WQ) {
                const _Float16* src;
                _Float16* dst;
                piece(t, j, src, dst);
                wreg[u] = *reinterpret_cast<const u32x4*>(src);
            }
        }
    };
    auto store_w = [&](int j) {
#pragma unroll
        for (int u = 0; u < WL; ++u) {
            const int t = tid + 256 * u;
            if (t < 4 * WQ) {
                const _Float16* src;
                _Float16* dst;
                piece(t, j, src, dst);
                *reinterpret_cast<u32x4*>(dst) = wreg[u];
            }
        }
    };
    load_w(0);
    store_w(0);
    __syncthreads();

    f32x16 y1[NTT], y2[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) y1[n][r] = y2[n][r] = 0.f;

    const int xo = (wave * 32 + l31) * XS + 8 * lhi;   // this lane's pixel row (B operand of GEMM1)
    const int w1o = l31 * XS + 8 * lhi;                 // hidden row (A operand of GEMM1)
    const int w2o = l31 * WS + 8 * lhi;                 // output-channel row (B operand of GEMM2)
    constexpr int NCHUNK = 2 * C / HX_HC;
    for (int j = 0; j < NCHUNK; ++j) {
        const bool more = (p.dbg & 2) ? false : j + 1 < NCHUNK;
        if (more) load_w(j + 1);
        f32x4 b1v[4];   // this chunk's hidden biases (issued early: the loads complete under GEMM1)
#pragma unroll
        for (int g = 0; g < 4; ++g) b1v[g] = *reinterpret_cast<const f32x4*>(p.b1 + j * HX_HC + g * 8 + 4 * lhi);
        // GEMM1 (transposed): Ht = W1c . X^T
        f32x16 h1, h2;
#pragma unroll
        for (int r = 0; r < 16; ++r) h1[r] = h2[r] = 0.f;
        // one wavefront per SIMD (LDS-limited): nothing hides a ds_read latency, so the fragments run through a
        // 3-deep register ring and the scheduler is told to interleave the reads of step g+2 with the MFMAs of step g
        if (!(p.dbg & 4)) {
            f16x8 ah[3], al[3], bh[3], bl[3];
#pragma unroll
            for (int g = 0; g < 2 && g < KS1; ++g) {
                ah[g] = *reinterpret_cast<const f16x8*>(&W1h[w1o + g * 16]);
                bh[g] = *reinterpret_cast<const f16x8*>(&Xh[xo + g * 16]);
                bl[g] = *reinterpret_cast<const f16x8*>(&Xl[xo + g * 16]);
                al[g] = *reinterpret_cast<const f16x8*>(&W1l[w1o + g * 16]);
            }
#pragma unroll
            for (int g = 0; g < KS1; ++g) {
                const int c = g % 3, nx = (g + 2) % 3;
                if (g + 2 < KS1) {
                    ah[nx] = *reinterpret_cast<const f16x8*>(&W1h[w1o + (g + 2) * 16]);
                    bh[nx] = *reinterpret_cast<const f16x8*>(&Xh[xo + (g + 2) * 16]);
                    bl[nx] = *reinterpret_cast<const f16x8*>(&Xl[xo + (g + 2) * 16]);
                    al[nx] = *reinterpret_cast<const f16x8*>(&W1l[w1o + (g + 2) * 16]);
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
                const float v = (p.dbg & 1) ? pre : hx_gelu(pre);
                _Float16 a, b;
                hx_split(v, a, b);
                amax = fmaxf(amax, fabsf(v));
                hh[r >> 3][r & 7] = a;
                hl[r >> 3][r & 7] = b;
            }
        }
        // GEMM2: Y += H . W2c^T (K = 32, two k-steps); W2's hidden columns were permuted at load time to match
        if (!(p.dbg & 8)) {
            constexpr int NP = 2 * NTT;   // (k-step, n-tile) pairs, same 3-deep fragment ring
            f16x8 bh[3], bl[3];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bh[i] = *reinterpret_cast<const f16x8*>(&W2h[w2o + (i % NTT) * 32 * WS + (i / NTT) * 16]);
                bl[i] = *reinterpret_cast<const f16x8*>(&W2l[w2o + (i % NTT) * 32 * WS + (i / NTT) * 16]);
            }
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                const int c = i % 3, nx = (i + 2) % 3, s = i / NTT, n = i % NTT;
                if (i + 2 < NP) {
                    bh[nx] = *reinterpret_cast<const f16x8*>(&W2h[w2o + ((i + 2) % NTT) * 32 * WS + ((i + 2) / NTT) * 16]);
                    bl[nx] = *reinterpret_cast<const f16x8*>(&W2l[w2o + ((i + 2) % NTT) * 32 * WS + ((i + 2) / NTT) * 16]);
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
        if (!(p.dbg & 2)) __syncthreads();
        if (more) {
            store_w(j + 1);
            __syncthreads();
        }
    }
    if (!(amax < 65504.f) && p.range_flag) atomicOr(p.range_flag, 1u);   // also catches NaN
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
            if (m < p.M) p.y[(size_t)m * p.yld + co] = fmaf(y2[n][r], 1.f / 2048.f, y1[n][r]) + bv + res[r];
        }
    }
}


template <int C>
static void launch_mixer_h3_c(const MixerParams& p, hipStream_t s) {
    const size_t sh = (size_t)(2 * (HX_BM + HX_HC) * (C + 8) + 2 * ((C + 31) / 32 * 32) * (HX_HC + 8)) * sizeof(_Float16);
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)lc_mixer_h3_kernel<C>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sh);
        attr_set = true;
    }
    hipLaunchKernelGGL((lc_mixer_h3_kernel<C>), dim3((p.M + HX_BM - 1) / HX_BM), dim3(256), sh, s, p);
}

void launch_mixer_fused_h3(const MixerParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    switch (p.C) {
        case 48: launch_mixer_h3_c<48>(p, s); break;
        case 96: launch_mixer_h3_c<96>(p, s); break;
        case 192: launch_mixer_h3_c<192>(p, s); break;
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
