// Fused PPLCNetV4 channel mixer (rec_lcnetv4.py:226-236: y = xg + W2 * GELU(W1 * xg + b1) + b2, xg = x * SE gate) for the NARROW
// blocks (C = 96; any multiple of 32 whose weights fit in LDS) - split-fp16 arithmetic (three MFMAs per product; rounds 2-4:
// x = hi + lo * 2^-11 into two fp32 accumulators as in kernels_mixer_h3.hip - round 5: one accumulator, see RS below).
//
// At C = 96 a mixer is HBM-bound: 8 M C^2 = 7.7 GFLOP against 80 MB of activations (M = 104 448): ~12 us of MFMA issue, ~16 us of
// VALU (GELU) and 20 us of HBM at 4 TB/s.  The round-1 kernel (one wavefront per SIMD, phases separated by barriers) and the
// weight-streaming kernel of round 2 (kernels_mixer_ws.hip) both measured 97-105 us there.  What this width allows and C = 192
// does not: ALL weights (hi and lo of W1 and W2: 147 KB at C = 96) fit in LDS.  So:
//   * one 16-wavefront workgroup per CU copies the weight image (pre-arranged as MFMA A fragments, 1 KB each) into LDS once, one
//     barrier, and after that NO wavefront ever waits for another: each loops over its own 16-pixel tiles;
//   * activations never touch LDS: X^T is the MFMA B operand from registers (16x16x32, lane = pixel, 8 channels), the hidden block's
//     C/D registers ARE the next B fragment (W2's columns are permuted to match), Y^T's C/D registers are four consecutive output
//     channels of the lane's pixel: residual and store are float4 at the addresses the tile was loaded from;
//   * four wavefronts per SIMD (<= 128 VGPRs) hide the tile loads and the LDS fragment reads of one another.
// Round 5 (RS): the tile is read ONCE.  W1's input channels are permuted in the image so that the 8 k-slots a lane feeds to k-step ks
// of GEMM1 are the 2 x 4 channels it owns in the C/D layout of Y^T (blocks 2 ks, 2 ks + 1: channels 16 ob + 4 kg .. + 4, the ws
// kernel's arrangement): the tile load uses the epilogue's float4 addresses and the residual is re-formed from the split fragments
// already in registers, x ~ hi + lo (the sum is exact in fp32; it differs from x by lo's rounding: lo = fp16(x - hi) rounds a 13-bit
// remainder to 11 bits, <= 2^-22 |x|, and where lo is an fp16 subnormal - |x| < 2^-3 - by an ABSOLUTE 2^-25; the residual stream therefore
// picks up to ~4 fp32 ulps per block instead of none: tests/test_gpu_parity.py::test_mixer_residual_from_split_fragments_small_inputs)
// instead of a second fetch of the tile and the gate
// (counters, round 4: 361 MB moved per launch against 255 MB algorithmic).  The fragments must stay live through the hidden loop for
// that, which the two accumulator sets of the round-2 arithmetic leave no room for at 128 registers (four wavefronts per SIMD) - so
// the kernel moves to the ONE-accumulator split of kernels_gemm_h1.hip / kernels_mixer_ws.hip: low planes unscaled (x = hi + lo;
// gfx950's fp16 matrix cores keep subnormal inputs), weights pre-scaled per matrix by a power of two (max |w| in [2^13, 2^14)), the
// sums multiplied by the exact inverse scales (p.ws_inv1 / ws_inv2) where they leave the accumulators.
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "rd_device.h"

namespace rd {

static constexpr int MR_WAVES = 16;

// x = hi + lo, hi = fp16(x), lo = fp16(x - hi): the difference is exact, the second rounding keeps 11 more bits (or stops at the
// fp16 subnormal spacing: absolute error 2^-25)
__device__ __forceinline__ void mr_split(float v, float neg1, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)__builtin_fmaf((float)hi, neg1, v);       // neg1: -1 in a register the compiler cannot fold (keeps it one v_fma_mix)
}

template <int C>
struct ResGeom {
    static constexpr int H2 = 2 * C;            // hidden width
    static constexpr int KS1 = C / 32;          // k-steps of GEMM1
    static constexpr int HB = H2 / 16;          // hidden blocks of 16
    static constexpr int NQ = H2 / 32;          // pairs of hidden blocks = k-steps of GEMM2
    static constexpr int OB = C / 16;           // output blocks of 16
    static constexpr int F1 = HB * KS1;         // W1 fragments per plane
    static constexpr int F2 = OB * NQ;          // W2 fragments per plane
    static constexpr size_t IMG_BYTES = (size_t)2 * (F1 + F2) * 1024;
    static constexpr size_t LDS_BYTES = IMG_BYTES + (size_t)(H2 + C) * sizeof(float);
    static constexpr size_t LDS_BYTES_DW = LDS_BYTES + (size_t)10 * C * sizeof(float);      // + the depthwise taps [9][C] and bias [C]
};

// image layout: [W1 hi: F1 fragments][W1 lo][W2 hi: F2][W2 lo]; W1 fragment (hb, ks), W2 fragment (ob, q)
// DW (round 6): the block's depthwise 3x3 / stride 1 / pad 1 (rec_lcnetv4.py:187-206, the token mixer of a no-SE block) is computed in the
// tile load: p.x is the block's INPUT, a lane forms the six float4 of ITS pixel's depthwise output from nine float4 taps each (clamped
// addresses, masked values: rows outside the map and columns outside the line's own width read as the conv's zero padding) with the taps'
// weights from LDS - bias first, then taps in (kh, kw) order, one fma per tap (the accumulation order of dwconv3x3_lds_kernel, whose
// compiled form mixes fused and unfused multiply-adds: the two routes agree to a few ulps, tests/test_gpu_mixer_dw.py) - and goes on to
// the split exactly as the loaded tile did.  The depthwise output is
// never written: one write + one read of the activation less per block; the taps' re-reads are L1 / L2 hits.
template <int C, bool GATED, bool RS, bool DW = false>
__global__ void __launch_bounds__(1024) lc_mixer_res_kernel(MixerParams p, const unsigned char* __restrict__ wimg, int n_tiles) {
    static_assert(!DW || (!GATED && RS), "the fused depthwise form: no gate, residual from the fragments");
    using G = ResGeom<C>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* B1s = reinterpret_cast<float*>(lds + G::IMG_BYTES);
    float* B2s = B1s + G::H2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const u32x4* src = reinterpret_cast<const u32x4*>(wimg);
        u32x4* dst = reinterpret_cast<u32x4*>(lds);
        for (int i = tid; i < (int)(G::IMG_BYTES / 16); i += 1024) dst[i] = src[i];
        for (int i = tid; i < G::H2; i += 1024) B1s[i] = p.b1[i];
        for (int i = tid; i < C; i += 1024) B2s[i] = p.b2[i];
        if constexpr (DW) {
            float* DWs = B2s + C;
            for (int i = tid; i < 9 * C; i += 1024) DWs[i] = p.dw_w[i];
            for (int i = tid; i < C; i += 1024) DWs[9 * C + i] = p.dw_b[i];
        }
    }
    __syncthreads();
    const unsigned char* W1h = lds;
    const unsigned char* W1l = lds + (size_t)G::F1 * 1024;
    const unsigned char* W2h = lds + (size_t)2 * G::F1 * 1024;
    const unsigned char* W2l = W2h + (size_t)G::F2 * 1024;
    const int px = lane & 15, kg = lane >> 4;
    const unsigned lo16 = (unsigned)lane * 16u;
    float amax = 0.f;
    float neg1 = -1.f;
    asm volatile("" : "+s"(neg1));
    const float inv1 = p.ws_inv1, inv2 = p.ws_inv2;
    typedef __amdgpu_buffer_rsrc_t rsrc_t;
    const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, 0x7fffffff, 0x00020000);      // (DW: the tap loads)

    for (int tile = (int)blockIdx.x * MR_WAVES + wave; tile < n_tiles; tile += (int)gridDim.x * MR_WAVES) {
        const int mm = tile * 16 + px;
        const int m = min(mm, p.M - 1);                 // rows past M re-read the last row and are never stored
        const float* xp = p.x + (size_t)m * p.xld + 4 * kg;
        const float* gp = GATED ? p.gate + (size_t)(m / p.HW) * C + 4 * kg : nullptr;
        // ---- X^T as B fragments: k-step ks, lane (px, kg) = channels 16 (2 ks) + 4 kg .. + 4 and 16 (2 ks + 1) + 4 kg .. + 4 (W1's
        // columns are permuted to this order in the image): the lane's own channels of output blocks 2 ks and 2 ks + 1
        f16x8 xh[G::KS1], xl[G::KS1];
        // DW: the nine taps of this lane's pixel = three row offsets + three column offsets (clamped into the map) and their validity
        // (byte offsets into a buffer descriptor over the input tensor: the launcher checks that it is smaller than 2^31 bytes)
        unsigned roff[3] = {0, 0, 0}, coff[3] = {0, 0, 0};
        unsigned tap_ok = 0;
        if constexpr (DW) {
            const int n = m / p.HW, rem = m - n * p.HW;
            const int h = rem / p.dwW, w = rem - h * p.dwW;
            const int wlim = p.dw_line_w ? min(p.dwW, p.dw_line_w[n * p.dw_line_w_stride]) : p.dwW;
            const unsigned xb = ((unsigned)n * (unsigned)p.HW * (unsigned)p.xld + 4u * (unsigned)kg) * 4u;
            unsigned rok = 0, cok = 0;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ih = h - 1 + k, iw = w - 1 + k;
                if ((unsigned)ih < (unsigned)p.dwH) rok |= 1u << k;
                if ((unsigned)iw < (unsigned)wlim) cok |= 1u << k;
                roff[k] = xb + (unsigned)(min(max(ih, 0), p.dwH - 1) * p.dwW) * (unsigned)p.xld * 4u;
                coff[k] = (unsigned)min(max(iw, 0), p.dwW - 1) * (unsigned)p.xld * 4u;
            }
#pragma unroll
            for (int k = 0; k < 9; ++k)
                if (((rok >> (k / 3)) & 1u) && ((cok >> (k % 3)) & 1u)) tap_ok |= 1u << k;
        }
#pragma unroll
        for (int ks = 0; ks < G::KS1; ++ks) {
            f32x4 v0, v1;
            if constexpr (DW) {
                const float* DWs = B2s + C;
                v0 = *reinterpret_cast<const f32x4*>(&DWs[9 * C + 32 * ks + 4 * kg]);
                v1 = *reinterpret_cast<const f32x4*>(&DWs[9 * C + 32 * ks + 16 + 4 * kg]);
                // one kernel row at a time: six unconditional float4 loads in flight (all eighteen at once do not fit 128 registers)
                auto taps = [&](auto k0c, auto k1c) {
                    constexpr int K0 = decltype(k0c)::value, K1 = decltype(k1c)::value;
                    f32x4 t0[K1 - K0], t1[K1 - K0];
#pragma unroll
                    for (int k = K0; k < K1; ++k) {
                        const unsigned off = roff[k / 3] + coff[k % 3];
                        t0[k - K0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 128 * ks, 0));
                        t1[k - K0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 128 * ks + 64, 0));
                    }
#pragma unroll
                    for (int k = K0; k < K1; ++k) {
                        const bool ok = (tap_ok >> k) & 1u;
                        const f32x4 w0 = *reinterpret_cast<const f32x4*>(&DWs[k * C + 32 * ks + 4 * kg]);
                        const f32x4 w1 = *reinterpret_cast<const f32x4*>(&DWs[k * C + 32 * ks + 16 + 4 * kg]);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            v0[e] = __builtin_fmaf(ok ? t0[k - K0][e] : 0.f, w0[e], v0[e]);
                            v1[e] = __builtin_fmaf(ok ? t1[k - K0][e] : 0.f, w1[e], v1[e]);
                        }
                    }
                };
                __builtin_amdgcn_sched_barrier(0);           // (keeps the next group's loads out of this group's registers)
                taps(std::integral_constant<int, 0>{}, std::integral_constant<int, 3>{});
                __builtin_amdgcn_sched_barrier(0);
                taps(std::integral_constant<int, 3>{}, std::integral_constant<int, 6>{});
                __builtin_amdgcn_sched_barrier(0);
                taps(std::integral_constant<int, 6>{}, std::integral_constant<int, 9>{});
                __builtin_amdgcn_sched_barrier(0);
            } else {
                v0 = *reinterpret_cast<const f32x4*>(xp + 32 * ks);
                v1 = *reinterpret_cast<const f32x4*>(xp + 32 * ks + 16);
            }
            if (GATED) {
                v0 *= *reinterpret_cast<const f32x4*>(gp + 32 * ks);
                v1 *= *reinterpret_cast<const f32x4*>(gp + 32 * ks + 16);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 h0, l0, h1, l1;
                mr_split(v0[e], neg1, h0, l0);
                mr_split(v1[e], neg1, h1, l1);
                xh[ks][e] = h0; xh[ks][4 + e] = h1;
                xl[ks][e] = l0; xl[ks][4 + e] = l1;
                amax = fmaxf(amax, fmaxf(fabsf(v0[e]), fabsf(v1[e])));
            }
        }
        f32x4 y1[G::OB];
#pragma unroll
        for (int ob = 0; ob < G::OB; ++ob) y1[ob] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
        for (int q = 0; q < G::NQ; ++q) {
            // ---- GEMM1: hidden blocks 2q, 2q+1 (16 hidden x 16 pixels each)
            f16x8 hh, hl;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                f32x4 a1 = {0.f, 0.f, 0.f, 0.f};
                const int hb = 2 * q + b;
#pragma unroll
                for (int ks = 0; ks < G::KS1; ++ks) {
                    const unsigned fo = (unsigned)(hb * G::KS1 + ks) * 1024u + lo16;
                    const f16x8 wh = *reinterpret_cast<const f16x8*>(W1h + fo);
                    const f16x8 wl = *reinterpret_cast<const f16x8*>(W1l + fo);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xh[ks], a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, xl[ks], a1, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, xh[ks], a1, 0, 0, 0);
                }
                // bias, GELU, split: C/D rows 4 kg + r of block hb = hidden 16 hb + 4 kg + r -> B slot 4 b + r of k-step q
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&B1s[16 * hb + 4 * kg]);
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const f32x2 v = rd_gelu2(f32x2{fmaf(a1[r], inv1, bv[r]), fmaf(a1[r + 1], inv1, bv[r + 1])});
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        _Float16 h, l;
                        mr_split(v[u], neg1, h, l);
                        hh[4 * b + r + u] = h;
                        hl[4 * b + r + u] = l;
                        amax = fmaxf(amax, fabsf(v[u]));
                    }
                }
            }
            // ---- GEMM2: every output block takes k-step q
#pragma unroll
            for (int ob = 0; ob < G::OB; ++ob) {
                const unsigned fo = (unsigned)(ob * G::NQ + q) * 1024u + lo16;
                const f16x8 wh = *reinterpret_cast<const f16x8*>(W2h + fo);
                const f16x8 wl = *reinterpret_cast<const f16x8*>(W2l + fo);
                y1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hh, y1[ob], 0, 0, 0);
                y1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wh, hl, y1[ob], 0, 0, 0);
                y1[ob] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wl, hh, y1[ob], 0, 0, 0);
            }
        }
        // ---- epilogue: C/D rows 4 kg + r of output block ob = channels 16 ob + 4 kg .. + 4 of pixel px: + b2 + gated x, float4 store
        if (mm < p.M) {
            const float* xr = p.x + (size_t)m * p.xld + 4 * kg;
            const float* gr = GATED ? p.gate + (size_t)(m / p.HW) * C + 4 * kg : nullptr;
            float* yp = p.y + (size_t)m * p.yld + 4 * kg;
#pragma unroll
            for (int ob = 0; ob < G::OB; ++ob) {
                f32x4 v;
                if constexpr (RS) {          // the gated tile, re-formed from its split fragments
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        v[e] = (float)xl[ob >> 1][(ob & 1) * 4 + e] + (float)xh[ob >> 1][(ob & 1) * 4 + e];
                } else {
                    v = *reinterpret_cast<const f32x4*>(xr + 16 * ob);
                    if (GATED) v *= *reinterpret_cast<const f32x4*>(gr + 16 * ob);
                }
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&B2s[16 * ob + 4 * kg]);
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = fmaf(y1[ob][e], inv2, bv[e]) + v[e];
                // a NaN operand is invisible to the fmaxf chains (they return the non-NaN operand) but reaches the output
                if (!(fabsf(o[0]) + fabsf(o[1]) + fabsf(o[2]) + fabsf(o[3]) < INFINITY)) amax = INFINITY;
                __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(yp + 16 * ob));
            }
        }
    }
    if (!(amax < 65504.f) && p.range_flag) rd_raise_flag(p.range_flag);   // (NaN: through the output check in the epilogue)
}

// the depthwise 3x3 of a no-SE block rides in this kernel's tile load - only with RD_MIXER_DW=1: measured, the 54 tap loads + 216 fmas per
// lane and tile cost the mixer more (31 launches: 3.36 -> 5.44 ms per 32-page step) than the depthwise launches they replace (1.59 ms);
// profiles/r6_mixer_dw.txt
bool mixer_res_fuses_dw(int C) {
    static const bool on = [] { const char* e = getenv("RD_MIXER_DW"); return e && e[0] == '1'; }();
    return on && mixer_res_supported(C);
}

bool mixer_res_supported(int C) {
    static const bool off = [] { const char* e = getenv("RD_MIXER_RES"); return e && e[0] == '0'; }();
    return !off && (C == 64 || C == 96);       // multiples of 32 whose image fits in LDS (C = 128: 256 KB)
}

// w1 [2C][C], w2 [C][2C] (BN folded) -> fragment image (layout above).  A fragment of v_mfma_f32_16x16x32_f16: lane l holds row
// l % 16, k = 8 (l / 16) + e.  W2's k-slot e of k-step q is hidden unit 16 (2q + e / 4) + 4 (l / 16) + e % 4 (the C/D registers of
// the two hidden blocks, see the kernel).
// Each matrix is pre-scaled by a power of two (max |w| * s in [2^13, 2^14)) and split into hi = fp16(w s), lo = fp16(w s - hi);
// inv[0] = 1 / s(W1), inv[1] = 1 / s(W2): what the kernel multiplies its sums by (MixerParams ws_inv1 / ws_inv2).
void prepare_mixer_weights_res(const float* w1, const float* w2, int C, std::vector<uint16_t>& img, float inv[2]) {
    const int H2 = 2 * C, KS1 = C / 32, HB = H2 / 16, NQ = H2 / 32, OB = C / 16;
    const int F1 = HB * KS1, F2 = OB * NQ;
    img.assign((size_t)2 * (F1 + F2) * 512, 0);
    auto scale_of = [](const float* w, size_t n) {
        float mx = 0.f;
        for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
        if (!(mx > 0.f) || !(mx < INFINITY)) return 1.f;
        int e = 0;
        (void)std::frexp(mx, &e);
        return std::ldexp(1.f, 14 - e);
    };
    const float s1 = scale_of(w1, (size_t)H2 * C), s2 = scale_of(w2, (size_t)H2 * C);
    inv[0] = 1.f / s1;
    inv[1] = 1.f / s2;
    float sc = s1;
    auto put = [&sc](float v, uint16_t& hb, uint16_t& lb) {
        const float vs = v * sc;
        const _Float16 h = (_Float16)vs;
        const _Float16 l = (_Float16)(vs - (float)h);
        __builtin_memcpy(&hb, &h, 2);
        __builtin_memcpy(&lb, &l, 2);
    };
    uint16_t* w1h = img.data();
    uint16_t* w1l = w1h + (size_t)F1 * 512;
    uint16_t* w2h = w1l + (size_t)F1 * 512;
    uint16_t* w2l = w2h + (size_t)F2 * 512;
    for (int hb = 0; hb < HB; ++hb)
        for (int ks = 0; ks < KS1; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    // k-slot e of lane group g = l / 16: channel 16 (2 ks + e / 4) + 4 g + e % 4 (see the kernel's tile load)
                    const int row = 16 * hb + (l & 15), k = 16 * (2 * ks + (e >> 2)) + 4 * (l >> 4) + (e & 3);
                    const size_t o = ((size_t)(hb * KS1 + ks) * 64 + l) * 8 + e;
                    put(w1[(size_t)row * C + k], w1h[o], w1l[o]);
                }
    sc = s2;
    for (int ob = 0; ob < OB; ++ob)
        for (int q = 0; q < NQ; ++q)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int row = 16 * ob + (l & 15);
                    const int hid = 16 * (2 * q + (e >> 2)) + 4 * (l >> 4) + (e & 3);
                    const size_t o = ((size_t)(ob * NQ + q) * 64 + l) * 8 + e;
                    put(w2[(size_t)row * H2 + hid], w2h[o], w2l[o]);
                }
}

template <int C>
static void launch_res(const MixerParams& p, hipStream_t s) {
    using G = ResGeom<C>;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return rd_cu_budget(n > 0 ? n : 256);
    }();
    const int n_tiles = (p.M + 15) / 16;
    const int n_wg = (n_tiles + MR_WAVES - 1) / MR_WAVES;
    const int grid = n_wg < n_cu ? n_wg : n_cu;
    const unsigned char* img = reinterpret_cast<const unsigned char*>(p.w1h);
    static unsigned long long ok0 = 0, ok1 = 0, ok2 = 0, ok3 = 0;
    static const bool rs = [] { const char* e = getenv("RD_RES_RS"); return !(e && e[0] == '0'); }();     // A/B switch: RD_RES_RS=0 = residual re-read (round 2-4)
    if (p.dw_w) {            // (the builder only asks for it without a gate; the residual then always comes from the fragments)
        static unsigned long long okd = 0;
        rd_allow_dynamic_lds((const void*)lc_mixer_res_kernel<C, false, true, true>, G::LDS_BYTES_DW, okd);
        hipLaunchKernelGGL((lc_mixer_res_kernel<C, false, true, true>), dim3(grid), dim3(1024), G::LDS_BYTES_DW, s, p, img, n_tiles);
    } else if (p.gate && rs) {
        rd_allow_dynamic_lds((const void*)lc_mixer_res_kernel<C, true, true>, G::LDS_BYTES, ok3);
        hipLaunchKernelGGL((lc_mixer_res_kernel<C, true, true>), dim3(grid), dim3(1024), G::LDS_BYTES, s, p, img, n_tiles);
    } else if (rs) {
        rd_allow_dynamic_lds((const void*)lc_mixer_res_kernel<C, false, true>, G::LDS_BYTES, ok2);
        hipLaunchKernelGGL((lc_mixer_res_kernel<C, false, true>), dim3(grid), dim3(1024), G::LDS_BYTES, s, p, img, n_tiles);
    } else if (p.gate) {
        rd_allow_dynamic_lds((const void*)lc_mixer_res_kernel<C, true, false>, G::LDS_BYTES, ok1);
        hipLaunchKernelGGL((lc_mixer_res_kernel<C, true, false>), dim3(grid), dim3(1024), G::LDS_BYTES, s, p, img, n_tiles);
    } else {
        rd_allow_dynamic_lds((const void*)lc_mixer_res_kernel<C, false, false>, G::LDS_BYTES, ok0);
        hipLaunchKernelGGL((lc_mixer_res_kernel<C, false, false>), dim3(grid), dim3(1024), G::LDS_BYTES, s, p, img, n_tiles);
    }
}

// p.w1h = the fragment image of prepare_mixer_weights_res
void launch_mixer_fused_res(const MixerParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    if (p.C == 96) launch_res<96>(p, s);
    else launch_res<64>(p, s);
}

}  // namespace rd
