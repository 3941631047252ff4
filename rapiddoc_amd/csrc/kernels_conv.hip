// Dense convolution as implicit GEMM on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Replaces what onnxruntime / torch do inside `session(...)` for every dense conv / linear layer of
// PP-OCRv6 det+rec and PPHGNetV2 (reference definitions: rec_lcnetv4.py:87-118, det_db_head.py:52-93,
// rec_pphgnetv2.py:860-918, necks/rnn.py:252-299).  fp32-in / fp32-accumulate MFMA is bit-for-bit an
// fmaf chain, which is what the <=1e-3 parity bar of BASELINE.json asks for.
//
// Tiling (wave64, CDNA4):
//   * block = 256 threads = 4 waves, block tile BM x BN (BM = 128, BN in {32,64,96,128}), K step 32
//   * A = im2col(X) gathered on the fly from NHWC (ci fastest => 16-byte loads), B = W[Ng][K] (the
//     natural [Cout][kh][kw][Cin] weight layout): both operands are K-contiguous, staged through LDS
//     as [row][36] (32 + 4 pad floats) so that every ds_read_b128 of an MFMA operand is conflict-free
//   * one ds_read_b128 feeds 4 MFMAs: lane l holds k = 4*(l>>5)+e of its row for e = 0..3
//   * register-staged prefetch of the next K tile while the MFMAs of the current one run
//   * epilogue fused: + bias (BN folded), activation, + residual, strided (channel-slice) store,
//     or the 2x2/stride-2 transposed-conv scatter
//   * 1-D grid with a bijective XCD swizzle so that the n-tiles sharing an A panel run on one XCD (L2)
#include "rd_device.h"

#include <cstdlib>

namespace rd {


static constexpr int BK = 32;
static constexpr int BKP = 36;

template <int BM, int BN, int WM, int WN, bool IS1X1>
__global__ void __launch_bounds__(WM* WN * 64) conv_igemm_kernel(ConvParams p, int ntn) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AL = BM * 8 / NT, BL = BN * 8 / NT;
    static_assert(AL >= 1 && BL >= 1 && NT == 256, "tile/loader mismatch");

    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * BKP];
    float* As = smem;
    float* Bs = smem + BM * BKP;

    // bijective XCD-aware remap of the linear block id (cdna_hip_programming.md section 5, T1)
    int tile_m, tile_n;
    {
        const int nwg = gridDim.x, id = blockIdx.x;
        const int xcd = id & 7, j = id >> 3, q = nwg >> 3, r = nwg & 7;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        tile_m = w / ntn;
        tile_n = w - tile_m * ntn;
    }
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int tid = threadIdx.x;
    const int lrow = tid >> 3, lkq = tid & 7;
    const int K = p.K;

    // ---- per-thread A row bookkeeping (fixed over the K loop).  Invalid rows / taps / K-tail quads are redirected to
    // a safe address and zeroed after the load, so that every global load of a K tile is issued unconditionally and
    // back to back (a conditional load is fenced by its own s_waitcnt and serialises the whole tile fetch).
    const float* arow[AL];
    bool avalid[AL];
    int a_ih0[AL], a_iw0[AL];
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + lrow + 32 * i;
        avalid[i] = m < p.M;
        const int mm = avalid[i] ? m : 0;
        if (IS1X1) {
            arow[i] = p.x + (size_t)mm * p.xld;
            a_ih0[i] = a_iw0[i] = 0;
        } else {
            const int ohw = p.OH * p.OW;
            const int b = mm / ohw, rem = mm - b * ohw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            a_ih0[i] = avalid[i] ? oh * p.SH - p.PT : -(1 << 28);
            a_iw0[i] = ow * p.SW - p.PL;
            arow[i] = p.x + (size_t)b * p.H * p.W * p.xld;
        }
    }
    const float* brow[BL];
    bool bvalid[BL];
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int n = n0 + lrow + 32 * i;
        bvalid[i] = n < p.Ng;
        brow[i] = p.w + (size_t)(bvalid[i] ? n : 0) * K;
    }

    f32x4 areg[AL], breg[BL];
    unsigned amask = 0, bmask = 0;  // validity of the quads in flight; applied when they are written to LDS, so that
                                    // nothing consumes a load result (and waits for it) before the MFMAs of this tile
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    auto load_tiles = [&](int k0) {
        const int k = k0 + 4 * lkq;
        const bool kvalid = k < K;
        const int kk = kvalid ? k : 0;
        amask = bmask = 0;
        if (IS1X1) {
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                areg[i] = *reinterpret_cast<const f32x4*>(arow[i] + kk);
                amask |= (unsigned)(kvalid && avalid[i]) << i;
            }
        } else {
            const int tap = kk / p.Cin, ci = kk - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int ih = a_ih0[i] + kh, iw = a_iw0[i] + kw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const size_t off = ok ? ((size_t)ih * p.W + iw) * p.xld + ci : 0;
                areg[i] = *reinterpret_cast<const f32x4*>(arow[i] + off);
                amask |= (unsigned)ok << i;
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            breg[i] = *reinterpret_cast<const f32x4*>(brow[i] + kk);
            bmask |= (unsigned)(kvalid && bvalid[i]) << i;
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < AL; ++i)
            *reinterpret_cast<f32x4*>(&As[(lrow + 32 * i) * BKP + 4 * lkq]) = ((amask >> i) & 1u) ? areg[i] : zero4;
#pragma unroll
        for (int i = 0; i < BL; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(lrow + 32 * i) * BKP + 4 * lkq]) = ((bmask >> i) & 1u) ? breg[i] : zero4;
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int KT = (K + BK - 1) / BK;
    load_tiles(0);
    store_tiles();
    __syncthreads();
    for (int kt = 0; kt < KT; ++kt) {
        const bool more = kt + 1 < KT;
        if (more) load_tiles((kt + 1) * BK);
        const int kleft = K - kt * BK;
        const int ng = kleft >= BK ? 4 : (kleft + 7) >> 3;
        const float* ap = &As[((wm * TM) * 32 + l31) * BKP + 4 * lhi];
        const float* bp = &Bs[((wn * TN) * 32 + l31) * BKP + 4 * lhi];
        auto mma_group = [&](int g) {
            f32x4 af[TM], bf[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) af[i] = *reinterpret_cast<const f32x4*>(ap + i * 32 * BKP + g * 8);
#pragma unroll
            for (int j = 0; j < TN; ++j) bf[j] = *reinterpret_cast<const f32x4*>(bp + j * 32 * BKP + g * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i][e], bf[j][e], acc[i][j], 0, 0, 0);
        };
        // (software-pipelined fragment reads via sched_group_barrier and a plain full unroll both measured 8-15 %
        //  slower than hipcc's own schedule of this loop at 3 waves/SIMD)
        for (int g = 0; g < ng; ++g) mma_group(g);
        __syncthreads();
        if (more) {
            store_tiles();
            __syncthreads();
        }
    }

    // ---- fused epilogue
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = n0 + (wn * TN + j) * 32 + l31;
        if (n >= p.Ng) continue;
        int co = n, dy = 0, dx = 0;
        if (p.out_mode == OUT_DECONV2X2) {
            const int tap = n / p.Cout;
            co = n - tap * p.Cout;
            dy = tap >> 1;
            dx = tap & 1;
        }
        const float bv = p.bias ? p.bias[co] : 0.f;
#pragma unroll
        for (int i = 0; i < TM; ++i) {
            const int mb = m0 + (wm * TM + i) * 32 + 4 * lhi;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) continue;
                float v = rd_act(acc[i][j][r] + bv, p.act);
                if (p.out_mode == OUT_NHWC) {
                    if (p.res) v += p.res[(size_t)m * p.rld + co];
                    p.y[(size_t)m * p.yld + co] = v;
                } else {
                    const int b = m / ohw, rem = m - b * ohw;
                    const int oh = rem / p.OW, ow = rem - oh * p.OW;
                    const size_t pix = ((size_t)b * (2 * p.OH) + 2 * oh + dy) * (2 * p.OW) + 2 * ow + dx;
                    p.y[pix * p.yld + co] = v;
                }
            }
        }
    }
}

struct TileCfg { int bn; const char* name; };

static inline TileCfg pick_cfg(const ConvParams& p) {
    const int n = p.Ng;
    if (n <= 32) return {32, "128x32"};
    if (n <= 64) return {64, "128x64"};
    if (n <= 96) return {96, "128x96"};
    // fewest padded columns; ties go to the wider tile
    const int w128 = (n + 127) / 128 * 128, w96 = (n + 95) / 96 * 96;
    if (w96 < w128) return {96, "128x96"};
    return {128, "128x128"};
}

static inline bool is_1x1(const ConvParams& p) {
    return p.KH == 1 && p.KW == 1 && p.SH == 1 && p.SW == 1 && p.PT == 0 && p.PL == 0 && p.OH == p.H && p.OW == p.W;
}

const char* conv_igemm_config_name(const ConvParams& p) { return pick_cfg(p).name; }

template <int BM, int BN, int WM, int WN>
static void launch_cfg(const ConvParams& p, hipStream_t s) {
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.Ng + BN - 1) / BN;
    dim3 grid(ntm * ntn), block(WM * WN * 64);
    if (is_1x1(p))
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, true>), grid, block, 0, s, p, ntn);
    else
        hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, false>), grid, block, 0, s, p, ntn);
}

// --------------------------------------------------------------------------------------------------
// Skinny GEMM for M <= 32 rows (autoregressive decode steps: M = number of sequences).  A 128-row MFMA tile would
// spend >= 75 % of its matrix-core time on padding and leave the chip with N/128 workgroups; this kernel is a
// weight-streaming design instead: a workgroup owns a slice of output columns, X[M][K] is staged through LDS in
// K chunks of 512 and shared by its 4 wavefronts, each wavefront owns whole columns (lanes split K, 16-byte
// coalesced weight loads, butterfly reduction), summation order is fixed (deterministic).
// --------------------------------------------------------------------------------------------------
template <int MT>
__global__ void __launch_bounds__(256) skinny_gemm_kernel(ConvParams p, int cols_per_wave) {
    constexpr int KC = 512;
    __shared__ __attribute__((aligned(16))) float xs[MT * KC];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_base = (blockIdx.x * 4 + wave) * cols_per_wave;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    for (int c0 = 0; c0 < cols_per_wave; c0 += 2) {
        // (all waves execute the same number of c0 iterations and K chunks: the barriers below are uniform)
        const int n0 = n_base + c0, n1 = n0 + 1;
        const bool v0 = n0 < p.Ng, v1 = n1 < p.Ng && c0 + 1 < cols_per_wave;
        float acc0[MT], acc1[MT];
#pragma unroll
        for (int m = 0; m < MT; ++m) acc0[m] = acc1[m] = 0.f;
        for (int k0 = 0; k0 < p.K; k0 += KC) {
            __syncthreads();
            for (int i = tid; i < MT * (KC / 4); i += 256) {
                const int m = i / (KC / 4), q = i - m * (KC / 4);
                const int k = k0 + 4 * q;
                *reinterpret_cast<f32x4*>(&xs[m * KC + 4 * q]) =
                    (m < p.M && k < p.K) ? *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.xld + k) : zero4;
            }
            __syncthreads();
            if (p.ln_g) {  // fused pre-LayerNorm: K <= KC, so the chunk holds whole rows; wave w normalises rows w, w+4, ...
                for (int m = wave; m < MT; m += 4) {
                    float v[KC / 64], s1 = 0.f;
#pragma unroll
                    for (int i = 0; i < KC / 64; ++i) {
                        const int k = lane + 64 * i;
                        v[i] = k < p.K ? xs[m * KC + k] : 0.f;
                        s1 += v[i];
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
                    const float mean = s1 / p.K;
                    float s2 = 0.f;
#pragma unroll
                    for (int i = 0; i < KC / 64; ++i) {
                        const int k = lane + 64 * i;
                        const float d = k < p.K ? v[i] - mean : 0.f;
                        s2 += d * d;
                    }
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
                    const float rstd = rsqrtf(s2 / p.K + 1e-5f);
#pragma unroll
                    for (int i = 0; i < KC / 64; ++i) {
                        const int k = lane + 64 * i;
                        if (k < p.K) xs[m * KC + k] = (v[i] - mean) * rstd * p.ln_g[k] + p.ln_b[k];
                    }
                }
                __syncthreads();
            }
#pragma unroll
            for (int h = 0; h < KC / 256; ++h) {
                const int kq = h * 256 + 4 * lane, k = k0 + kq;
                const bool kv = k < p.K;
                const f32x4 w0 = (v0 && kv) ? *reinterpret_cast<const f32x4*>(p.w + (size_t)n0 * p.K + k) : zero4;
                const f32x4 w1 = (v1 && kv) ? *reinterpret_cast<const f32x4*>(p.w + (size_t)n1 * p.K + k) : zero4;
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs[m * KC + kq]);
                    acc0[m] += xv[0] * w0[0] + xv[1] * w0[1] + xv[2] * w0[2] + xv[3] * w0[3];
                    acc1[m] += xv[0] * w1[0] + xv[1] * w1[1] + xv[2] * w1[2] + xv[3] * w1[3];
                }
            }
        }
#pragma unroll
        for (int m = 0; m < MT; ++m) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                acc0[m] += __shfl_xor(acc0[m], o, 64);
                acc1[m] += __shfl_xor(acc1[m], o, 64);
            }
        }
        // lane m writes row m of the two columns
        float r0 = 0.f, r1 = 0.f;
#pragma unroll
        for (int m = 0; m < MT; ++m) {
            if (lane == m) { r0 = acc0[m]; r1 = acc1[m]; }
        }
        if (lane < p.M) {
            if (v0) {
                float v = rd_act(r0 + (p.bias ? p.bias[n0] : 0.f), p.act);
                if (p.res) v += p.res[(size_t)lane * p.rld + n0];
                p.y[(size_t)lane * p.yld + n0] = v;
            }
            if (v1) {
                float v = rd_act(r1 + (p.bias ? p.bias[n1] : 0.f), p.act);
                if (p.res) v += p.res[(size_t)lane * p.rld + n1];
                p.y[(size_t)lane * p.yld + n1] = v;
            }
        }
    }
}

// Round 3: the decode step of the formula head is ~50 DEPENDENT launches of this kernel on 1 - 4 MB of weights each, i.e. every
// launch is a latency chain (X staging -> LayerNorm -> weight loads -> FMAs -> reduction -> store), not a bandwidth problem
// (profiles/r3_formula_decode.txt: 11.6 us average, 6.3 us for 1 MB of weights).  skinny2 shortens the chain:
//   * a wavefront issues ALL weight loads of its first K pass (CW columns x KPL float4 per lane) BEFORE it touches X, so the
//     HBM round trip runs under the staging + LayerNorm of X instead of behind it;
//   * X (all of K) is staged and normalised ONCE per workgroup (the old kernel re-staged it for every column pair);
//   * CW x MT accumulators are reduced over the 64 lanes by a halving butterfly (each exchange step halves the values a lane
//     carries: NV + 6 shuffles instead of 6 NV), fixed order, deterministic;
//   * wider column groups per wavefront (CW = 32 / MT): 8 - 16 independent 16-byte loads in flight per lane.
template <int MT, int CW, int KPL>
__global__ void __launch_bounds__(256) skinny2_gemm_kernel(ConvParams p, int kpad) {
    constexpr int NV = MT * CW, KP = 256 * KPL;
    extern __shared__ __attribute__((aligned(16))) float xs2[];          // [MT][kpad]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_base = (blockIdx.x * 4 + wave) * CW;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const int passes = kpad / KP;
    f32x4 wr[CW][KPL];
    auto load_w = [&](int pass) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int n = min(n_base + c, p.Ng - 1);
            // unconditional loads from clamped addresses, masked afterwards: a select on a conditional load makes the compiler wait for
            // each load before it issues the next (the ISA of the first version had CW * KPL `global_load ; s_waitcnt vmcnt(0)` pairs)
#pragma unroll
            for (int h = 0; h < KPL; ++h) {
                const int k = pass * KP + h * 256 + 4 * lane;
                wr[c][h] = *reinterpret_cast<const f32x4*>(p.w + (size_t)n * p.K + min(k, p.K - 4));
            }
#pragma unroll
            for (int h = 0; h < KPL; ++h)
                if (pass * KP + h * 256 + 4 * lane >= p.K) wr[c][h] = zero4;
        }
    };
    load_w(0);
    // X -> LDS (zero padded to kpad), then the fused pre-LayerNorm: wave w normalises rows w, w + 4, ...
    for (int i = tid; i < MT * (kpad / 4); i += 256) {
        const int m = i / (kpad / 4), k = 4 * (i - m * (kpad / 4));
        *reinterpret_cast<f32x4*>(&xs2[m * kpad + k]) =
            (m < p.M && k < p.K) ? *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.xld + k) : zero4;
    }
    __syncthreads();
    if (p.ln_g) {          // (K <= 512 here: launch_skinny)
        for (int m = wave; m < MT; m += 4) {
            float v[8], s1 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lane + 64 * i;
                v[i] = k < p.K ? xs2[m * kpad + k] : 0.f;
                s1 += v[i];
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
            const float mean = s1 / p.K;
            float s2 = 0.f;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lane + 64 * i;
                const float d = k < p.K ? v[i] - mean : 0.f;
                s2 += d * d;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
            const float rstd = rsqrtf(s2 / p.K + 1e-5f);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int k = lane + 64 * i;
                if (k < p.K) xs2[m * kpad + k] = (v[i] - mean) * rstd * p.ln_g[k] + p.ln_b[k];
            }
        }
        __syncthreads();
    }
    float acc[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) acc[i] = 0.f;
    for (int pass = 0; pass < passes; ++pass) {
        if (pass) load_w(pass);
#pragma unroll
        for (int h = 0; h < KPL; ++h) {
            const int kq = pass * KP + h * 256 + 4 * lane;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f32x4 xv = *reinterpret_cast<const f32x4*>(&xs2[m * kpad + kq]);
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    acc[c * MT + m] += xv[0] * wr[c][h][0] + xv[1] * wr[c][h][1] + xv[2] * wr[c][h][2] + xv[3] * wr[c][h][3];
            }
        }
    }
    // halving butterfly: after the step with offset o a lane carries half as many values; a lane whose bit o is set keeps the
    // upper half.  When one value is left the remaining steps are plain sums (those lanes end up with copies).
    int cnt = NV, idx = 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        if (cnt > 1) {
            cnt >>= 1;
            const bool up = (lane & o) != 0;
#pragma unroll
            for (int j = 0; j < NV / 2; ++j) {
                if (j < cnt) {
                    const float lo = acc[j], hi = acc[j + cnt];
                    const float send = up ? lo : hi, keep = up ? hi : lo;
                    acc[j] = keep + __shfl_xor(send, o, 64);
                }
            }
            if (up) idx += cnt;
        } else {
            acc[0] += __shfl_xor(acc[0], o, 64);
        }
    }
    // value idx = c * MT + m; with NV < 64 the lanes that differ only in the low (plain-sum) bits hold copies: the lowest writes
    constexpr int LOW = NV >= 64 ? 0 : (64 / NV - 1);
    const int c = idx / MT, m = idx - c * MT, n = n_base + c;
    if ((lane & LOW) == 0 && m < p.M && n < p.Ng) {
        float v = rd_act(acc[0] + (p.bias ? p.bias[n] : 0.f), p.act);
        if (p.res) v += p.res[(size_t)m * p.rld + n];
        p.y[(size_t)m * p.yld + n] = v;
    }
}

template <int MT, int CW, int KPL>
static void launch_skinny2(const ConvParams& p, hipStream_t s) {
    const int kpad = (p.K + 256 * KPL - 1) / (256 * KPL) * (256 * KPL);
    const size_t lds = (size_t)MT * kpad * sizeof(float);
    static unsigned long long lds_ok = 0;
    rd_allow_dynamic_lds((const void*)skinny2_gemm_kernel<MT, CW, KPL>, lds, lds_ok);
    const int blocks = (p.Ng + 4 * CW - 1) / (4 * CW);
    hipLaunchKernelGGL((skinny2_gemm_kernel<MT, CW, KPL>), dim3(blocks), dim3(256), lds, s, p, kpad);
}

static void launch_skinny(const ConvParams& p, hipStream_t s) {
    static const bool v2 = [] { const char* e = getenv("RD_SKINNY2"); return !(e && e[0] == '0'); }();
    if (v2 && p.K <= 2048 && (!p.ln_g || p.K <= 512)) {
        const bool wide = p.K > 512;        // K passes of 1024 (fc2: K = 2048) instead of 512
        if (p.M <= 8) { if (wide) launch_skinny2<8, 4, 4>(p, s); else launch_skinny2<8, 4, 2>(p, s); return; }
        if (p.M <= 16) { if (wide) launch_skinny2<16, 2, 4>(p, s); else launch_skinny2<16, 2, 2>(p, s); return; }
        if (!wide) { launch_skinny2<32, 2, 2>(p, s); return; }       // (M = 32, K = 2048: 256 KB of X - the chunked kernel below)
    }
    // columns per wavefront: enough workgroups to fill the chip, not so many that X is re-staged excessively
    int cpw = 2;
    while ((p.Ng + 4 * cpw - 1) / (4 * cpw) > 2048 && cpw < 16) cpw *= 2;
    const int blocks = (p.Ng + 4 * cpw - 1) / (4 * cpw);
    if (p.M <= 8) hipLaunchKernelGGL(skinny_gemm_kernel<8>, dim3(blocks), dim3(256), 0, s, p, cpw);
    else if (p.M <= 16) hipLaunchKernelGGL(skinny_gemm_kernel<16>, dim3(blocks), dim3(256), 0, s, p, cpw);
    else hipLaunchKernelGGL(skinny_gemm_kernel<32>, dim3(blocks), dim3(256), 0, s, p, cpw);
}

bool skinny_gemm_applies(int M, int K) { return M <= 32 && K % 4 == 0 && K >= 64 && K <= 512; }

void launch_conv_igemm(const ConvParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    if (p.allow_skinny && p.M <= 32 && p.out_mode == OUT_NHWC && is_1x1(p) && p.K % 4 == 0 && p.K >= 64 && (!p.ln_g || p.K <= 512)) {
        launch_skinny(p, s);
        return;
    }
    // (ln_g is only honoured by the skinny path; callers check skinny_gemm_applies() before relying on it)
    switch (pick_cfg(p).bn) {
        case 32: launch_cfg<128, 32, 4, 1>(p, s); break;
        case 64: launch_cfg<128, 64, 4, 1>(p, s); break;
        case 96: launch_cfg<128, 96, 4, 1>(p, s); break;
        default: launch_cfg<128, 128, 2, 2>(p, s); break;
    }
}

// --------------------------------------------------------------------------------------------------
// Stem: 3x3 stride-2 pad-1 conv on the caller's NCHW fp32 image (Cin = 3), BN folded, ReLU.
// HBM-bound (27 MACs per output value); one thread = one output pixel x CO_T output channels.
// --------------------------------------------------------------------------------------------------
template <int CO_T>
__global__ void __launch_bounds__(256) stem_conv3x3s2_kernel(StemParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsm[];  // [27][Cout] + bias[Cout]
    const int nw = 27 * p.Cout;
    for (int i = threadIdx.x; i < nw + p.Cout; i += 256) wsm[i] = i < nw ? p.w[i] : (p.bias ? p.bias[i - nw] : 0.f);
    __syncthreads();
    const int groups = p.Cout / CO_T;
    const long total = (long)p.N * p.OH * p.OW * groups;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int g = idx % groups;
        long pix = idx / groups;
        const int ow = pix % p.OW;
        pix /= p.OW;
        const int oh = pix % p.OH;
        const int b = pix / p.OH;
        float acc[CO_T];
#pragma unroll
        for (int c = 0; c < CO_T; ++c) acc[c] = wsm[nw + g * CO_T + c];
        const float* xb = p.x + (size_t)b * p.in_ch * p.H * p.W;
        const size_t cstride = p.in_ch == 1 ? 0 : (size_t)p.H * p.W;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = oh * 2 - 1 + kh;
            if ((unsigned)ih >= (unsigned)p.H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = ow * 2 - 1 + kw;
                if ((unsigned)iw >= (unsigned)p.W) continue;
#pragma unroll
                for (int ci = 0; ci < 3; ++ci) {
                    const float xv = xb[ci * cstride + (size_t)ih * p.W + iw];
                    const float* wr = &wsm[((kh * 3 + kw) * 3 + ci) * p.Cout + g * CO_T];
#pragma unroll
                    for (int c = 0; c < CO_T; ++c) acc[c] = fmaf(xv, wr[c], acc[c]);
                }
            }
        }
        float* yo = p.y + (((size_t)b * p.OH + oh) * p.OW + ow) * p.yld + g * CO_T;
#pragma unroll
        for (int c = 0; c < CO_T; c += 4) {
            f32x4 v = {rd_act(acc[c], p.act), rd_act(acc[c + 1], p.act), rd_act(acc[c + 2], p.act),
                       rd_act(acc[c + 3], p.act)};
            *reinterpret_cast<f32x4*>(yo + c) = v;
        }
    }
}

void launch_stem_conv3x3s2(const StemParams& p, hipStream_t s) {
    constexpr int CO_T = 8;  // every stem here has Cout in {24, 32, 48}
    const long total = (long)p.N * p.OH * p.OW * (p.Cout / CO_T);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    const size_t sh = (size_t)(28 * p.Cout) * sizeof(float);
    hipLaunchKernelGGL(stem_conv3x3s2_kernel<CO_T>, dim3(blocks > 0 ? blocks : 1), dim3(256), sh, s, p);
}

}  // namespace rd
