// fp32-accurate implicit-GEMM convolution on the 2.5 PFLOP/s fp16 matrix cores ("h3" = 3 half-precision MFMAs per
// product).  Same geometry, fused epilogue and launch interface as kernels_conv.hip, different arithmetic:
//
//   every fp32 operand x is split as  x = hi + lo * 2^-11,  hi = fp16(x),  lo = fp16((x - hi) * 2^11)
//   (hi carries 11 significand bits, lo the next 11: |x - hi - lo*2^-11| <= 2^-22 |x|; the 2^11 scale keeps lo in
//   fp16's normal range), and            a*b ~= ah*bh + 2^-11 (ah*bl + al*bh)        (al*bl ~ 2^-22 |ab| dropped)
//   acc1 += ah*bh,  acc2 += ah*bl + al*bh   with v_mfma_f32_32x32x16_f16 (fp16 products are exact in fp32, fp32
//   accumulate);  result = acc1 + acc2 * 2^-11.   Per-product relative error <~ 5e-7, i.e. fp32 round-off class,
//   three orders of magnitude inside the 1e-3 parity budget, at 16/3 the fp32-MFMA rate.
// Operand range: |x| must stay below 65504 (fp16 max); tiny values degrade gracefully (absolute error < 2e-11).
//
// Activations are split on the fly while a K tile is staged into LDS (5 VALU ops per element, hidden under the
// MFMAs); weights are split once at load time (rd_load_weights) and stored as two fp16 matrices [Ng][Kp], Kp = K rounded up to 32.
#include <cstdlib>
#include <vector>

#include "rd_device.h"

namespace rd {


static constexpr int HBK = 32;    // K tile (fp32 elements)
static constexpr int HLD = 40;    // LDS row stride in halfs (64 B data + 16 B pad: conflict-free ds_read_b128)

__device__ __forceinline__ void split4(const f32x4 v, f16x4& hi, f16x4& lo) { rd_split4(v, hi, lo); }

template <int BM, int BN, int WM, int WN, bool IS1X1>
__global__ void __launch_bounds__(WM* WN * 64, (WM * WN == 4 && BM * BN >= 128 * 128) ? 2 : 1) conv_igemm_h3_kernel(ConvParams p, int ntn, int ntiles) {
    constexpr int NT = WM * WN * 64;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int AL = BM * 8 / NT, BL = BN * 8 / NT;
    static_assert(AL >= 1 && BL >= 1 && (BM * 8) % NT == 0 && (BN * 8) % NT == 0 && (NT == 256 || NT == 512), "tile/loader mismatch");

    // 8-wavefront tiles keep two LDS stages (one barrier per K tile instead of two); the 4-wavefront tiles stay
    // single-staged so that several workgroups fit a CU
    constexpr bool DB = NT == 512;
    constexpr int STAGE = 2 * (BM + BN) * HLD;
    __shared__ __attribute__((aligned(16))) _Float16 smem[(DB ? 2 : 1) * STAGE];
    constexpr int AH0 = 0, AL0 = BM * HLD, BH0 = 2 * BM * HLD, BL0 = 2 * BM * HLD + BN * HLD;

    // Output tiles are numbered so that XCD x (workgroup id % 8) owns one contiguous run of them: consecutive tiles share
    // their A rows, which then stay in that XCD's L2.  The 8-wavefront configurations run PERSISTENT workgroups (one
    // per CU, tiles v = id, id + grid, ...): the first K tile of the next output tile is requested before the epilogue
    // of the current one, so its latency and the store drain no longer sit between two workgroups on a CU whose LDS
    // only fits one of them.
    int m0 = 0, n0 = 0;
    auto map_tile = [&](int v) {
        const int xcd = v & 7, j = v >> 3, q = ntiles >> 3, r = ntiles & 7;
        const int w = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
        const int tile_m = w / ntn;
        m0 = tile_m * BM;
        n0 = (w - tile_m * ntn) * BN;
    };
    const int tid = threadIdx.x;
    const int lrow = tid >> 3, lkq = tid & 7;
    const int K = p.K, Kp = (p.K + 31) & ~31;

    // 32-bit element offsets from the uniform bases p.x / p.wh / p.wl instead of per-lane 64-bit pointers (the launcher
    // checks that the tensors stay below 2^32 elements): the 256x128 tile runs at the 256-VGPR limit
    unsigned arow[AL];
    bool avalid[AL];
    int a_hw0[AL];           // k x k: (first input row << 16) | (first input column & 0xffff), both signed 16-bit
    unsigned bsrc[BL];
    bool bvalid[BL], bwhich[BL];
    int b_lds[BL], b_col[BL];
    auto setup_tile = [&](int v) {
    map_tile(v);
#pragma unroll
    for (int i = 0; i < AL; ++i) {
        const int m = m0 + lrow + (NT / 8) * i;
        avalid[i] = m < p.M;
        const int mm = avalid[i] ? m : 0;
        if (IS1X1) {
            arow[i] = (unsigned)mm * (unsigned)p.xld;
            a_hw0[i] = 0;
        } else {
            const int ohw = p.OH * p.OW;
            const int b = mm / ohw, rem = mm - b * ohw;
            const int oh = rem / p.OW, ow = rem - oh * p.OW;
            const int ih0 = avalid[i] ? oh * p.SH - p.PT : -16384;     // rows past M: every tap lands outside the image
            a_hw0[i] = (ih0 << 16) | ((ow * p.SW - p.PL) & 0xffff);
            arow[i] = (unsigned)b * (unsigned)(p.H * p.W) * (unsigned)p.xld;
        }
    }
    // B loader: BN*8 16-byte pieces (hi then lo), piece t -> (which, row, 8-half column)
#pragma unroll
    for (int i = 0; i < BL; ++i) {
        const int t = tid + NT * i;
        const int which = t / (BN * 4), rem = t - which * (BN * 4);
        const int r = rem >> 2, c = rem & 3;
        const int n = n0 + r;
        bvalid[i] = n < p.Ng;
        bwhich[i] = which != 0;
        bsrc[i] = (unsigned)(bvalid[i] ? n : 0) * (unsigned)Kp;
        b_col[i] = c * 8;
        b_lds[i] = (which ? BL0 : BH0) + r * HLD + c * 8;
    }
    };
    int v = blockIdx.x;
    setup_tile(v);

    // staging registers of one K tile.  The 8-wavefront (persistent) configurations keep TWO tiles in flight: with one,
    // the K loop runs at one memory round trip (~1.8 us measured, 48 KB per CU in flight = 27 GB/s per CU) per K tile.
    struct Stage {
        f32x4 a[AL];
        u32x4 b[BL];
        unsigned am, bm;
    };
    Stage s0, s1;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const u32x4 zero4u = {0u, 0u, 0u, 0u};

    auto load_tiles = [&](int k0, Stage& st) {
        const int k = k0 + 4 * lkq;
        const bool kvalid = k < K;
        const int kk = kvalid ? k : 0;
        st.am = st.bm = 0;
        if (IS1X1) {
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                st.a[i] = *reinterpret_cast<const f32x4*>(p.x + (arow[i] + (unsigned)kk));
                st.am |= (unsigned)(kvalid && avalid[i]) << i;
            }
        } else {
            const int tap = kk / p.Cin, ci = kk - tap * p.Cin;
            const int kh = tap / p.KW, kw = tap - kh * p.KW;
#pragma unroll
            for (int i = 0; i < AL; ++i) {
                const int ih = (a_hw0[i] >> 16) + kh, iw = (int)(short)(a_hw0[i] & 0xffff) + kw;
                const bool ok = kvalid && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
                const unsigned off = ok ? (unsigned)(ih * p.W + iw) * (unsigned)p.xld + (unsigned)ci : 0u;
                st.a[i] = *reinterpret_cast<const f32x4*>(p.x + (arow[i] + off));
                st.am |= (unsigned)ok << i;
            }
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) {
            const int kb = k0 + b_col[i];
            const bool ok = kb < Kp && bvalid[i];
            st.b[i] = *reinterpret_cast<const u32x4*>(reinterpret_cast<const _Float16*>(bwhich[i] ? p.wl : p.wh) + (bsrc[i] + (unsigned)(ok ? kb : 0)));
            st.bm |= (unsigned)ok << i;
        }
    };
    auto store_tiles = [&](int buf, const Stage& sg) {
        _Float16* st = smem + buf * STAGE;
#pragma unroll
        for (int i = 0; i < AL; ++i) {
            f16x4 hi, lo;
            const f32x4 av = ((sg.am >> i) & 1u) ? sg.a[i] : zero4;
            split4(av, hi, lo);
            const int o = (lrow + (NT / 8) * i) * HLD + 4 * lkq;
            *reinterpret_cast<f16x4*>(&st[AH0 + o]) = hi;
            *reinterpret_cast<f16x4*>(&st[AL0 + o]) = lo;
        }
#pragma unroll
        for (int i = 0; i < BL; ++i) *reinterpret_cast<u32x4*>(&st[b_lds[i]]) = ((sg.bm >> i) & 1u) ? sg.b[i] : zero4u;
    };

    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const int l31 = lane & 31, lhi = lane >> 5;

    const int KT = (K + HBK - 1) / HBK;
    const int aoff = ((wm * TM) * 32 + l31) * HLD + 8 * lhi;
    const int boff = ((wn * TN) * 32 + l31) * HLD + 8 * lhi;
    unsigned emax = 0;
    load_tiles(0, s0);
    if (DB && KT > 1) load_tiles(HBK, s1);
    for (;;) {   // output tiles of this (persistent) workgroup
    f32x16 acc1[TM][TN], acc2[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][j][r] = acc2[i][j][r] = 0.f;
    auto compute = [&](int buf, int kt) {
        const _Float16* Ah = smem + buf * STAGE + AH0;
        const _Float16* Al = smem + buf * STAGE + AL0;
        const _Float16* Bh = smem + buf * STAGE + BH0;
        const _Float16* Bl = smem + buf * STAGE + BL0;
        const int kleft = K - kt * HBK;
        const int nks = kleft > 16 ? 2 : 1;
        for (int ks = 0; ks < nks; ++ks) {
            f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                ah[i] = *reinterpret_cast<const f16x8*>(&Ah[aoff + i * 32 * HLD + ks * 16]);
                al[i] = *reinterpret_cast<const f16x8*>(&Al[aoff + i * 32 * HLD + ks * 16]);
            }
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(&Bh[boff + j * 32 * HLD + ks * 16]);
                bl[j] = *reinterpret_cast<const f16x8*>(&Bl[boff + j * 32 * HLD + ks * 16]);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    acc1[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc1[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc2[i][j], 0, 0, 0);
                    acc2[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc2[i][j], 0, 0, 0);
                }
        }
    };
    store_tiles(0, s0);
    __syncthreads();
    if (DB) {
        // two LDS stages, two register stages: tile kt is multiplied from LDS while kt+1 is split into the other stage
        // and kt+2 is on its way from memory
        for (int kt = 0; kt < KT; kt += 2) {
            if (kt + 2 < KT) load_tiles((kt + 2) * HBK, s0);
            compute(0, kt);
            if (kt + 1 < KT) store_tiles(1, s1);
            __syncthreads();
            if (kt + 1 >= KT) break;
            if (kt + 3 < KT) load_tiles((kt + 3) * HBK, s1);
            compute(1, kt + 1);
            if (kt + 2 < KT) store_tiles(0, s0);
            __syncthreads();
        }
    } else {
        for (int kt = 0; kt < KT; ++kt) {
            const bool more = kt + 1 < KT;
            if (more) load_tiles((kt + 1) * HBK, s0);
            compute(0, kt);
            __syncthreads();
            if (more) {
                store_tiles(0, s0);
                __syncthreads();
            }
        }
    }

    // the next output tile's first K tile is requested now and lands while the epilogue below drains
    const int em0 = m0, en0 = n0;
    const int vnext = v + (int)gridDim.x;
    const bool has_next = vnext < ntiles;
    if (has_next) {
        setup_tile(vnext);
        load_tiles(0, s0);
        if (DB && KT > 1) load_tiles(HBK, s1);
    }
    // Range guard.  An activation beyond the fp16 range splits into hi = +-inf, which makes every output of its row
    // inf / NaN before the activation function - so the check sits here, on 32 outputs per thread, instead of in the
    // K loop (tracking max |a| there cost 35 % of the kernel's throughput).  Non-finite results that fp32 would also
    // produce only cost the caller a redundant fp32 re-run.
    const int ohw = p.OH * p.OW;
    // (only the persistent 256 x 64 tile: the 256 x 128 one sits at the register limit, and the 4-wavefront tiles live on occupancy -
    //  the second epilogue cost them 29 registers, i.e. a wavefront per SIMD)
    constexpr bool FAST_OK = BM == 256 && BN == 64;
    const bool interior_m = FAST_OK && p.fast_epi && em0 + BM <= p.M;      // (workgroup-uniform)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
        const int n = en0 + (wn * TN + j) * 32 + l31;
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
            const int mb = em0 + (wm * TM + i) * 32 + 4 * lhi;
            if constexpr (FAST_OK) if (p.out_mode == OUT_NHWC && interior_m) {
                // Interior row tile (round 5, as in kernels_gemm_h1.hip): no bounds test, every load / store a buffer access (descriptor at
                // the wavefront block's first row, lane offset in the vector operand, row in the scalar one): one instruction per store
                typedef __amdgpu_buffer_rsrc_t rsrc_t;
                const int wrow = __builtin_amdgcn_readfirstlane(em0 + (wm * TM + i) * 32);
                const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)wrow * p.yld, 0, 0x7fffffff, 0x00020000);
                const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res + (size_t)wrow * p.rld : p.y), 0, 0x7fffffff, 0x00020000);
                const unsigned yoff = ((unsigned)(4 * lhi) * (unsigned)p.yld + (unsigned)co) * 4u;
                const unsigned roff = ((unsigned)(4 * lhi) * (unsigned)p.rld + (unsigned)co) * 4u;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = half * 8 + e;
                        o[e] = fmaf(acc2[i][j][r], 1.f / 2048.f, acc1[i][j][r]) + bv;
                        emax = max(emax, __float_as_uint(o[e]) & 0x7fffffffu);
                    }
                    if (p.act == ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = rd_gelu(o[e]);
                    } else if (p.act == ACT_RELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
                    } else if (p.act != ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = rd_act(o[e], p.act);
                    }
                    if (p.res) {
                        float rs[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = half * 8 + e;
                            rs[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (int)roff, (int)((unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.rld * 4u), 0));
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += rs[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = half * 8 + e;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[e]), ry, (int)yoff, (int)((unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.yld * 4u), 0);
                    }
                }
                continue;
            }
            if (p.out_mode == OUT_NHWC) {
                // eight values at a time: the activation switch and the residual test stay outside the element loops and
                // the residual loads are unconditional (clamped row) and batched ahead of the stores - a load issued behind
                // a store otherwise waits for it (loads and stores share the vmcnt counter on this ISA)
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float o[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = half * 8 + e;
                        o[e] = fmaf(acc2[i][j][r], 1.f / 2048.f, acc1[i][j][r]) + bv;
                        emax = max(emax, __float_as_uint(o[e]) & 0x7fffffffu);
                    }
                    if (p.act == ACT_GELU) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = rd_gelu(o[e]);
                    } else if (p.act != ACT_NONE) {
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] = rd_act(o[e], p.act);
                    }
                    if (p.res) {
                        float rs[8];
#pragma unroll
                        for (int e = 0; e < 8; ++e) {
                            const int r = half * 8 + e;
                            rs[e] = p.res[(size_t)min(mb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.rld + co];
                        }
#pragma unroll
                        for (int e = 0; e < 8; ++e) o[e] += rs[e];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = half * 8 + e;
                        const int m = mb + (r & 3) + 8 * (r >> 2);
                        if (m < p.M) p.y[(size_t)m * p.yld + co] = o[e];
                    }
                }
                continue;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {          // 2x2 transposed convolution: scatter to the upsampled grid
                const int m = mb + (r & 3) + 8 * (r >> 2);
                if (m >= p.M) continue;
                const float pre = fmaf(acc2[i][j][r], 1.f / 2048.f, acc1[i][j][r]) + bv;
                emax = max(emax, __float_as_uint(pre) & 0x7fffffffu);
                const float v = rd_act(pre, p.act);
                const int b = m / ohw, rem = m - b * ohw;
                const int oh = rem / p.OW, ow = rem - oh * p.OW;
                const size_t pix = ((size_t)b * (2 * p.OH) + 2 * oh + dy) * (2 * p.OW) + 2 * ow + dx;
                p.y[pix * p.yld + co] = v;
            }
        }
    }
    if (!has_next) break;
    v = vnext;
    }
    if (emax >= 0x7f800000u && p.range_flag) rd_raise_flag(p.range_flag);
}

static inline int h3_pick_bn(const ConvParams& p) {
    static const int forced = [] { const char* e = getenv("RD_H3_BN"); return e ? atoi(e) : 0; }();
    if (forced) return forced;
    // Measured on MI355X (TF/s of fp32-equivalent work, M = 131072; fp32 MFMA kernel in brackets):
    //   K192/N384  K384/N768  K768/N384  K2176/N512  4096^2
    //     102        150        204        253        288     256x128 tile, 8 wavefronts (4x2), two LDS stages
    //      77        113        143        170        182     128x64 tile, 4 wavefronts
    //     [87]      [110]      [116]      [126]      [130]
    // The A tile is split once per K step whatever the tile width, so wide tiles amortise the split; narrow tiles are
    // for narrow layers only.
    const int n = p.Ng;
    if (n > 96) return 258;
    if (n > 32 && n <= 64 && p.M >= 65536) return 260;
    if (n > 64) return 96;
    if (n <= 32 || p.K <= 256) return 32;
    return 64;
}
static inline bool h3_is_1x1(const ConvParams& p) {
    return p.KH == 1 && p.KW == 1 && p.SH == 1 && p.SW == 1 && p.PT == 0 && p.PL == 0 && p.OH == p.H && p.OW == p.W;
}
template <int BM, int BN, int WM, int WN>
static void h3_launch_cfg(const ConvParams& p_in, hipStream_t s) {
    static const int fast_epi = [] { const char* e = getenv("RD_CONV_FAST_EPI"); return e && e[0] == '0' ? 0 : 1; }();
    ConvParams p = p_in;
    p.fast_epi = fast_epi;
    const int ntm = (p.M + BM - 1) / BM, ntn = (p.Ng + BN - 1) / BN, ntiles = ntm * ntn;
    // 8-wavefront tiles: persistent workgroups, one per CU (their two LDS stages leave room for one only)
    static const int n_cu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const bool persistent = WM * WN == 8;
    dim3 grid(persistent ? (ntiles < n_cu ? ntiles : n_cu) : ntiles), block(WM * WN * 64);
    if (h3_is_1x1(p))
        hipLaunchKernelGGL((conv_igemm_h3_kernel<BM, BN, WM, WN, true>), grid, block, 0, s, p, ntn, ntiles);
    else
        hipLaunchKernelGGL((conv_igemm_h3_kernel<BM, BN, WM, WN, false>), grid, block, 0, s, p, ntn, ntiles);
}

void launch_conv_igemm_h3(const ConvParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    if (conv_stream_h3_applies(p)) {   // small K (stems): HBM-bound, operands streamed from global memory at high occupancy
        launch_conv_stream_h3(p, s);
        return;
    }
    if (conv3x3_h1_applies(p)) {       // 3x3 stride 1 pad 1, <= 96 output channels: one accumulator set, two workgroups per CU (round 6)
        launch_conv3x3_h1(p, s);
        return;
    }
    if (conv_direct_h3_applies(p)) {   // 2x2 / 3x3 stride 1, <= 96 output channels: patch in LDS, taps as address offsets
        launch_conv_direct_h3(p, s);
        return;
    }
    if (gemm_h1_applies(p)) {       // pointwise, K % 32 == 0, wide: single-accumulator split, two workgroups per CU (kernels_gemm_h1.hip)
        launch_gemm_h1(p, s);
        return;
    }
    if (gemm_h3_dma_applies(p)) {   // pointwise, other K: the round-3 LDS-DMA pipeline (kernels_gemm_h3_dma.hip)
        launch_gemm_h3_dma(p, s);
        return;
    }
    // the staged kernels address X and the split weights with 32-bit element offsets; a tensor beyond 2^32 elements (16 GB of
    // fp32 activations in one layer) takes the fp32 MFMA kernel, which uses 64-bit addresses
    if ((unsigned long long)p.N * p.H * p.W * (unsigned long long)p.xld >= (1ull << 32) ||
        (unsigned long long)p.Ng * (unsigned long long)((p.K + 31) & ~31) >= (1ull << 32)) {
        launch_conv_igemm(p, s);
        return;
    }
    switch (h3_pick_bn(p)) {
        case 32: h3_launch_cfg<128, 32, 4, 1>(p, s); break;
        case 64: h3_launch_cfg<128, 64, 4, 1>(p, s); break;
        case 96: h3_launch_cfg<128, 96, 4, 1>(p, s); break;
        case 258: h3_launch_cfg<256, 128, 4, 2>(p, s); break;
        default: h3_launch_cfg<256, 64, 4, 2>(p, s); break;   // 260
    }
}

// host helper: split a weight matrix [rows][K] fp32 into hi/lo fp16 matrices [rows][Kp] (Kp = K rounded up to 32, zero filled:
// the LDS-DMA GEMM reads whole 32-wide K tiles of the weights and relies on the zeros past K)
void split_weights_h3(const float* w, int rows, int K, std::vector<uint16_t>& hi, std::vector<uint16_t>& lo) {
    const int Kp = (K + 31) & ~31;
    hi.assign((size_t)rows * Kp, 0);
    lo.assign((size_t)rows * Kp, 0);
    for (int r = 0; r < rows; ++r)
        for (int k = 0; k < K; ++k) {
            const float v = w[(size_t)r * K + k];
            const _Float16 h = (_Float16)v;
            const _Float16 l = (_Float16)((v - (float)h) * 2048.f);
            uint16_t hb, lb;
            __builtin_memcpy(&hb, &h, 2);
            __builtin_memcpy(&lb, &l, 2);
            hi[(size_t)r * Kp + k] = hb;
            lo[(size_t)r * Kp + k] = lb;
        }
}

}  // namespace rd
