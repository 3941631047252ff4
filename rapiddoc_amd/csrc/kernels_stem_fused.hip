// Fused stem front of PPLCNetV4 / PPHGNetV2 (rec_lcnetv4.py:148-169, rec_pphgnetv2.py:979-1056 StemBlock):
//
//     x [N,3,H,W] --stem1: conv3x3 s2 +BN+ReLU--> e [H/2,W/2,C1] --stem2a: conv2x2 (pad r/b) +ReLU--> a [C1/2]
//                                                    |                        --stem2b: conv2x2 (pad r/b) +ReLU--> b [C1]
//                                                    +--maxpool 2x2 s1 (pad r/b)--> p [C1]          cat = [p | b]  (2 C1 channels)
//
// as ONE kernel.  Round 2 ran it as four (stem_conv3x3s2 at 1.9 TB/s, two conv_stream_h3 at 2.1-3.5 TB/s, maxpool at 4.4 TB/s):
// 13 ms of a 101-ms step, all of it HBM round trips of half-resolution tensors with 12-48 channels - e alone (written once, read
// twice) is 160 MB per 64-line recogniser batch against 10 MB of input image.  Here a workgroup owns an 8 x 32 tile of `cat`:
// the image patch it needs (21 x 69 x 3) goes to LDS, e (10 x 34), a (9 x 33) live only in LDS as fp32 tiles, and the three
// convolutions run on the split-fp16 matrix cores straight from those tiles (A fragments = 8 consecutive channels of one tap of
// one pixel: two ds_read_b128, split in registers like the LDS-DMA GEMM does; pixel strides of 4 x odd floats make the
// fragment reads conflict-free).  stem1's K = 27 is laid out as 3 k-steps of 16 slots (one kernel row each: 9 taps x channels
// contiguous in the channel-interleaved patch, 7 zero-weight slots).  Halo recompute: 1.33x for stem1, 1.16x for stem2a.
// HBM traffic per tile: the patch + the cat tile.  Persistent workgroups; the next tile's patch is prefetched into registers.
// Arithmetic (round 5): the split of kernels_gemm_h1.hip - x = hi + lo with the low plane UNSCALED (gfx950's matrix cores keep fp16
// subnormals), each weight matrix pre-scaled by a power of two so that its low plane is a normal fp16 number (max |w| s in [2^13, 2^14)),
// the inverse scale applied in the epilogue; three MFMAs per product, fp32 accumulate.  1.5 VALU operations per split element instead
// of the 3.5 the scaled form compiled to here.  The `fp32` precision mode keeps the four separate fp32 kernels.
#include <cmath>
#include <cstdlib>
#include <vector>

#include "rd_device.h"

namespace rd {

static constexpr int SF_TH = 8, SF_TW = 32;                       // cat tile
static constexpr int SF_EH = SF_TH + 2, SF_EW = SF_TW + 2;        // e tile (stem2a needs +1, stem2b another +1)
static constexpr int SF_AH = SF_TH + 1, SF_AW = SF_TW + 1;        // a tile
static constexpr int SF_PH = 2 * SF_EH + 1, SF_PW = 2 * SF_EW + 1;   // image patch of the e tile (3x3, stride 2)
static constexpr int SF_PROW = 208;                               // floats per patch row (69 x 3 = 207, even for 8-byte reads)
static constexpr int SF_PATCH_FLOATS = SF_PH * SF_PROW + 16;      // + overrun of the last row's zero-weight slots
static constexpr int SF_NT = 512;

template <int C1>
struct SfGeom {
    static constexpr int CA = C1, NA = C1 / 2, CB = (NA + 7) / 8 * 8;
    static constexpr int K1 = 48, K2A = 4 * CA, K2B = 4 * CB;
    static constexpr int SE = CA + 4, SA = CB + 4;                 // fp32 pixel strides: (stride / 4) odd
    static_assert(((SE / 4) & 1) == 1 && ((SA / 4) & 1) == 1, "pixel strides must be 4 x odd floats");
    static constexpr int pad8(int k) { return ((k / 8) & 1) ? k : k + 8; }   // row stride in halfs with (stride / 8) odd
    static constexpr int R1 = pad8(K1), R2A = pad8(K2A), R2B = pad8(K2B);
    // weight image (halfs): [w1 hi | w1 lo | w2a hi | w2a lo | w2b hi | w2b lo], rows = output channels
    static constexpr int W1_H = C1 * R1, W2A_H = NA * R2A, W2B_H = C1 * R2B;
    static constexpr int W_HALFS = 2 * (W1_H + W2A_H + W2B_H);
    static constexpr int IMG_HALFS = W_HALFS + 16;               // + six floats: the inverse scales and the scales of w1 / w2a / w2b
    static constexpr int BIAS_FLOATS = C1 + NA + C1;
    // (both tiles are padded to whole 32-pixel MFMA blocks: the last block's rows beyond the tile land in the pad, so that every
    //  block stores through the same unconditional epilogue)
    static constexpr int E_FLOATS = (SF_EH * SF_EW + 31) / 32 * 32 * SE;
    static constexpr int A_FLOATS_RAW = (SF_AH * SF_AW + 31) / 32 * 32 * SA;
    static constexpr int A_FLOATS = A_FLOATS_RAW > SF_PATCH_FLOATS ? A_FLOATS_RAW : SF_PATCH_FLOATS;   // the patch shares this region
    static constexpr size_t LDS_BYTES = (size_t)W_HALFS * 2 + (size_t)(BIAS_FLOATS + E_FLOATS + A_FLOATS) * 4 + 64;
    static constexpr int MB1 = (SF_EH * SF_EW + 31) / 32, NB1 = (C1 + 31) / 32;
    static constexpr int MB2 = (SF_AH * SF_AW + 31) / 32;
    static constexpr int MB3 = SF_TH * SF_TW / 32, NB3 = (C1 + 31) / 32;
    static_assert(NA <= 32 && MB3 == 8, "one N block for stem2a, one M block per wavefront for stem2b");
};

struct StemFusedParams {
    const float* x; int N, H, W, in_ch;       // NCHW image (in_ch 1: the grey plane is read three times)
    const uint16_t* wimg;                       // prepare_stem_fused_weights
    const float* bias;                          // [C1 | C1/2 | C1]
    float* y; int yld;                          // cat NHWC [N][H2][W2][2 C1]: p at channel 0, b at channel C1
    int H2, W2, tiles_x, tiles_y;
    unsigned* range_flag;
    const int32_t* line_tab;                    // optional per-image widths (rd_kernels.h LineTab): image n is line_tab[4 n + 1] e-columns wide
    int dbg;                                    // developer: RD_STEM_DBG ablation bits (results garbage): 1 no stem1, 2 no stem2a, 4 no pool, 8 no stem2b, 16 no patch traffic
};

// x = hi + lo, hi = fp16(x), lo = fp16(x - hi) (kernels_gemm_h1.hip h1_split8): one packed conversion per pair whose halves are the f16
// sources of two v_fma_mix{lo,hi}_f16; `neg1` = -1.0f in a scalar register the compiler cannot see through (a literal folds the fused
// multiply-add into cvt-back + subtract + cvt)
typedef _Float16 sf_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void sf_split8(const f32x4 a, const f32x4 b, float neg1, f16x8& hi, f16x8& lo) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const sf_f16x2 ha = __builtin_convertvector(f32x2{a[e], a[e + 1]}, sf_f16x2);
        const sf_f16x2 hb = __builtin_convertvector(f32x2{b[e], b[e + 1]}, sf_f16x2);
        hi[e] = ha[0];
        hi[e + 1] = ha[1];
        hi[4 + e] = hb[0];
        hi[4 + e + 1] = hb[1];
        lo[e] = (_Float16)__builtin_fmaf((float)ha[0], neg1, a[e]);
        lo[e + 1] = (_Float16)__builtin_fmaf((float)ha[1], neg1, a[e + 1]);
        lo[4 + e] = (_Float16)__builtin_fmaf((float)hb[0], neg1, b[e]);
        lo[4 + e + 1] = (_Float16)__builtin_fmaf((float)hb[1], neg1, b[e + 1]);
    }
}

template <int C1>
__global__ void __launch_bounds__(SF_NT, 2) stem_fused_kernel(StemFusedParams p) {
    using G = SfGeom<C1>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    _Float16* Wl = reinterpret_cast<_Float16*>(lds);
    float* Bs = reinterpret_cast<float*>(lds + (size_t)G::W_HALFS * 2);
    float* Es = Bs + ((G::BIAS_FLOATS + 3) & ~3);
    float* As = Es + G::E_FLOATS;                  // also the image patch (before stem2a overwrites it)
    const _Float16* W1h = Wl;
    const _Float16* W1l = Wl + G::W1_H;
    const _Float16* W2Ah = Wl + 2 * G::W1_H;
    const _Float16* W2Al = W2Ah + G::W2A_H;
    const _Float16* W2Bh = W2Ah + 2 * G::W2A_H;
    const _Float16* W2Bl = W2Bh + G::W2B_H;
    const float* b1 = Bs;
    const float* b2a = Bs + C1;
    const float* b2b = Bs + C1 + G::NA;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const float* scl = reinterpret_cast<const float*>(p.wimg + G::W_HALFS);       // [inv1, inv2a, inv2b, s1, s2a, s2b] (powers of two)
    const float inv1 = scl[0], inv2a = scl[1], inv2b = scl[2], sc1 = scl[3], sc2a = scl[4], sc2b = scl[5];
    float neg1 = -1.f;
    asm volatile("" : "+s"(neg1));

    // weights + biases -> LDS, once per (persistent) workgroup
    for (int i = tid; i < G::W_HALFS / 8; i += SF_NT)
        reinterpret_cast<u32x4*>(Wl)[i] = reinterpret_cast<const u32x4*>(p.wimg)[i];
    for (int i = tid; i < G::BIAS_FLOATS; i += SF_NT) Bs[i] = p.bias[i];

    const int tiles_per_img = p.tiles_x * p.tiles_y, ntiles = p.N * tiles_per_img;
    constexpr int PATCH_ELEMS = SF_PH * SF_PW * 3;
    constexpr int PRE = (PATCH_ELEMS + SF_NT - 1) / SF_NT;
    float pre[PRE];
    auto tile_origin = [&](int t, int& n, int& ty0, int& tx0) {
        n = t / tiles_per_img;
        const int r = t - n * tiles_per_img;
        const int tyi = r / p.tiles_x;
        ty0 = tyi * SF_TH;
        tx0 = (r - tyi * p.tiles_x) * SF_TW;
    };
    // which patch elements this thread moves never changes: (row, column, plane, LDS slot) once, per tile only the origin is added
    int p_rc[PRE], p_lds[PRE], p_ci[PRE];
#pragma unroll
    for (int k = 0; k < PRE; ++k) {
        const int i = tid + k * SF_NT;
        const int row = i / (SF_PW * 3), rem = i - row * (SF_PW * 3);
        const int col = rem / 3, ci = rem - col * 3;
        p_rc[k] = i < PATCH_ELEMS ? (row << 16) | col : -1;
        p_lds[k] = row * SF_PROW + rem;
        p_ci[k] = (p.in_ch == 1 ? 0 : ci) * p.H * p.W;
    }
    // global -> registers.  The loads are UNCONDITIONAL from clamped addresses and the zero padding is applied when the values go to
    // LDS (store_patch): a select on the loaded value right here makes the compiler wait for every load before it issues the next
    // (ISA of the first version: nine `global_load_dword ; s_waitcnt vmcnt(0)` pairs = nine serial memory round trips per tile with
    // all eight wavefronts of the CU behind them; ablation: 50 of 262 us on the recogniser's stem)
    auto load_patch = [&](int t) {
        int n, ty0, tx0;
        tile_origin(t, n, ty0, tx0);
        const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;
        const float* img = p.x + (size_t)n * p.in_ch * p.H * p.W;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int iy = min(max(iy0 + (p_rc[k] >> 16), 0), p.H - 1), ix = min(max(ix0 + (p_rc[k] & 0xffff), 0), p.W - 1);
            pre[k] = img[p_ci[k] + iy * p.W + ix];
        }
    };
    auto store_patch = [&](int ty0, int tx0) {    // (zero outside the image: the conv's padding)
        const int iy0 = 2 * ty0 - 1, ix0 = 2 * tx0 - 1;
#pragma unroll
        for (int k = 0; k < PRE; ++k) {
            const int iy = iy0 + (p_rc[k] >> 16), ix = ix0 + (p_rc[k] & 0xffff);
            const bool ok = (unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W;
            if (p_rc[k] >= 0) As[p_lds[k]] = ok ? pre[k] : 0.f;
        }
        // the slots behind the last real column of every row and behind the last row are read with zero weights: keep them finite
        if (tid < SF_PH) As[tid * SF_PROW + SF_PW * 3] = 0.f;
        if (tid >= 64 && tid < 80) As[SF_PH * SF_PROW + tid - 64] = 0.f;
    };

    unsigned emax = 0;      // largest |pre-activation| bit pattern of stem2b's outputs / the pool's: inf and NaN sort above every finite value
    int t = blockIdx.x;
    if (t < ntiles) load_patch(t);
    __syncthreads();
    for (; t < ntiles; t += gridDim.x) {
        int n, ty0, tx0;
        tile_origin(t, n, ty0, tx0);
        if (!(p.dbg & 16)) store_patch(ty0, tx0);
        __syncthreads();

        // the image's own e / a / cat width: p.W2, or - under a line table - the text line's (columns beyond it are computed but
        // stored as the zero padding stem2a, the pool, stem2b and stem3 expect there)
        const int W2n = p.line_tab ? min(p.W2, p.line_tab[n * kLineTabStride + 1]) : p.W2;
        // a tile whose whole e halo lies inside the map needs no per-element bounds tests (most tiles)
        const bool inner = ty0 + SF_EH <= p.H2 && tx0 + SF_EW <= W2n;

        // ---------------- stem1: e = ReLU(conv3x3 s2 (patch)) on the matrix cores, K = 3 kernel rows x 16 slots.  The bias rides in
        // the accumulator init; non-finite values are not tested here: they reach stem2b's outputs and the pool, which are
        for (int mb = wave; mb < G::MB1 && !(p.dbg & 1); mb += 8) {
            const int m = min(mb * 32 + l31, SF_EH * SF_EW - 1);
            const int ey = m / SF_EW, ex = m - ey * SF_EW;
            f32x16 acc1[G::NB1], acc2[G::NB1];
#pragma unroll
            for (int nb = 0; nb < G::NB1; ++nb) {
                const float bv = b1[min(nb * 32 + l31, C1 - 1)] * sc1;       // (the accumulators hold sums of x (w s1))
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc1[nb][r] = bv; acc2[nb][r] = 0.f; }
            }
            const float* src0 = As + 2 * ey * SF_PROW + 2 * ex * 3 + 8 * lhi;
            // Software pipeline (round 5): the fragments of kernel row kh + 1 are requested BEFORE the matrix instructions of row kh
            // are issued (the compiler's own order was load -> wait -> split -> MFMA per step: every step paid the LDS latency with
            // two wavefronts per SIMD to hide it).  sched_barrier keeps the requests where they are written.
            f32x2 q[4];
            f16x8 wh[G::NB1], wl[G::NB1];
            auto fetch1 = [&](int kh) {
                const float* src = src0 + kh * SF_PROW;
#pragma unroll
                for (int i = 0; i < 4; ++i) q[i] = *reinterpret_cast<const f32x2*>(src + 2 * i);
#pragma unroll
                for (int nb = 0; nb < G::NB1; ++nb) {
                    const _Float16* wb = W1h + min(nb * 32 + l31, C1 - 1) * G::R1 + 8 * lhi + kh * 16;
                    wh[nb] = *reinterpret_cast<const f16x8*>(wb);
                    wl[nb] = *reinterpret_cast<const f16x8*>(wb + G::W1_H);
                }
            };
            fetch1(0);
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const f32x4 v0 = f32x4{q[0][0], q[0][1], q[1][0], q[1][1]}, v1 = f32x4{q[2][0], q[2][1], q[3][0], q[3][1]};
                f16x8 bh[G::NB1], bl[G::NB1];
#pragma unroll
                for (int nb = 0; nb < G::NB1; ++nb) { bh[nb] = wh[nb]; bl[nb] = wl[nb]; }
                f16x8 ah, al;
                sf_split8(v0, v1, neg1, ah, al);
                __builtin_amdgcn_sched_barrier(0);
                if (kh + 1 < 3) fetch1(kh + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < G::NB1; ++nb) {
                    acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nb], acc1[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nb], acc2[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nb], acc2[nb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // One epilogue for every block (round 5): the bounds of the map are NOT tested per element here - an edge tile zeroes what
            // lies beyond the map in one pass behind the barrier (below).  Before, the last block of every tile, the second channel
            // block of C1 = 48 (16 of 32 lanes) and every block of an edge tile (a third of the recogniser's tiles: 24 rows = three
            // tile rows, the last one cut) went through ~10 VALU operations of index arithmetic per element.
#pragma unroll
            for (int nb = 0; nb < G::NB1; ++nb) {
                const int nn = nb * 32 + l31;
                float* dst = Es + (mb * 32 + 4 * lhi) * G::SE + nn;
                if (nn < C1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        dst[((r & 3) + 8 * (r >> 2)) * G::SE] = fmaxf((acc1[nb][r] + acc2[nb][r]) * inv1, 0.f);
                }
            }
        }
        __syncthreads();
        // rows / columns of the tile that lie inside the map (the line's own width under a line table); beyond them e and a are the
        // zero padding of stem2a / the pool / stem2b
        const int vh = p.H2 - ty0, vw = W2n - tx0;
        if (!inner) {
            constexpr int C4 = C1 / 4;
            for (int i = tid; i < SF_EH * SF_EW * C4; i += SF_NT) {
                const int pix = i / C4, cg = i - pix * C4;
                const int cy = pix / SF_EW, cx = pix - cy * SF_EW;
                if (cy >= vh || cx >= vw) *reinterpret_cast<f32x4*>(Es + pix * G::SE + 4 * cg) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
        }

        // ---------------- stem2a: a = ReLU(conv2x2 (e)), K = 4 taps x C1 channels.  Ten 32-pixel blocks on eight wavefronts: the
        // six wavefronts without a second block do the max-pool in the meantime (it only needs e)
        for (int mb = wave; mb < G::MB2 && !(p.dbg & 2); mb += 8) {
            const int m = min(mb * 32 + l31, SF_AH * SF_AW - 1);
            const int ay = m / SF_AW, ax = m - ay * SF_AW;
            const int nrow = min(l31, G::NA - 1);
            const float bv = b2a[nrow] * sc2a;
            f32x16 acc1, acc2;
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc1[r] = bv; acc2[r] = 0.f; }
            const float* src0 = Es + (ay * SF_EW + ax) * G::SE;
            const _Float16* wb = W2Ah + nrow * G::R2A + 8 * lhi;
            f32x4 nv0, nv1;
            f16x8 nbh, nbl;
            auto fetch2a = [&](int j) {
                // k0 = 16 j + 8 lhi -> (tap, channel): both halves resolved at compile time, one select per k-step
                constexpr int CAc = G::CA;
                const int t0 = (16 * j) / CAc, c0 = (16 * j) % CAc, t1 = (16 * j + 8) / CAc, c1 = (16 * j + 8) % CAc;
                const int o0 = ((t0 >> 1) * SF_EW + (t0 & 1)) * G::SE + c0, o1 = ((t1 >> 1) * SF_EW + (t1 & 1)) * G::SE + c1;
                const float* src = src0 + (lhi ? o1 : o0);
                nv0 = *reinterpret_cast<const f32x4*>(src);
                nv1 = *reinterpret_cast<const f32x4*>(src + 4);
                nbh = *reinterpret_cast<const f16x8*>(wb + 16 * j);
                nbl = *reinterpret_cast<const f16x8*>(wb + G::W2A_H + 16 * j);
            };
            fetch2a(0);
#pragma unroll
            for (int j = 0; j < G::K2A / 16; ++j) {        // (pipelined like stem1: step j + 1's fragments travel under step j's MFMAs)
                const f32x4 v0 = nv0, v1 = nv1;
                const f16x8 bh = nbh, bl = nbl;
                f16x8 ah, al;
                sf_split8(v0, v1, neg1, ah, al);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < G::K2A / 16) fetch2a(j + 1);
                __builtin_amdgcn_sched_barrier(0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2, 0, 0, 0);
                acc2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
            float* dst = As + (mb * 32 + 4 * lhi) * G::SA + l31;
            const bool lane_on = l31 < G::CB;               // channels NA .. CB - 1 are the zero padding of the a tile
            const bool real = l31 < G::NA;
            if (lane_on) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = (acc1[r] + acc2[r]) * inv2a;
                    dst[((r & 3) + 8 * (r >> 2)) * G::SA] = real ? fmaxf(v, 0.f) : 0.f;
                }
            }
        }
        // ---------------- pool: p = max over e[y..y+1][x..x+1] (e is zero beyond the map: F.pad(0,1,0,1)) -> cat[..., :C1];
        // by the wavefronts that had one stem2a block only
        {
            constexpr int C4 = C1 / 4;
            constexpr int FIRST = G::MB2 > 8 ? G::MB2 - 8 : 0, NTH = (8 - FIRST) * 64;
            if (wave >= FIRST && !(p.dbg & 4)) {
                for (int i = tid - FIRST * 64; i < SF_TH * SF_TW * C4; i += NTH) {
                    const int pix = i / C4, cg = i - pix * C4;
                    const int py = pix >> 5, px = pix & 31;
                    const int gy = ty0 + py, gx = tx0 + px;
                    if (gy >= p.H2 || gx >= p.W2) continue;
                    const float* e0 = Es + (py * SF_EW + px) * G::SE + 4 * cg;
                    const f32x4 a = *reinterpret_cast<const f32x4*>(e0), b = *reinterpret_cast<const f32x4*>(e0 + G::SE);
                    const f32x4 c = *reinterpret_cast<const f32x4*>(e0 + SF_EW * G::SE), d = *reinterpret_cast<const f32x4*>(e0 + (SF_EW + 1) * G::SE);
                    f32x4 o;
#pragma unroll
                    for (int k = 0; k < 4; ++k) o[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(c[k], d[k]));
                    emax = max(emax, __float_as_uint(o[0] + o[1] + o[2] + o[3]) & 0x7fffffffu);      // (e >= 0: the sum is non-finite iff a term is)
                    __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(p.y + (((size_t)n * p.H2 + gy) * p.W2 + gx) * p.yld + 4 * cg));
                }
            }
        }
        __syncthreads();
        if (!inner) {            // a beyond the map: the zero padding of stem2b
            constexpr int B4 = G::CB / 4;
            for (int i = tid; i < SF_AH * SF_AW * B4; i += SF_NT) {
                const int pix = i / B4, cg = i - pix * B4;
                const int ay = pix / SF_AW, ax = pix - ay * SF_AW;
                if (ay >= vh || ax >= vw) *reinterpret_cast<f32x4*>(As + pix * G::SA + 4 * cg) = f32x4{0.f, 0.f, 0.f, 0.f};
            }
            __syncthreads();
        }

        // the next tile's patch travels while stem2b runs
        const int tn = t + (int)gridDim.x;
        if (tn < ntiles && !(p.dbg & 16)) load_patch(tn);

        // ---------------- stem2b: b = ReLU(conv2x2 (a)) -> cat[..., C1:], one tile row (32 pixels) per wavefront
        if (!(p.dbg & 8)) {
            const int by = wave, bx = l31;
            f32x16 acc1[G::NB3], acc2[G::NB3];
#pragma unroll
            for (int nb = 0; nb < G::NB3; ++nb) {
                const float bv = b2b[min(nb * 32 + l31, C1 - 1)] * sc2b;
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc1[nb][r] = bv; acc2[nb][r] = 0.f; }
            }
            const float* src0 = As + (by * SF_AW + bx) * G::SA;
            f32x4 nv0, nv1;
            f16x8 nwh[G::NB3], nwl[G::NB3];
            auto fetch2b = [&](int j) {
                constexpr int CBc = G::CB;
                const int t0 = (16 * j) / CBc, c0 = (16 * j) % CBc, t1 = (16 * j + 8) / CBc, c1 = (16 * j + 8) % CBc;
                const int o0 = ((t0 >> 1) * SF_AW + (t0 & 1)) * G::SA + c0, o1 = ((t1 >> 1) * SF_AW + (t1 & 1)) * G::SA + c1;
                const float* src = src0 + (lhi ? o1 : o0);
                nv0 = *reinterpret_cast<const f32x4*>(src);
                nv1 = *reinterpret_cast<const f32x4*>(src + 4);
#pragma unroll
                for (int nb = 0; nb < G::NB3; ++nb) {
                    const int nrow = min(nb * 32 + l31, C1 - 1);
                    nwh[nb] = *reinterpret_cast<const f16x8*>(W2Bh + nrow * G::R2B + 16 * j + 8 * lhi);
                    nwl[nb] = *reinterpret_cast<const f16x8*>(W2Bl + nrow * G::R2B + 16 * j + 8 * lhi);
                }
            };
            fetch2b(0);
#pragma unroll
            for (int j = 0; j < G::K2B / 16; ++j) {        // (pipelined like stem1)
                const f32x4 v0 = nv0, v1 = nv1;
                f16x8 bh[G::NB3], bl[G::NB3];
#pragma unroll
                for (int nb = 0; nb < G::NB3; ++nb) { bh[nb] = nwh[nb]; bl[nb] = nwl[nb]; }
                f16x8 ah, al;
                sf_split8(v0, v1, neg1, ah, al);
                __builtin_amdgcn_sched_barrier(0);
                if (j + 1 < G::K2B / 16) fetch2b(j + 1);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nb = 0; nb < G::NB3; ++nb) {
                    acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[nb], acc1[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[nb], acc2[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[nb], acc2[nb], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // The next tile's patch (requested above, before these MFMAs) must have landed BEFORE the stores below are issued: loads
            // and stores retire in issue order, and the compiler's wait for `pre` at the top of the next tile would otherwise be a
            // vmcnt(0) behind 32 stores per wavefront - an HBM write round trip per tile with all eight wavefronts of the CU waiting
            // (ablation: 50 of 262 us on the recogniser's stem).  Here the loads have had the whole MFMA phase.
            __builtin_amdgcn_s_waitcnt(0x0F70);       // vmcnt(0)
            const int gy = ty0 + by;
            float* yrow = p.y + (((size_t)n * p.H2 + min(gy, p.H2 - 1)) * p.W2 + tx0 + 4 * lhi) * p.yld + C1;
#pragma unroll
            for (int nb = 0; nb < G::NB3; ++nb) {
                const int nn = nb * 32 + l31;
                if (inner) {
                    if (nn < C1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const float v = (acc1[nb][r] + acc2[nb][r]) * inv2b;
                            emax = max(emax, __float_as_uint(v) & 0x7fffffffu);
                            __builtin_nontemporal_store(fmaxf(v, 0.f), &yrow[(size_t)((r & 3) + 8 * (r >> 2)) * p.yld + nn]);
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int off = (r & 3) + 8 * (r >> 2);
                        const float v = (acc1[nb][r] + acc2[nb][r]) * inv2b;
                        emax = max(emax, __float_as_uint(v) & 0x7fffffffu);
                        if (gy < p.H2 && tx0 + 4 * lhi + off < p.W2 && nn < C1)
                            yrow[(size_t)off * p.yld + nn] = tx0 + 4 * lhi + off < W2n ? fmaxf(v, 0.f) : 0.f;   // (zero beyond a line's own width)
                    }
                }
            }
        }
        __syncthreads();        // e / a are free for the next tile
    }
    if (emax >= 0x7f800000u && p.range_flag) rd_raise_flag(p.range_flag);
}

bool stem_fused_supported(int c1) { return c1 == 24 || c1 == 32 || c1 == 48; }

// power-of-two scale of one weight matrix: max |w| s in [2^13, 2^14) (prepare_gemm_h1_weights) -> exponent
static int sf_scale_exp(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
    if (!(mx > 0.f) || !std::isfinite(mx)) return 0;
    int x = 0;
    (void)std::frexp(mx, &x);
    const int ex = 14 - x;
    return ex > 100 ? 100 : ex < -100 ? -100 : ex;
}

template <int C1>
static void sf_prepare(const float* w1, const float* w2a, const float* w2b, std::vector<uint16_t>& img) {
    using G = SfGeom<C1>;
    img.assign(G::IMG_HALFS, 0);
    const int ex[3] = {sf_scale_exp(w1, (size_t)27 * C1), sf_scale_exp(w2a, (size_t)G::NA * G::K2A), sf_scale_exp(w2b, (size_t)C1 * 4 * G::NA)};
    auto put = [&](size_t hi_at, size_t lo_at, float v, int e) {
        const float vs = std::ldexp(v, e);
        const _Float16 h = (_Float16)vs;
        const _Float16 l = (_Float16)(vs - (float)h);
        __builtin_memcpy(&img[hi_at], &h, 2);
        __builtin_memcpy(&img[lo_at], &l, 2);
    };
    // stem1: w1 [27][C1] with row (kh * 3 + kw) * 3 + ci  ->  [co][kh * 16 + kw * 3 + ci]
    for (int co = 0; co < C1; ++co)
        for (int kh = 0; kh < 3; ++kh)
            for (int j = 0; j < 9; ++j)
                put((size_t)co * G::R1 + kh * 16 + j, (size_t)G::W1_H + (size_t)co * G::R1 + kh * 16 + j, w1[(size_t)(kh * 9 + j) * C1 + co], ex[0]);
    // stem2a: folded [NA][4 * C1] with k = tap * C1 + c (tap = dy * 2 + dx): same order
    const size_t o2a = 2 * (size_t)G::W1_H;
    for (int n = 0; n < G::NA; ++n)
        for (int k = 0; k < G::K2A; ++k)
            put(o2a + (size_t)n * G::R2A + k, o2a + G::W2A_H + (size_t)n * G::R2A + k, w2a[(size_t)n * G::K2A + k], ex[1]);
    // stem2b: folded [C1][4 * NA] with k = tap * NA + c  ->  tap * CB + c (channels padded to a multiple of 8 per tap)
    const size_t o2b = o2a + 2 * (size_t)G::W2A_H;
    for (int n = 0; n < C1; ++n)
        for (int tap = 0; tap < 4; ++tap)
            for (int c = 0; c < G::NA; ++c)
                put(o2b + (size_t)n * G::R2B + tap * G::CB + c, o2b + G::W2B_H + (size_t)n * G::R2B + tap * G::CB + c,
                    w2b[(size_t)n * 4 * G::NA + tap * G::NA + c], ex[2]);
    float scl[6];
    for (int k = 0; k < 3; ++k) { scl[k] = std::ldexp(1.f, -ex[k]); scl[3 + k] = std::ldexp(1.f, ex[k]); }
    __builtin_memcpy(&img[G::W_HALFS], scl, sizeof(scl));
}
// w1: the stem layout [27][C1] (kh, kw, ci major; co fastest), w2a / w2b: folded [Cout][kh * kw * Cin] (ci fastest), BN folded
void prepare_stem_fused_weights(int c1, const float* w1, const float* w2a, const float* w2b, std::vector<uint16_t>& img) {
    if (c1 == 24) sf_prepare<24>(w1, w2a, w2b, img);
    else if (c1 == 32) sf_prepare<32>(w1, w2a, w2b, img);
    else sf_prepare<48>(w1, w2a, w2b, img);
}

template <int C1>
static void sf_launch(StemFusedParams& p, hipStream_t s, int n_cu) {
    static unsigned long long lds_ok = 0;
    rd_allow_dynamic_lds((const void*)stem_fused_kernel<C1>, SfGeom<C1>::LDS_BYTES, lds_ok);
    const int ntiles = p.N * p.tiles_x * p.tiles_y;
    hipLaunchKernelGGL(stem_fused_kernel<C1>, dim3(ntiles < n_cu ? ntiles : n_cu), dim3(SF_NT), SfGeom<C1>::LDS_BYTES, s, p);
}

// x NCHW [N][in_ch][H][W] -> cat NHWC [N][H2][W2][2 c1] (row stride yld floats)
void launch_stem_fused(int c1, const float* x, int N, int H, int W, int in_ch, const uint16_t* wimg, const float* bias, float* y, int yld,
                       unsigned* range_flag, hipStream_t s, const int32_t* line_tab) {
    static const int n_cu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    StemFusedParams p{};
    p.x = x; p.N = N; p.H = H; p.W = W; p.in_ch = in_ch;
    p.wimg = wimg; p.bias = bias; p.y = y; p.yld = yld;
    p.H2 = (H + 2 - 3) / 2 + 1;
    p.W2 = (W + 2 - 3) / 2 + 1;
    p.tiles_y = (p.H2 + SF_TH - 1) / SF_TH;
    p.tiles_x = (p.W2 + SF_TW - 1) / SF_TW;
    p.range_flag = range_flag;
    p.line_tab = line_tab;
    static const int dbg = [] { const char* e = getenv("RD_STEM_DBG"); return e ? atoi(e) : 0; }();
    p.dbg = dbg;
    if (c1 == 24) sf_launch<24>(p, s, n_cu);
    else if (c1 == 32) sf_launch<32>(p, s, n_cu);
    else sf_launch<48>(p, s, n_cu);
}

}  // namespace rd
