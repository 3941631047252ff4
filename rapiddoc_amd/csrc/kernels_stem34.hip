// Stem tail in one kernel: 3x3 / stride 2 / pad 1 convolution (+ BN + act) followed by the 1x1 convolution (+ BN + act) that consumes it -
// `stem3` -> `stem4` of PPLCNetV4's LargeStem (rec_lcnetv4.py:154-169: 2 c1 -> c1 at stride 2, then c1 -> c2) and of PPHGNetV2's StemBlock
// (rec_pphgnetv2.py:1040-1056), split-fp16 arithmetic with ONE fp32 accumulator set (kernels_conv3x3_h1.hip / kernels_gemm_h1.hip).
//
// Why (round 6): through round 5 stem3 ran as an im2col GEMM (conv_igemm_h3_kernel<256 x 64>: 110-122 TFLOP/s, two accumulator sets) and
// stem4 as a streaming 1x1 (2.7-3.2 TB/s): 3.5 + 1.6 ms per 32-page step for the recogniser alone, i.e. ~0.5 TB moved where the pair's
// algorithmic traffic (input once, c2-wide output once) is 0.19 TB - the c1-wide tensor between them is written and read back, and the
// im2col gather fetches every input pixel 2.25 times through L2.  Here
//   * a workgroup (four wavefronts, 32 output pixels of one row each) owns a 4 x 32 (or 2 x 64) OUTPUT tile; its 9 x 65 input patch is
//     staged in LDS 16 input channels at a time, split ONCE into two fp16 planes.  The patch's even and odd columns live in separate runs:
//     a tap's 32 pixels are then consecutive 16-byte entries (per plane and k-half one entry per pixel, unpadded: a 16-lane read phase
//     covers 256 contiguous bytes - conflict-free) although the convolution strides by two.  Patch requests cover 32 channels (whole
//     128-byte lines) and stay in registers until their half is written;
//   * a pass's stem3 weights (nine taps x hi / lo fragments per 32-wide output block, 36 KB) are copied by linear `global_load_lds` while the
//     next patch half is being split and written; nothing is waited for inside a pass (two barriers per pass, none per tap); 75 KB of LDS,
//     two workgroups per CU;
//   * stem3 is computed TRANSPOSED (weights = the MFMA's A operand, pixels = B): its accumulators D^T[channel][pixel] hold, per lane, the
//     eight k-slots of its pixel for each 16-channel step once stem4's input channels are permuted on the host to the C/D register order -
//     bias + act + split in registers and they ARE the A fragments of the 1x1 (the fused mixers' trick): the c1-wide tensor never leaves
//     the register file.  stem4's fragments (<= 18 KB) are read straight from L2; its output D[pixel][channel] is stored as whole 128-byte
//     channel runs through a range-checked buffer descriptor.
// One kernel for every launch size; results do not depend on M.
// STATUS: correct (tests/test_gpu_stem34.py) and NOT the default (stem34_enabled below): three structures were measured (per-tap weight
// ring with four-wavefront workgroups; one eight-wavefront workgroup per CU with double-buffered pass weights; this one) and none beats
// the two-kernel path by more than 7 % - ablations in profiles/r6_stem34.txt.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "rd_device.h"

namespace rd {

static constexpr int S34_WAVES = 4;                                 // wavefronts per workgroup = 32-pixel output row segments per tile
static constexpr int S34_SLAB_STEPS = 9;                            // k-steps (taps) per 16-channel pass

typedef _Float16 s34_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s34_split4(const f32x4 v, float neg1, f16x4& hi, f16x4& lo) {
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const s34_f16x2 h = __builtin_convertvector(f32x2{v[e], v[e + 1]}, s34_f16x2);
        hi[e] = h[0];
        hi[e + 1] = h[1];
        lo[e] = (_Float16)__builtin_fmaf((float)h[0], neg1, v[e]);
        lo[e + 1] = (_Float16)__builtin_fmaf((float)h[1], neg1, v[e + 1]);
    }
}

// Output tile = TR rows x (4 / TR) segments of 32 columns (TR = 4: 4 x 32, TR = 2: 2 x 64; the launcher takes the shape that wastes fewer
// outputs on the map at hand).  Wavefront w owns row w % TR, segment w / TR.
template <int TR>
struct S34Geom {
    static constexpr int SEG = S34_WAVES / TR;                      // 32-column segments per tile row
    static constexpr int TC = 32 * SEG;                             // output columns per tile
    static constexpr int PH = 2 * TR + 1, PW = 2 * TC + 1;          // input patch
    static constexpr int NE = TC + 1;                               // even-column entries per patch row, then TC odd ones
    static constexpr int PP = PH * PW;                              // patch pixels (1105 / 1161)
    static constexpr int SUB = PP * 16;                             // bytes of one (plane, k-half) run: one 16-byte entry per patch pixel
    static constexpr int PATCH = 4 * SUB;                           // hi / lo planes x k-halves
    static constexpr int NSLOT = (PP * 8 + 255) / 256;              // float4 slots per thread and 32-channel request round
};

template <int NB>
struct S34Frag { f16x8 xh, xl, wh[NB], wl[NB]; };

// NB = 32-wide blocks of stem3's output channels (1 or 2), MB = 32-wide blocks of stem4's (1 .. 3); HALF: N1 <= 32 NB - 16
//
// Second form (the first - four wavefronts, 4 x 32 tiles, two workgroups per CU, stem3's weights through a four-slab LDS-DMA ring with a
// barrier per tap as in conv3x3_h1_kernel - measured 235 us on the recogniser's stem where the two kernels it replaces took 248: a tap
// here is 6 MFMAs per wavefront, a third of the stride-1 kernel's, so three slabs of lead are ~0.4 us against a DMA round trip of 1 - 2 us
// and every step waited for its slab; profiles/r6_stem34.txt).  Now ONE workgroup of eight wavefronts per CU and nothing is waited for
// inside a pass:
//   * LDS holds the patch of the CURRENT 16 input channels (70 - 74 KB: per (plane, k-half) one 16-byte entry per patch pixel, unpadded -
//     a 16-lane read phase covers 256 contiguous bytes, conflict-free without padding) and TWO buffers of a whole pass's weights (nine taps x
//     NB x {hi, lo} fragments = 36 KB at NB = 2): the next pass's arrive by LDS-DMA while this pass computes;
//   * the NEXT request round of the patch (32 channels = whole 128-byte lines, possibly the next tile's) is in flight in registers during the
//     pass as well; one full wait + two barriers per pass (everybody done reading / the new patch half written), none per tap.
template <int NB, int MB, int TR, bool HALF>
__global__ void __launch_bounds__(256, 2) conv3x3s2_pw_h1_kernel(Stem34Params p, int tiles_r, int tiles_c, int ntiles) {
    using G = S34Geom<TR>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int SLAB = NB * 2 * 1024;                     // one tap: hi and lo fragment per output block
    constexpr int WPASS = S34_SLAB_STEPS * SLAB;            // a pass's weights
    constexpr int WPIECES = WPASS / 1024;                   // 1-KB DMA pieces per pass (18 / 36)
    constexpr int K2 = 2 * NB - (HALF ? 1 : 0);             // 16-channel steps of the 1x1: ceil(N1 / 16) (HALF: the last block's upper half is padding)
    constexpr int NSLOT = G::NSLOT;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    unsigned char* Pt = smem;                               // [plane][k-half][patch entry] x 16 bytes
    unsigned char* Wb = smem + G::PATCH;                    // one pass buffer
    float* Bs = reinterpret_cast<float*>(Wb + WPASS);       // b3 (NB * 32, zero padded) | b4 (MB * 32, zero padded)
    float neg1 = -1.f;
    asm volatile("" : "+s"(neg1));
    const int passes = p.Cin / 16;

    // ---- stem3's weights of pass `ps` -> buffer `buf` (linear copy: the image is in consumption order)
    auto dma_pass = [&](int ps) {
        if (p.abl & 16) return;
        const unsigned char* src = reinterpret_cast<const unsigned char*>(p.w3) + (size_t)ps * WPASS + lane * 16;
        unsigned char* dst = Wb;
#pragma unroll
        for (int u = 0; u < (WPIECES + S34_WAVES - 1) / S34_WAVES; ++u) {
            const int f = wave + u * S34_WAVES;
            if (f < WPIECES)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024),
                                                 (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
        }
    };

    // ---- patch staging: a request round = 32 input channels of the patch = PP pixels x 8 channel groups (of 4) float4 slots; thread t
    // owns slots t, t + 512, ...: slot i is pixel (t + 512 i) / 8, group ((t + 512 i) % 8) ^ 4 (i % 2) - a thread's slots alternate between
    // the round's two 16-channel halves (each half is split and written when its pass comes)
    unsigned xmax = 0;
    u32x4 pre[NSLOT];
    unsigned pre_ok = 0;
    auto load_patch = [&](int im, int oh0, int ow0, int c0, int halves) {
        if (p.abl & 4) return;
        typedef __amdgpu_buffer_rsrc_t rsrc_t;
        const rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x + (size_t)im * p.H * p.W * p.xld + c0), 0, 0x7fffffff, 0x00020000);
        int stid = tid;
        asm volatile("" : "+v"(stid));                      // (keeps the slot decode out of the registers that live through the steps)
        pre_ok = 0;
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int q = stid + 256 * i;
            const int px = min(q >> 3, G::PP - 1), grp = (q & 7) ^ ((i & 1) << 2);
            const int row = px / G::PW, col = px - row * G::PW;
            const int ih = 2 * oh0 - 1 + row, iw = 2 * ow0 - 1 + col;
            if ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) pre_ok |= 1u << i;
            const unsigned gsel = grp < 4 * halves ? 16u * (unsigned)grp : 0u;       // (a 16-channel tail: the second half is never read)
            const unsigned off = ((unsigned)min(max(ih, 0), p.H - 1) * (unsigned)p.W + (unsigned)min(max(iw, 0), p.W - 1)) * (unsigned)p.xld * 4u + gsel;
            pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rx, (int)off, 0, 0);
        }
    };
    auto write_patch = [&](int half) {
        if (p.abl & 2) return;
        int stid = tid;
        asm volatile("" : "+v"(stid));
        const int tsel = ((stid >> 2) & 1) ^ half;           // this thread's slots of `half` are those with i % 2 == tsel
#pragma unroll
        for (int i = 0; i < NSLOT; ++i) {
            const int q = stid + 256 * i;
            if ((i & 1) != tsel || q >= G::PP * 8) continue;
            const int px = q >> 3, grp = q & 3;              // channels 4 grp .. + 4 of the half: k-half grp / 2, bytes 8 (grp % 2) of its entry
            const int row = px / G::PW, col = px - row * G::PW;
            const f32x4 x4 = (pre_ok >> i) & 1u ? __builtin_bit_cast(f32x4, pre[i]) : f32x4{0.f, 0.f, 0.f, 0.f};
            f16x4 hi, lo;
            s34_split4(x4, neg1, hi, lo);
#pragma unroll
            for (int e = 0; e < 4; ++e) xmax = max(xmax, __float_as_uint(x4[e]) & 0x7fffffffu);     // (NaN / inf patterns compare above every finite one)
            // even columns first (NE entries), then the odd ones
            const unsigned pos = (unsigned)(row * G::PW + ((col & 1) ? G::NE + (col >> 1) : (col >> 1)));
            const unsigned o = (unsigned)(grp >> 1) * G::SUB + pos * 16u + (unsigned)(grp & 1) * 8u;
            *reinterpret_cast<f16x4*>(Pt + o) = hi;
            *reinterpret_cast<f16x4*>(Pt + 2 * G::SUB + o) = lo;
        }
    };

    // ---- fragments of one step: B = this wavefront's 32 output pixels at tap (kh, kw): patch row 2 r + kh, column 2 ox + kw (kw = 0: even
    // entry ox, 1: odd entry ox, 2: even entry ox + 1); A = the pass buffer's fragments of that tap
    const int wr = wave % TR, wseg = wave / TR;
    const unsigned x_lane = (unsigned)lhi * G::SUB + (unsigned)((2 * wr) * G::PW + 32 * wseg + l31) * 16u;
    const unsigned w_lane = (unsigned)lane * 16u;
    auto read_frag = [&](S34Frag<NB>& f, int tap) {
        const int kh = tap / 3, kw = tap - 3 * kh;
        const unsigned a = x_lane + (unsigned)(kh * G::PW + (kw == 1 ? G::NE : kw == 2 ? 1 : 0)) * 16u;
        f.xh = *reinterpret_cast<const f16x8*>(Pt + a);
        f.xl = *reinterpret_cast<const f16x8*>(Pt + 2 * G::SUB + a);
        const unsigned char* wb = Wb + tap * SLAB + w_lane;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f.wh[nb] = *reinterpret_cast<const f16x8*>(wb + (2 * nb) * 1024);
            f.wl[nb] = *reinterpret_cast<const f16x8*>(wb + (2 * nb + 1) * 1024);
        }
    };

    f32x16 acc[NB];
    S34Frag<NB> fr[2];
    auto run_pass = [&]() {
        if (p.abl & 1) return;
        read_frag(fr[0], 0);
#pragma unroll
        for (int s = 0; s < S34_SLAB_STEPS; ++s) {
            S34Frag<NB>& cur = fr[s & 1];
            if (s + 1 < S34_SLAB_STEPS) read_frag(fr[(s + 1) & 1], s + 1);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh[nb], cur.xh, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wh[nb], cur.xl, acc[nb], 0, 0, 0);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.wl[nb], cur.xh, acc[nb], 0, 0, 0);
        }
    };

    auto decode = [&](int v, int& im, int& oh0, int& ow0) {
        // XCD-contiguous tile order (conv3x3_h1_kernel): an XCD's workgroups walk one contiguous run of the tile list
        const int xcd = v & 7, jj = v >> 3, q = ntiles >> 3, rm = ntiles & 7;
        int t = (xcd < rm ? xcd * (q + 1) : rm * (q + 1) + (xcd - rm) * q) + jj;
        const int tc = t % tiles_c;
        t /= tiles_c;
        const int tr = t % tiles_r;
        im = t / tiles_r;
        oh0 = tr * TR;
        ow0 = tc * G::TC;
    };

    unsigned emax = 0;
    const uint16_t* w4 = p.w4;
    int v = blockIdx.x;
    if (v >= ntiles) return;
    // the biases live in LDS: as global loads inside the epilogue's `if`s each one was fenced by its own vmcnt(0) - twelve serial round trips
    // per tile, each also waiting for the stores in front of it (75 of 238 us, profiles/r6_stem34.txt)
    for (int i = tid; i < NB * 32; i += 256) Bs[i] = p.b3[i];
    for (int i = tid; i < MB * 32; i += 256) Bs[NB * 32 + i] = i < p.N2 ? p.b4[i] : 0.f;
    int img, oh0, ow0;
    decode(v, img, oh0, ow0);
    if (p.abl >> 10) {      // developer experiment: start every other workgroup of an XCD late
        const bool late = (p.abl & 512) ? ((int)blockIdx.x >= (int)gridDim.x / 2) : (((int)blockIdx.x >> 3) & 1);
        if (late) for (int i = 0; i < (p.abl >> 10); ++i) __builtin_amdgcn_s_sleep(127);
    }
    // prologue: pass 0's weights and the first request round; the first half written
    dma_pass(0);
    load_patch(img, oh0, ow0, 0, min(2, passes));
    write_patch(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

#pragma unroll 1
    while (true) {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[nb][i] = 0.f;
        const int vn = v + (int)gridDim.x;
        int img_n = 0, oh_n = 0, ow_n = 0;
        if (vn < ntiles) decode(vn, img_n, oh_n, ow_n);
#pragma unroll 1
        for (int pass = 0; pass < passes; ++pass) {
            // on entry: the patch half of `pass` and its weights are in LDS.  Requested now, landing under this pass's MFMAs: when this pass
            // is the second half of its round (or the 16-channel tail), the next request round of the patch - possibly the next tile's
            const bool last = pass + 1 == passes;
            const bool more = !last || vn < ntiles;
            const bool round_ends = (pass & 1) || last;
            if (round_ends && more) {
                if (!last) load_patch(img, oh0, ow0, (pass + 1) * 16, min(2, passes - pass - 1));
                else load_patch(img_n, oh_n, ow_n, 0, min(2, passes));
            }
            run_pass();
            if (last && !(p.abl & 8)) {
                // ---- stem3's epilogue in registers: h = act3(acc * inv3 + b3); lane (pixel l31, half lhi) holds channel 32 nb + 8 (r / 4) +
                // 4 lhi + r % 4 in register r: registers 8 (j % 2) .. + 8 of block j / 2 are its eight k-slots of the 1x1's step j (w4's input
                // channels are permuted to this order in the image)
                f16x8 hb[K2], lb[K2];
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int lhie = lane_e >> 5, l31e = lane_e & 31;
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
#pragma unroll
                    for (int r4 = 0; r4 < 4; ++r4) {
                        if (2 * nb + (r4 >> 1) >= K2) continue;              // (HALF: channels that are padding)
                        const f32x4 bv = *reinterpret_cast<const f32x4*>(Bs + 32 * nb + 8 * r4 + 4 * lhie);
                        f32x4 h;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            h[e] = fmaf(acc[nb][4 * r4 + e], p.w3_inv, bv[e]);
                            emax = max(emax, __float_as_uint(h[e]) & 0x7fffffffu);
                        }
                        if (p.act3 == ACT_RELU) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = fmaxf(h[e], 0.f);
                        } else if (p.act3 != ACT_NONE) {
#pragma unroll
                            for (int e = 0; e < 4; ++e) h[e] = rd_act(h[e], p.act3);
                        }
                        f16x4 hi, lo;
                        s34_split4(h, neg1, hi, lo);
                        const int j = 2 * nb + (r4 >> 1), e0 = 4 * (r4 & 1);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            hb[j][e0 + e] = hi[e];
                            lb[j][e0 + e] = lo[e];
                        }
                    }
                }
                // ---- the 1x1 (transposed): D2^T[32 mb + ..][pixel] = sum_j W4[mb][j] . H^T[j]; fragments straight from L2 (the same <= 18 KB
                // for every tile of every workgroup)
                const int oh = oh0 + wr;
                typedef __amdgpu_buffer_rsrc_t rsrc_t;
                const unsigned img_bytes = (unsigned)p.OH * (unsigned)p.OW * (unsigned)p.yld * 4u;      // (launcher: < 2^31)
                const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + (size_t)img * p.OH * p.OW * p.yld, 0, (int)img_bytes, 0x00020000);
                // The 1x1 is computed NON-transposed (H's fragments serve as the A operand as they are - A and B fragments have the same lane
                // layout): D2[pixel][channel], lane = output channel 32 mb + l31, registers = 16 of the wavefront's 32 pixels - a store
                // instruction writes two whole 128-byte channel runs (the transposed form wrote 32-byte pieces at the pixel stride: 47 us for
                // the recogniser stem's 100 MB).  Unconditional: out-of-map pixels / channels beyond N2 get an offset past the descriptor's end.
                const unsigned rowbase = (unsigned)(min(oh, p.OH - 1) * p.OW) * (unsigned)p.yld * 4u;
                const int pw0 = ow0 + 32 * wseg + 4 * lhie;        // first pixel column of this lane's register 0
                unsigned omax = 0;
#pragma unroll
                for (int mb = 0; mb < MB; ++mb) {
                    f32x16 a2;
#pragma unroll
                    for (int i = 0; i < 16; ++i) a2[i] = 0.f;
                    f16x8 wh[K2], wl[K2];
#pragma unroll
                    for (int j = 0; j < K2; ++j) {
                        wh[j] = *reinterpret_cast<const f16x8*>(w4 + ((size_t)(mb * K2 + j) * 2) * 512 + lane_e * 8);
                        wl[j] = *reinterpret_cast<const f16x8*>(w4 + ((size_t)(mb * K2 + j) * 2 + 1) * 512 + lane_e * 8);
                    }
#pragma unroll
                    for (int j = 0; j < K2; ++j) {
                        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb[j], wh[j], a2, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(lb[j], wh[j], a2, 0, 0, 0);
                        a2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(hb[j], wl[j], a2, 0, 0, 0);
                    }
                    const int n = 32 * mb + l31e;
                    const float bv = Bs[NB * 32 + n];
                    const bool nok = n < p.N2 && oh < p.OH && !(p.abl & 32);
                    float o[16];
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        o[i] = fmaf(a2[i], p.w4_inv, bv);
                        omax = max(omax, __float_as_uint(o[i]) & 0x7fffffffu);
                    }
                    if (p.act4 == ACT_RELU) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[i] = fmaxf(o[i], 0.f);
                    } else if (p.act4 != ACT_NONE) {
#pragma unroll
                        for (int i = 0; i < 16; ++i) o[i] = rd_act(o[i], p.act4);
                    }
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int pw = pw0 + (i & 3) + 8 * (i >> 2);           // C/D register i: pixel row 8 (i / 4) + 4 lhi + i % 4 of the 32
                        const unsigned off = (nok && pw < p.OW) ? rowbase + ((unsigned)pw * (unsigned)p.yld + (unsigned)n) * 4u : 0xfffffff0u;
                        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[i]), ry, (int)off, 0, 2);
                    }
                }
                if (omax >= 0x7f800000u) emax = omax;       // a non-finite output (fp32 values beyond the fp16 range are fine here)
            }
            if (!more) break;
            // every wavefront is done reading this pass's patch half and weights: the next pass's weights are requested (cyclic: the next
            // tile starts at pass 0 again) and fly while the next patch half is split and written
            asm volatile("s_barrier" ::: "memory");
            dma_pass(last ? 0 : pass + 1);
            write_patch(last ? 0 : ((pass + 1) & 1));
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        if (vn >= ntiles) break;
        v = vn; img = img_n; oh0 = oh_n; ow0 = ow_n;
    }
    // stem3's activations feed the fp16 split: anything at or beyond the fp16 range (or non-finite, anywhere) -> fp32 re-run
    if ((emax >= 0x477fe000u || xmax >= 0x477fe000u) && p.range_flag) rd_raise_flag(p.range_flag);
}

// Whether the engine routes stem3 -> stem4 through this kernel.  OFF by default: at the bench's shapes it measures 221 - 256 us against 248
// for the pair it replaces on the recogniser's stem (505 vs 529 detector, 574 vs 577 B4) and the 32-page step does not move (79.16 vs
// 79.15 ms, profiles/r6_stem34.txt) - its phases (patch requests at the HBM limit, split + LDS write, MFMAs, epilogue) add up instead of
// overlapping; RD_STEM34=1 switches it on (parity and launch-invariance tests run it through rd_debug_stem34 either way).
bool stem34_enabled() {
    static const bool on = [] { const char* e = getenv("RD_STEM34"); return e && e[0] == '1'; }();
    return on;
}
bool stem34_shape_ok(int cin, int n1, int n2) {
    return cin % 16 == 0 && cin >= 16 && cin <= 256 && n1 >= 8 && n1 <= 64 && n1 % 4 == 0 && n2 >= 8 && n2 <= 96 && n2 % 4 == 0 && !(n1 > 48 && n2 > 64);
}

static int wgs_per_cu() {
    static const int n = [] { const char* e = getenv("RD_STEM34_WGS"); return e ? atoi(e) : 2; }();     // developer A/B
    return n;
}
template <int NB, int MB, int TR, bool HALF>
static void launch_stem34_t(const Stem34Params& p, hipStream_t s, int n_cu) {
    using G = S34Geom<TR>;
    const int tiles_r = (p.OH + TR - 1) / TR, tiles_c = (p.OW + G::TC - 1) / G::TC;
    const int ntiles = p.N * tiles_r * tiles_c;
    const size_t lds = (size_t)G::PATCH + (size_t)S34_SLAB_STEPS * NB * 2048 + (size_t)(NB + MB) * 32 * sizeof(float);
    static unsigned long long ok = 0;
    rd_allow_dynamic_lds((const void*)conv3x3s2_pw_h1_kernel<NB, MB, TR, HALF>, lds, ok);
    hipLaunchKernelGGL((conv3x3s2_pw_h1_kernel<NB, MB, TR, HALF>), dim3((unsigned)std::min(ntiles, wgs_per_cu() * n_cu)), dim3(256), lds, s, p, tiles_r, tiles_c, ntiles);
}

void launch_stem34(const Stem34Params& p_in, hipStream_t s) {
    if (p_in.N <= 0 || p_in.OH <= 0 || p_in.OW <= 0) return;
    static const int abl = [] { const char* e = getenv("RD_STEM34_ABL"); return e ? atoi(e) : 0; }();
    Stem34Params p = p_in;
    p.abl = abl;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const int nb = (p.N1 + 31) / 32, mb = (p.N2 + 31) / 32;
    // tile shape: 4 x 32 or 2 x 64 outputs - by the MAP (never by the batch: an image's tiling, hence nothing about its arithmetic, which is
    // per output element anyway, depends on N), whichever covers it with fewer wasted outputs; ties -> 4 x 32
    auto covered = [&](int tr, int tc) { return (long)((p.OH + tr - 1) / tr * tr) * ((p.OW + tc - 1) / tc * tc); };
    static const int force_tr = [] { const char* e = getenv("RD_STEM34_TR"); return e ? atoi(e) : 0; }();
    const bool tr2 = force_tr ? force_tr == 2 : covered(2, 64) < covered(4, 32);
    const bool half = p.N1 <= 32 * nb - 16;          // the last 32-wide block of stem3's channels is at most half full
#define RD_S34_CASE(NBv, MBv, FULL)                                             \
    if (nb == NBv && mb == MBv) {                                               \
        if (half && tr2) launch_stem34_t<NBv, MBv, 2, true>(p, s, n_cu);        \
        else if (half) launch_stem34_t<NBv, MBv, 4, true>(p, s, n_cu);          \
        else if constexpr (FULL) {                                              \
            if (tr2) launch_stem34_t<NBv, MBv, 2, false>(p, s, n_cu);           \
            else launch_stem34_t<NBv, MBv, 4, false>(p, s, n_cu);               \
        }                                                                       \
        return;                                                                 \
    }
    // (two full blocks into three - N1 > 48 and N2 > 64, no network's stem - is not instantiated: it does not fit the register file without
    //  spilling; stem34_shape_ok refuses it)
    RD_S34_CASE(1, 1, true) RD_S34_CASE(1, 2, true) RD_S34_CASE(1, 3, true) RD_S34_CASE(2, 1, true) RD_S34_CASE(2, 2, true) RD_S34_CASE(2, 3, false)
#undef RD_S34_CASE
}

static float s34_scale_exp(const float* w, size_t n, int& ex) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
    ex = 0;
    if (mx > 0.f && std::isfinite(mx)) {
        int x = 0;
        (void)std::frexp(mx, &x);
        ex = 14 - x;
        ex = ex > 100 ? 100 : ex < -100 ? -100 : ex;
    }
    return std::ldexp(1.f, -ex);
}
static void s34_put(float v, int ex, uint16_t& hb, uint16_t& lb) {
    const float vs = std::ldexp(v, ex);
    const _Float16 hh = (_Float16)vs;
    const _Float16 ll = (_Float16)(vs - (float)hh);
    __builtin_memcpy(&hb, &hh, 2);
    __builtin_memcpy(&lb, &ll, 2);
}

// Host: the two weight images.  w3 = folded weights [N1][9 Cin], k = (kh * 3 + kw) * Cin + ci; w4 = [N2][N1].
// img3: slab (pass, tap) = 16 input channels of one tap; per 32-wide output block the hi and the lo fragment (1 KB each): lane (l31, lhi)
// carries w3[32 nb + l31][tap * Cin + 16 pass + 8 lhi + e].  img4: fragment (mb, j, plane): lane (l31, lhi) carries
// w4[32 mb + l31][16 j + 8 (e / 4) + 4 lhi + e % 4] - the channel the kernel's C/D register 8 (j % 2) + e of block j / 2 holds.
// Each matrix is scaled by a power of two so that max |w| lands in [2^13, 2^14); inv[0], inv[1] = the inverse scales.
void prepare_stem34_weights(const float* w3, const float* w4, int Cin, int N1, int N2, std::vector<uint16_t>& img3, std::vector<uint16_t>& img4,
                            float inv[2]) {
    const int K = 9 * Cin, nb_n = (N1 + 31) / 32, mb_n = (N2 + 31) / 32, passes = Cin / 16, K2 = 2 * nb_n - (N1 <= 32 * nb_n - 16 ? 1 : 0);
    int ex3 = 0, ex4 = 0;
    inv[0] = s34_scale_exp(w3, (size_t)N1 * K, ex3);
    inv[1] = s34_scale_exp(w4, (size_t)N2 * N1, ex4);
    img3.assign((size_t)passes * 9 * nb_n * 2 * 512, 0);
    size_t slab = 0;
    for (int pass = 0; pass < passes; ++pass)
        for (int tap = 0; tap < 9; ++tap, ++slab)
            for (int nb = 0; nb < nb_n; ++nb)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = nb * 32 + (lane & 31);
                    if (n >= N1) continue;
                    for (int e = 0; e < 8; ++e) {
                        const int k = tap * Cin + pass * 16 + 8 * (lane >> 5) + e;
                        const size_t base = ((slab * nb_n + nb) * 2) * 512 + (size_t)lane * 8 + e;
                        s34_put(w3[(size_t)n * K + k], ex3, img3[base], img3[base + 512]);
                    }
                }
    img4.assign((size_t)mb_n * K2 * 2 * 512, 0);
    for (int mb = 0; mb < mb_n; ++mb)
        for (int j = 0; j < K2; ++j)
            for (int lane = 0; lane < 64; ++lane) {
                const int n = mb * 32 + (lane & 31);
                if (n >= N2) continue;
                for (int e = 0; e < 8; ++e) {
                    const int c = 16 * j + 8 * (e >> 2) + 4 * (lane >> 5) + (e & 3);
                    if (c >= N1) continue;
                    const size_t base = ((size_t)(mb * K2 + j) * 2) * 512 + (size_t)lane * 8 + e;
                    s34_put(w4[(size_t)n * N1 + c], ex4, img4[base], img4[base + 512]);
                }
            }
}

}  // namespace rd
