// Split-fp16 GEMM for pointwise convolutions as an LDS-DMA pipeline (precision "auto" / "h3").
//
// Same arithmetic as kernels_conv_h3.hip (x = hi + lo*2^-11, three v_mfma_f32_32x32x16_f16 per product, fp32 accumulate),
// different data path.  The register-staged kernel there runs at 14-37 % of the fp16 MFMA pipe: every K tile is a chain
// load -> split in VALU -> ds_write -> barrier -> ds_read -> MFMA in which the phases of the wavefronts sharing a SIMD do
// not overlap, and it cannot keep two K tiles in flight without spilling.  Here
//   * both operands travel global -> LDS by `global_load_lds_dwordx4` (no staging VGPRs, no ds_write, no VALU):
//     the activations as RAW fp32 rows, the weights already split at load time;
//   * three LDS stages of 48 KB, counted `s_waitcnt vmcnt(6)` + a raw `s_barrier` per K tile: one tile is always in
//     flight behind the one being multiplied (6 = DMA instructions per wavefront and tile);
//   * the activation split moves to fragment-read time: a lane reads its 8 fp32 values of a k-step with two
//     ds_read_b128 and splits them in registers (v_cvt_pk_f16_f32 + v_fma_mix*_f16);
//   * a DMA writes wave-uniform base + lane*16, so the LDS images are unpadded; bank conflicts are avoided by XOR-swizzling
//     the 16-byte chunk index with the row on the SOURCE address (A: chunk ^ ((row >> 1) & 7) in 128-byte rows, B:
//     chunk ^ ((row >> 2) & 3) in 64-byte rows - both conflict-free for the ds_read_b128 lane groups of gfx950).
// 256x128 output tile, 8 wavefronts (8 x 1, 32x128 each), persistent workgroups, XCD-contiguous tile order, epilogue and
// range guard as in kernels_conv_h3.hip.
// Tried and dropped: the same pipeline as an implicit GEMM for k x k convolutions (every lane's 16-byte chunk gathered by
// the DMA from its tap, padding taps from a zero page, BN = 32 / 64 / 128): no faster than the register-staged kernel on
// the stem and 3x3 layers (81 vs 92 TFLOP/s at N = 48..64) - with 3..27 K tiles per output tile those layers are bound by
// the per-tile fixed costs, not by the staging path.
#include <cstdio>
#include <type_traits>
#include <cstdlib>

#include "rd_device.h"

namespace rd {


static constexpr int DM = 256, DN = 128, DK = 32;
static constexpr int D_A_BYTES = DM * DK * 4;             // 32 KB raw fp32 activations per stage
static constexpr int D_B_BYTES = DN * DK * 2;             // 8 KB per weight plane (hi, lo)
static constexpr int D_STAGE = D_A_BYTES + 2 * D_B_BYTES; // 48 KB
static constexpr int D_NSTAGE = 3;

__device__ __forceinline__ void dma16(const void* src, unsigned lds_byte_offset, unsigned char* smem) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + lds_byte_offset), 16, 0, 0);
}

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, f16x8& hi, f16x8& lo) {
    f16x4 h0, l0, h1, l1;
    rd_split4(a, h0, l0);
    rd_split4(b, h1, l1);
    hi = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
    lo = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
}

// Epilogue of one 32x32 accumulator tile (16 rows of one output column per lane): combine the two accumulators, bias,
// range guard, activation, residual, store.  The activation switch and the residual test sit OUTSIDE the 16-element loops
// and the residual loads are unconditional (clamped row): per-element branches would serialise the loads and stores.
__device__ __forceinline__ void dma_finish_tile(const ConvParams& p, const f32x16& a1, const f32x16& a2, int mb, int n, float bv,
                                                unsigned& emax) {
    // eight values at a time: sixteen values + sixteen residuals + their 64-bit addresses next to 128 accumulator registers
    // made the kernel spill (23 VGPRs at commit 27b771f); the halves keep the batched, unconditional residual loads
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = half * 8 + e;
            o[e] = fmaf(a2[r], 1.f / 2048.f, a1[r]) + bv;
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
                rs[e] = p.res[(size_t)min(mb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.rld + n];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] += rs[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int r = half * 8 + e;
            const int m = mb + (r & 3) + 8 * (r >> 2);
            if (m < p.M) __builtin_nontemporal_store(o[e], &p.y[(size_t)m * p.yld + n]);
        }
    }
}

// order 0: tile w of step i = XCD-contiguous chunk + i * gridDim / 8 (the N tiles of an M tile run AT THE SAME TIME on neighbouring CUs of one
// XCD: the A tile is fetched from HBM once and fanned out by that XCD's L2);  order 1: workgroup b owns the contiguous run
// [b * per, (b + 1) * per) - the N tiles of an M tile run ONE AFTER THE OTHER on one CU: every CU streams a DIFFERENT A tile on its first
// pass (256 x more distinct bytes in flight against the HBM latency) and re-reads it from L2 / MALL on the others.
__device__ __forceinline__ int gemm_tile_of(int v, int ntiles, int order) {
    if (order == 0) {
        const int xcd = v & 7, j = v >> 3, q = ntiles >> 3, r = ntiles & 7;
        return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
    }
    const int G = (int)gridDim.x, per = (ntiles + G - 1) / G;
    const int b = v % G, i = v / G;
    const int w = b * per + i;
    return (i < per && w < ntiles) ? w : -1;
}
__device__ __forceinline__ bool gemm_has_tile(int v, int ntiles, int order) {
    if (order == 0) return v < ntiles;
    const int G = (int)gridDim.x, per = (ntiles + G - 1) / G;
    return v / G < per && (v % G) * per + v / G < ntiles;
}

// trace (developer, RD_GEMM_TRACE=1; nullptr otherwise): wavefronts 0 and 4 of workgroup 0 stamp s_memtime at the phase boundaries of their
// first 60 K-tile iterations into the 16 KB of LDS behind the three stages; copied out at the end (tools/mb_gemm_trace.py prints the deltas)
#define RD_GSTAMP(slot)                                                                                   \
    do {                                                                                                  \
        if constexpr (TRACE) if (tr && it < 60) {                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                            \
            unsigned long long t_;                                                                        \
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                    \
            if (lane == 0) trl[it * 8 + (slot)] = t_;                                                     \
            __builtin_amdgcn_sched_barrier(0);                                                            \
        }                                                                                                 \
    } while (0)
// IL (round 4): the six LDS-DMA pieces of tile kt + 2 are issued BETWEEN the MFMA groups of tile kt instead of in front of them.  The phase
// stamps of the round-3 form (RD_GEMM_TRACE, tools/mb_gemm_trace.py; K = 768, per K tile and SIMD) read: barrier skew + DMA issue 500-750
// cycles + fragment reads / split 200-450 with the matrix pipe idle, then 2 x 24 MFMAs (1536) - a DMA piece costs its wavefront 80-125 issue
// cycles, and the two wavefronts of a SIMD pay them at the same time.  Between MFMAs the same pieces issue while the pipe works.
template <bool TRACE, bool IL>
__global__ void __launch_bounds__(512) gemm_h3_dma_kernel(ConvParams p, int ntn, int ntiles, int order, unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool tr = TRACE && trace != nullptr && blockIdx.x == 0 && (wave == 0 || wave == 4);
    unsigned long long* trl = reinterpret_cast<unsigned long long*>(smem + D_NSTAGE * D_STAGE + (wave == 4 ? 8192 : 0));
    int it = 0;
    const int wm = wave;      // 8 x 1 wavefronts of 32 x 128: every A row tile is split (2 VALU per element) by ONE wavefront
                              // (round 1: 4 x 2 of 64 x 64, split twice; VALU time is not hidden on this chip)
    const int l31 = lane & 31, lhi = lane >> 5;
    const int K = p.K, KT = (K + DK - 1) / DK, Kp = KT * DK;   // weight rows are zero padded to Kp (split_weights_h3)
    const _Float16* wh = reinterpret_cast<const _Float16*>(p.wh);
    const _Float16* wl = reinterpret_cast<const _Float16*>(p.wl);

    // ---- DMA source addressing of this lane.  A: 4 instructions per wavefront and K tile, instruction j fills tile rows
    // 32*wave + 8*j .. +7 (lane -> row + lane/8, chunk position lane%8).  B: one instruction per plane, rows 16*wave + lane/4.
    // Per-lane address state is kept small (the kernel runs at the 256-VGPR limit with its 128 accumulators): 32-bit element
    // offsets from the uniform bases p.x / wh / wl (the launcher routes tensors of >= 2^31 elements to the 16-wavefront
    // kernel), one chunk offset for the four A instructions (their swizzles differ by a constant: the row's swizzle bits
    // are (4 jj + lane / 16) & 7, i.e. instruction jj flips bit 2 of the chunk index when jj is odd).
    int m0 = 0, n0 = 0;
    unsigned aoff[4];       // row base (elements from p.x); the chunk's channel offset is kc0 ^ (16 * (jj & 1)), clamped per K tile
    const int kc0 = 4 * ((lane & 7) ^ ((lane >> 4) & 7));
    unsigned boff = 0;      // elements from wh / wl
    auto setup_tile = [&](int v) {
        const int w = gemm_tile_of(v, ntiles, order);
        const int tile_m = w / ntn;
        m0 = tile_m * DM;
        n0 = (w - tile_m * ntn) * DN;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int row = 32 * wave + 8 * jj + (lane >> 3);
            const int m = min(m0 + row, p.M - 1);               // rows past M re-read the last row; never stored
            aoff[jj] = (unsigned)m * (unsigned)p.xld;
        }
        const int brow = 16 * wave + (lane >> 2);
        const int bc = (lane & 3) ^ ((brow >> 2) & 3);
        boff = (unsigned)min(n0 + brow, p.Ng - 1) * (unsigned)Kp + 8u * bc;
    };
    auto issue_tile = [&](int kt, int stage) {
        const unsigned base = (unsigned)stage * D_STAGE;
        const int k0 = kt * DK;
#pragma unroll
        // K need not be a multiple of 32: chunks past K re-read the row's last chunk (finite data) and meet zero weights
        for (int jj = 0; jj < 4; ++jj)
            dma16(p.x + (aoff[jj] + (unsigned)min(k0 + (kc0 ^ (16 * (jj & 1))), K - 4)), base + (unsigned)(4 * wave + jj) * 1024u, smem);
        dma16(wh + (boff + (unsigned)k0), base + D_A_BYTES + (unsigned)wave * 1024u, smem);
        dma16(wl + (boff + (unsigned)k0), base + D_A_BYTES + D_B_BYTES + (unsigned)wave * 1024u, smem);
    };
    // one of the six pieces of a tile (0-3: A rows, 4 / 5: weight planes), fenced so that it stays where it is written
    auto issue_piece = [&](int kt, int stage, int piece) {
        const unsigned base = (unsigned)stage * D_STAGE;
        const int k0 = kt * DK;
        __builtin_amdgcn_sched_barrier(0);
        if (piece < 4)
            dma16(p.x + (aoff[piece] + (unsigned)min(k0 + (kc0 ^ (16 * (piece & 1))), K - 4)), base + (unsigned)(4 * wave + piece) * 1024u, smem);
        else if (piece == 4)
            dma16(wh + (boff + (unsigned)k0), base + D_A_BYTES + (unsigned)wave * 1024u, smem);
        else
            dma16(wl + (boff + (unsigned)k0), base + D_A_BYTES + D_B_BYTES + (unsigned)wave * 1024u, smem);
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- fragment addressing (byte offsets inside a stage)
    // A: [ks] first of the two chunks (k-step 1 = k-step 0 with bit 2 of the chunk index flipped, the partner chunk is the
    // address ^ 16); B: column block j sits j * 2048 bytes further (its swizzle bits (Rn >> 2) & 3 do not depend on j) and
    // k-step 1 flips bit 1 of the chunk index: two registers hold all ten fragment addresses, the rest are immediates
    const int a_off0 = (wm * 32 + l31) * 128 + (((2 * lhi) ^ (((wm * 32 + l31) >> 1) & 7)) << 4);
    const int b_off0 = D_A_BYTES + l31 * 64 + ((lhi ^ ((l31 >> 2) & 3)) << 4);
    auto a_off = [&](int ks) { return a_off0 ^ (ks << 6); };
    auto b_off = [&](int j, int ks) { return (b_off0 ^ (ks << 5)) + j * 2048; };

    unsigned emax = 0;
    bool fresh = true;
    int v = blockIdx.x;
    if (!gemm_has_tile(v, ntiles, order)) return;       // (order 1: the last workgroups may own no tile)
    setup_tile(v);
    issue_tile(0, 0);
    if (KT > 1) issue_tile(1, 1);
    for (;;) {
        f32x16 acc1[4], acc2[4];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[j][r] = acc2[j][r] = 0.f;

        int stage = 0;
        auto k_tile = [&](int kt, auto pf_c) {
            // tile kt has landed (this wavefront's own DMAs: all but the newest 6), then everybody's; the barrier also says
            // every wavefront is done reading stage (kt + 2) % 3, which the next DMAs overwrite
            // (the first K tile after an epilogue waits for everything: the epilogue's stores share the counter and are
            //  not ordered against the DMA loads)
            RD_GSTAMP(0);
            if (kt + 1 < KT && (kt > 0 || fresh)) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            RD_GSTAMP(1);
            constexpr bool pf = decltype(pf_c)::value;       // tile kt + 2 exists (compile time: no branch inside the MFMA groups)
            const int pstage = stage >= 1 ? stage - 1 : 2;
            if constexpr (!IL) {
                if constexpr (pf) issue_tile(kt + 2, pstage);
            }
            RD_GSTAMP(2);
            const unsigned char* st = smem + stage * D_STAGE;
            if constexpr (IL) {
                f32x4 xa[2][2];
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    xa[ks][0] = *reinterpret_cast<const f32x4*>(st + a_off(ks));
                    xa[ks][1] = *reinterpret_cast<const f32x4*>(st + (a_off(ks) ^ 16));
                }
                f16x8 bh[4], bl[4], bh1[4], bl1[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 0));
                    bl[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 0) + D_B_BYTES);
                }
                f16x8 ah0, al0, ah1, al1;
                split8(xa[0][0], xa[0][1], ah0, al0);
                split8(xa[1][0], xa[1][1], ah1, al1);
                RD_GSTAMP(3);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[j], acc1[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[j], acc2[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[j], acc2[j], 0, 0, 0);
                    // under these MFMAs: one A piece of tile kt + 2, and the k-step-1 weight fragments of column block j
                    if constexpr (pf) issue_piece(kt + 2, pstage, j);
                    bh1[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 1));
                    bl1[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 1) + D_B_BYTES);
                }
                RD_GSTAMP(4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh1[j], acc1[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl1[j], acc2[j], 0, 0, 0);
                    acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh1[j], acc2[j], 0, 0, 0);
                    if constexpr (pf) { if (j < 2) issue_piece(kt + 2, pstage, 4 + j); }
                }
            } else {
                // both A fragments and the B fragments of k-step 0 are read up front, both splits done before the first MFMA; the B
                // fragments of k-step 1 are read under the MFMAs of k-step 0 (all 16 up front do not fit 256 VGPRs next to 128 accumulators)
                f32x4 xa[2][2];
    #pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    xa[ks][0] = *reinterpret_cast<const f32x4*>(st + a_off(ks));
                    xa[ks][1] = *reinterpret_cast<const f32x4*>(st + (a_off(ks) ^ 16));
                }
                f16x8 ah0, al0, ah1, al1;
                {
                    f16x8 bh[4], bl[4];
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bh[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 0));
                        bl[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 0) + D_B_BYTES);
                    }
                    split8(xa[0][0], xa[0][1], ah0, al0);
                    split8(xa[1][0], xa[1][1], ah1, al1);
                    RD_GSTAMP(3);
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bh[j], acc1[j], 0, 0, 0);
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah0, bl[j], acc2[j], 0, 0, 0);
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al0, bh[j], acc2[j], 0, 0, 0);
                    }
                }
                RD_GSTAMP(4);
                {
                    f16x8 bh[4], bl[4];
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        bh[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 1));
                        bl[j] = *reinterpret_cast<const f16x8*>(st + b_off(j, 1) + D_B_BYTES);
                    }
    #pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bh[j], acc1[j], 0, 0, 0);
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah1, bl[j], acc2[j], 0, 0, 0);
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al1, bh[j], acc2[j], 0, 0, 0);
                    }
                }
            }
            RD_GSTAMP(5);
            if constexpr (TRACE) ++it;
            stage = stage == 2 ? 0 : stage + 1;
        };
        {
            int kt = 0;
            for (; kt + 2 < KT; ++kt) k_tile(kt, std::true_type{});
            for (; kt < KT; ++kt) k_tile(kt, std::false_type{});
        }
        // every wavefront must be past its last LDS read before the next output tile's DMAs land in stages 0 / 1
        asm volatile("s_barrier" ::: "memory");
        const int em0 = m0, en0 = n0;
        const int vnext = v + (int)gridDim.x;
        const bool has_next = gemm_has_tile(vnext, ntiles, order);
        if (has_next) {
            setup_tile(vnext);
            issue_tile(0, 0);
            if (KT > 1) issue_tile(1, 1);
        }
        // ---- epilogue (bias, activation, residual, range guard on the pre-activation value: kernels_conv_h3.hip)
        if constexpr (TRACE) { if (it > 0) { --it; RD_GSTAMP(6); ++it; } }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = en0 + j * 32 + l31;
            if (n >= p.Ng) continue;
            const float bv = p.bias ? p.bias[n] : 0.f;
            dma_finish_tile(p, acc1[j], acc2[j], em0 + wm * 32 + 4 * lhi, n, bv, emax);
        }
        if constexpr (TRACE) { if (it > 0) { --it; RD_GSTAMP(7); ++it; } }
        if (!has_next) break;
        v = vnext;
        fresh = false;
    }
    if (emax >= 0x7f800000u && p.range_flag) rd_raise_flag(p.range_flag);
    if constexpr (TRACE) {
        if (tr && lane == 0)
            for (int i = 0; i < 60 * 8; ++i) trace[(wave == 4 ? 512 : 0) + i] = trl[i];
    }
}
#undef RD_GSTAMP

// ------------------------------------------------------------------------------------------------------------------
// 16-wavefront variant of the same pipeline: 256x128 tile, wavefronts as 8 x 2 with 32x64 each (64 accumulator registers
// instead of 128), so FOUR wavefronts share a SIMD instead of two.  (Round 1 had them 4 x 4 with 64x32: every A row tile was
// split - 2 VALU per element - by four wavefronts; VALU work does not run under another wavefront's MFMAs on this chip
// (tools/probe_mfma_valu.hip), so the redundant splits were 2/3 of the MFMA time.  32x64 halves them at the same LDS traffic.)  The 8-wavefront kernel spends its time in phases that
// do not overlap inside one wavefront (fragment read -> split -> MFMA); with twice the wavefronts the SIMD has another
// wavefront's MFMAs to issue while one splits.  Costs: every A row tile is read and split by 4 wavefronts instead of 2.
// ABL (developer, RD_GEMM_DBG; results garbage): 1 no stores, 2 no MFMAs, 4 no activation / residual in the epilogue, 8 no epilogue
template <int ABL>
__global__ void __launch_bounds__(1024) gemm_h3_dma16_kernel(ConvParams p, int ntn, int ntiles, int order) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // 0..15
    const int wm = wave >> 1, wn = wave & 1;     // 8 x 2 wavefronts of 32 x 64: an A row tile is split by TWO wavefronts (4 x 4 of 64 x 32: four)
    const int l31 = lane & 31, lhi = lane >> 5;
    const int K = p.K, KT = (K + DK - 1) / DK, Kp = KT * DK;
    const _Float16* wh = reinterpret_cast<const _Float16*>(p.wh);
    const _Float16* wl = reinterpret_cast<const _Float16*>(p.wl);

    // DMA per wavefront and K tile: A instructions 2*wave, 2*wave+1 (rows 16*wave .. +15), B instruction `wave` of
    // [hi plane (8) | lo plane (8)]
    int m0 = 0, n0 = 0;
    const float* asrc[2];
    const _Float16* bsrc;
    auto setup_tile = [&](int v) {
        const int w = gemm_tile_of(v, ntiles, order);
        const int tile_m = w / ntn;
        m0 = tile_m * DM;
        n0 = (w - tile_m * ntn) * DN;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const int row = 16 * wave + 8 * jj + (lane >> 3);
            const int c = (lane & 7) ^ ((row >> 1) & 7);
            asrc[jj] = p.x + (size_t)min(m0 + row, p.M - 1) * p.xld;
            (void)c;
        }
        const int brow = 16 * (wave & 7) + (lane >> 2);
        const int bc = (lane & 3) ^ ((brow >> 2) & 3);
        bsrc = ((wave >> 3) ? wl : wh) + (size_t)min(n0 + brow, p.Ng - 1) * Kp + 8 * bc;
    };
    auto issue_tile = [&](int kt, int stage) {
        const unsigned base = (unsigned)stage * D_STAGE;
        const int k0 = kt * DK;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {   // the chunk's channel offset is recomputed here: this kernel has no register to spare
            const int row = 16 * wave + 8 * jj + (lane >> 3);
            const int kc = 4 * ((lane & 7) ^ ((row >> 1) & 7));
            dma16(asrc[jj] + min(k0 + kc, K - 4), base + (unsigned)(2 * wave + jj) * 1024u, smem);
        }
        dma16(bsrc + k0, base + D_A_BYTES + (unsigned)(wave >> 3) * D_B_BYTES + (unsigned)(wave & 7) * 1024u, smem);
    };
    int a_off[2], b_off[2][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        const int R = wm * 32 + l31;
        a_off[ks] = R * 128 + (((2 * (lhi + 2 * ks)) ^ ((R >> 1) & 7)) << 4);
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int Rn = wn * 64 + j * 32 + l31;
            b_off[j][ks] = D_A_BYTES + Rn * 64 + (((lhi + 2 * ks) ^ ((Rn >> 2) & 3)) << 4);
        }
    }

    unsigned emax = 0;
    bool fresh = true;
    int v = blockIdx.x;
    if (!gemm_has_tile(v, ntiles, order)) return;       // (order 1: the last workgroups may own no tile)
    setup_tile(v);
    issue_tile(0, 0);
    if (KT > 1) issue_tile(1, 1);
    for (;;) {
        f32x16 acc1[2], acc2[2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[i][r] = acc2[i][r] = 0.f;
        int stage = 0;
        for (int kt = 0; kt < KT; ++kt) {
            if (kt + 1 < KT && (kt > 0 || fresh)) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (kt + 2 < KT) issue_tile(kt + 2, stage >= 1 ? stage - 1 : 2);
            const unsigned char* st = smem + stage * D_STAGE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                f16x8 ah, al;
                const f32x4 x0 = *reinterpret_cast<const f32x4*>(st + a_off[ks]);
                const f32x4 x1 = *reinterpret_cast<const f32x4*>(st + (a_off[ks] ^ 16));
                split8(x0, x1, ah, al);
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(st + b_off[j][ks]);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(st + b_off[j][ks] + D_B_BYTES);
                    if constexpr (!(ABL & 2)) {
                        acc1[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[j], 0, 0, 0);
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[j], 0, 0, 0);
                        acc2[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[j], 0, 0, 0);
                    } else {
                        acc1[j][0] += (float)ah[0] + (float)bh[0] + (float)bl[0] + (float)al[0];
                    }
                }
            }
            stage = stage == 2 ? 0 : stage + 1;
        }
        asm volatile("s_barrier" ::: "memory");
        const int em0 = m0, en0 = n0;
        const int vnext = v + (int)gridDim.x;
        const bool has_next = gemm_has_tile(vnext, ntiles, order);
        if (has_next) {
            setup_tile(vnext);
            issue_tile(0, 0);
            if (KT > 1) issue_tile(1, 1);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {           // the wavefront's two 32-column blocks
            const int n = en0 + wn * 64 + i * 32 + l31;
            if constexpr ((ABL & 8) != 0) { emax = max(emax, __float_as_uint(acc1[i][0] + acc2[i][0]) & 0x7fffffffu); continue; }
            if (n >= p.Ng) continue;
            // (requesting the bias before the next tile's DMA pieces, so that its wait does not drain them, measured level in a same-box
            //  A/B; so did nothing for the packed rd_gelu2 below, which measured 6 % slower: 128 VGPRs)
            const float bv = p.bias ? p.bias[n] : 0.f;
            const int mb = em0 + wm * 32 + 4 * lhi;
            // eight values at a time (128 VGPRs leave no room for sixteen): activation switch and residual test outside
            // the element loops, residual loads unconditional and batched ahead of the stores
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float o[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = half * 8 + e;
                    o[e] = fmaf(acc2[i][r], 1.f / 2048.f, acc1[i][r]) + bv;
                    emax = max(emax, __float_as_uint(o[e]) & 0x7fffffffu);
                }
                if constexpr ((ABL & 4) != 0) {
                } else if (p.act == ACT_GELU) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = rd_gelu(o[e]);
                } else if (p.act != ACT_NONE) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] = rd_act(o[e], p.act);
                }
                if (p.res && !(ABL & 4)) {
                    float rs[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int r = half * 8 + e;
                        rs[e] = p.res[(size_t)min(mb + (r & 3) + 8 * (r >> 2), p.M - 1) * p.rld + n];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e) o[e] += rs[e];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = half * 8 + e;
                    const int m = mb + (r & 3) + 8 * (r >> 2);
                    if (m < p.M && !((ABL & 1) && o[e] != 12345.678f)) __builtin_nontemporal_store(o[e], &p.y[(size_t)m * p.yld + n]);
                }
            }
        }
        if (!has_next) break;
        v = vnext;
        fresh = false;
    }
    if (emax >= 0x7f800000u && p.range_flag) rd_raise_flag(p.range_flag);
}

bool gemm_h3_dma_applies(const ConvParams& p) {
    static const bool off = [] { const char* e = getenv("RD_H3_DMA"); return e && e[0] == '0'; }();
    return !off && p.wh && p.KH == 1 && p.KW == 1 && p.SH == 1 && p.SW == 1 && p.PT == 0 && p.PL == 0 && p.OH == p.H && p.OW == p.W &&
           p.out_mode == OUT_NHWC && !p.ascale && p.K % 4 == 0 && p.K >= 2 * DK && p.Ng >= 96 && (p.xld % 4) == 0;
}

// which of the two kernels launch_gemm_h3_dma runs for p (the per-op profile names it)
bool gemm_h3_dma_uses16(const ConvParams& p) {
    static const int force16 = [] { const char* e = getenv("RD_H3_DMA16"); return e ? (e[0] == '1' ? 1 : 0) : -1; }();
    // the 8-wavefront kernel addresses its operands with 32-bit element offsets from p.x / p.wh
    const bool fits32 = (unsigned long long)p.M * (unsigned long long)p.xld + (unsigned long long)p.K < (1ull << 32) &&
                        (unsigned long long)p.Ng * (unsigned long long)((p.K + DK - 1) / DK * DK) < (1ull << 32);
    static const bool old_route = [] { const char* e = getenv("RD_GEMM_ROUTE"); return e && e[0] == 'o'; }();      // A/B: round 3's K <= 384 rule
    if (old_route && force16 < 0) return !fits32 || p.K <= 384;
    return !fits32 || (force16 >= 0 ? force16 == 1 : (p.K <= 192 && p.act == ACT_GELU));
}

void launch_gemm_h3_dma(const ConvParams& p, hipStream_t s) {
    const int ntm = (p.M + DM - 1) / DM, ntn = (p.Ng + DN - 1) / DN, ntiles = ntm * ntn;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    const size_t sh = (size_t)D_NSTAGE * D_STAGE;
    static const int order_env = [] { const char* e = getenv("RD_GEMM_ORDER"); return e ? atoi(e) : 0; }();
    const int order = order_env;
    static unsigned long long lds_ok = 0, lds_ok16 = 0;
    rd_allow_dynamic_lds((const void*)gemm_h3_dma_kernel<false, false>, sh, lds_ok);
    // Which kernel (TFLOP/s-equivalent, 16 vs 8 wavefronts, same box, tools/mb_gemm_k16.py, round 4 - the 8-wavefront kernel with its DMA pieces
    // between the MFMA groups): K384 N768 GELU M131072 199 / 207, M43056 221 / 232; K384 N384 240 / 277; K256 N512 184 / 198; K192 N192
    // 113-124 / 128-136; K96 N96 100 / 112; only the short-K GELU layers keep the 16-wavefront tile: K192 N384 GELU 145 / 139, 183 / 169
    // (their epilogue is a third of the tile's time and drains faster from 64 accumulator registers per wavefront).  Round 3 routed every
    // K <= 384 to the 16-wavefront kernel on measurements at M = 26112-52224, where it led by 5 %.
    const bool use16 = gemm_h3_dma_uses16(p);
    if (use16) {
        static const int dbg = [] { const char* e = getenv("RD_GEMM_DBG"); return e ? atoi(e) : 0; }();
#define RD_DMA16(A)                                                                                                  \
    do {                                                                                                             \
        static unsigned long long ok_ = 0;                                                                           \
        rd_allow_dynamic_lds((const void*)gemm_h3_dma16_kernel<A>, sh, ok_);                                         \
        hipLaunchKernelGGL(gemm_h3_dma16_kernel<A>, dim3(ntiles < n_cu ? ntiles : n_cu), dim3(1024), sh, s, p, ntn, ntiles, order); \
    } while (0)
        switch (dbg) {
            case 1: RD_DMA16(1); break;
            case 2: RD_DMA16(2); break;
            case 4: RD_DMA16(4); break;
            case 8: RD_DMA16(8); break;
            case 10: RD_DMA16(10); break;
            default: RD_DMA16(0); break;
        }
#undef RD_DMA16
        (void)lds_ok16;
        return;
    }
    static const bool trace_on = [] { const char* e = getenv("RD_GEMM_TRACE"); return e && e[0] == '1'; }();
    if (trace_on) {      // developer: phase stamps of workgroup 0 (one launch, synchronous; prints cycles per phase to stderr)
        static unsigned long long* tbuf = nullptr;
        static unsigned long long tok = 0;
        if (!tbuf) (void)hipMalloc(&tbuf, 1024 * sizeof(unsigned long long));
        (void)hipMemset(tbuf, 0, 1024 * sizeof(unsigned long long));
        rd_allow_dynamic_lds((const void*)gemm_h3_dma_kernel<true, false>, sh + 16384, tok);
        hipLaunchKernelGGL((gemm_h3_dma_kernel<true, false>), dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), sh + 16384, s, p, ntn, ntiles, order, tbuf);
        (void)hipStreamSynchronize(s);
        unsigned long long h[1024];
        (void)hipMemcpy(h, tbuf, sizeof(h), hipMemcpyDeviceToHost);
        static int printed = 0;
        if (printed++ < 1) {
            const int KT = (p.K + DK - 1) / DK;
            fprintf(stderr, "gemm trace M=%d K=%d N=%d KT=%d (cycles of s_memtime, 100 MHz-independent shader clock): it: wait issue reads+split mfma0 mfma1 | epilogue\n", p.M, p.K, p.Ng, KT);
            for (int w = 0; w < 2; ++w)
                for (int i = 0; i < 60 && h[w * 512 + i * 8 + 5]; ++i) {
                    const unsigned long long* t = h + w * 512 + i * 8;
                    fprintf(stderr, "w%d it %2d: %6llu %6llu %6llu %6llu %6llu | next-top %6llu  epi %6llu\n", w * 4, i, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3],
                            t[5] - t[4], (i + 1 < 60 && h[w * 512 + (i + 1) * 8]) ? h[w * 512 + (i + 1) * 8] - t[5] : 0ull, t[7] > t[6] ? t[7] - t[6] : 0ull);
                }
        }
        return;
    }
    static const int il_env = [] { const char* e = getenv("RD_GEMM_IL"); return e ? atoi(e) : 1; }();
    if (il_env) {
        static unsigned long long ok_il = 0;
        rd_allow_dynamic_lds((const void*)gemm_h3_dma_kernel<false, true>, sh, ok_il);
        hipLaunchKernelGGL((gemm_h3_dma_kernel<false, true>), dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), sh, s, p, ntn, ntiles, order, (unsigned long long*)nullptr);
        return;
    }
    hipLaunchKernelGGL((gemm_h3_dma_kernel<false, false>), dim3(ntiles < n_cu ? ntiles : n_cu), dim3(512), sh, s, p, ntn, ntiles, order, (unsigned long long*)nullptr);
}

}  // namespace rd
