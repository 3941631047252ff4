// Split-fp16 GEMM for pointwise convolutions on ONE fp32 accumulator set (precision "auto" / "h3"), round 5.
//
// kernels_gemm_h3_dma.hip carries a product a*b as three MFMAs into TWO accumulators (hi*hi, and the 2^-11-scaled cross terms):
// 128 accumulator registers for a 32x128 tile per wavefront, ten ds_read_b128 per twelve MFMAs, eight wavefronts of ONE workgroup
// per CU that meet at every barrier and pay their DMA issue / fragment reads / splits at the same time (profiles/r4_gemm_trace.txt:
// the matrix pipe is ~50 % busy inside the K loop and idle during every epilogue).  Here
//   * the low planes are UNSCALED: x = hi + lo with hi = fp16(x), lo = fp16(x - hi).  |lo| <= 2^-11 |x| is a normal fp16 number for
//     |x| >= 2^-3 and a subnormal one (absolute error <= 2^-25) below; gfx950's fp16 matrix cores keep subnormal inputs
//     (tools/probe_mfma.hip).  Weights are pre-scaled per MATRIX by an exact power of two so that max|w| lands in
//     [2^13, 2^14): their low plane is normal down to 2^-17 of the matrix' largest weight; the epilogue multiplies the sum by the
//     inverse power of two (exact).  hi*hi + hi*lo + lo*hi then go into ONE accumulator: a plain fp16 GEMM over a K-concatenated
//     operand (lo*lo <= 2^-22 is dropped as before);
//   * with 64 accumulator registers freed, a wavefront owns 64 x 128 (two row blocks x four column blocks, 128 accumulators): twelve
//     ds_read_b128 per 24 MFMAs, and every activation row is read and split by exactly one wavefront;
//   * a workgroup is FOUR wavefronts (one per SIMD) on a 256 x 128 tile with 80 KB of LDS, so TWO workgroups share a CU
//     (__launch_bounds__(256, 2)).  Their barriers are independent: while one workgroup waits, reads fragments, splits, runs its
//     epilogue or its next tile's prologue, the other one's MFMAs own the matrix pipe of the same SIMDs;
//   * LDS (80 KB): the raw fp32 activation tile of a 32-wide K tile, double buffered (2 x 32 KB, full 128-byte lines from global
//     memory, chunk ^ ((row >> 1) & 7) swizzle on the source address as in the round-3 kernel), and the weights in HALF K tiles
//     (16 wide, 8 KB: four column blocks x {hi, lo} fragments of 1 KB), double buffered.  The weights are L2 resident and arrive
//     one half step ahead; the activations (HBM) one whole K tile ahead;
//   * weights are stored as a stream of 1-KB MFMA fragments in consumption order (prepare_gemm_h1_weights): a weight DMA is a
//     linear copy and a fragment read is ds_read_b128 at fragment + 16 * lane, conflict-free by construction;
//   * every DMA is `buffer_load_dwordx4 ... lds` with the K offset in the scalar offset operand: no VALU per piece;
//   * the DMA stream runs ACROSS output tiles: the next tile's first activation tile and weight half step are requested during the
//     last K tile of the current one and land under its epilogue.
// Needs K % 32 == 0, K >= 64; everything else (edges in M and N, activation, residual, range guard) as the round-3 kernel.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "rd_device.h"

namespace rd {

static constexpr int HM = 256, HN = 128, HK = 32;
static constexpr int H_A = HM * HK * 4;              // 32 KB: raw fp32 activations of one K tile
static constexpr int H_BH = 8 * 1024;                // 8 KB: one 16-wide half step of the weights = 8 fragments of 1 KB
static constexpr int H_LDS = 2 * H_A + 2 * H_BH;     // 80 KB

typedef __amdgpu_buffer_rsrc_t h1_rsrc;

__device__ __forceinline__ h1_rsrc h1_make_rsrc(const void* base) {
    // raw buffer (stride 0), every offset in range; dword 3 = the gfx9 data-format word for untyped 32-bit accesses
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
__device__ __forceinline__ void h1_dma16(h1_rsrc r, unsigned voff, unsigned soff, unsigned lds_byte_offset, unsigned char* smem) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(smem + lds_byte_offset), 16, (int)voff, (int)soff, 0, 0);
}

// x = hi + lo, hi = fp16(x), lo = fp16(x - hi): the difference is exact in fp32 (hi is x rounded to 11 bits), the second rounding
// keeps 11 more bits (or stops at the subnormal spacing 2^-24).  One v_cvt + one v_fma_mix per element.
// `neg1` is -1.0f held in a scalar register the compiler cannot see through: with a literal -1 the fused multiply-add folds to a
// subtraction and the split becomes cvt + cvt back + (packed) sub + cvt (2.5 VALU per element, packed fp32 among them).
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void h1_split8(const f32x4 a, const f32x4 b, float neg1, f16x8& hi, f16x8& lo) {
    // pairs: one packed conversion (v_cvt_pk_f16_f32) whose halves are the f16 sources of the two v_fma_mix{lo,hi}_f16: 1.5 VALU per element
#pragma unroll
    for (int e = 0; e < 4; e += 2) {
        const f16x2 ha = __builtin_convertvector(f32x2{a[e], a[e + 1]}, f16x2);
        const f16x2 hb = __builtin_convertvector(f32x2{b[e], b[e + 1]}, f16x2);
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

// tile v of the persistent loop -> output tile: XCD-contiguous runs (workgroup b sits on XCD b % 8), so the N tiles of an M tile run
// at the same time on one XCD and its L2 fans the activation tile out (kernels_gemm_h3_dma.hip, order 0)
__device__ __forceinline__ int h1_tile_of(int v, int ntiles) {
    const int xcd = v & 7, j = v >> 3, q = ntiles >> 3, r = ntiles & 7;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + j;
}

// Epilogue of one 32 x 32 accumulator block (16 rows of one output column per lane, so that one store instruction writes two full
// 128-byte lines): inverse weight scale, bias, activation, residual, store - eight values at a time (the batched, unconditional
// residual loads of the round-3 kernel).  Tried and dropped (round 5): the MFMAs with the weights as row operand and the channels of
// a block permuted so that a lane holds 16 CONSECUTIVE channels of one pixel - four float4 stores per block instead of sixteen 4-byte
// ones, but every store instruction then touches 32 lines with 32 bytes each: 77 us of stores per launch instead of 26 (M = 131072,
// K = 768, N = 384; profiles/r5_gemm_h1_perf2.txt).
// Range guard: an activation beyond the fp16 range makes EVERY channel of its pixel non-finite (inf * w, or NaN for w = 0), so only
// the first column block of a wavefront carries the check (CHECK).
template <int ABL, bool CHECK>
__device__ __forceinline__ void h1_finish_block(const ConvParams& p, const f32x16& a, int mb, int n, float sinv, float bv, unsigned& emax) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float o[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            o[e] = fmaf(a[half * 8 + e], sinv, bv);
            if constexpr (CHECK) emax = max(emax, __float_as_uint(o[e]) & 0x7fffffffu);
        }
        if constexpr ((ABL & 4) != 0) {
        } else if (p.act == ACT_GELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = rd_gelu(o[e]);
        } else if (p.act == ACT_RELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = fmaxf(o[e], 0.f);
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

// ABL (developer, RD_GEMM1_DBG; results garbage): 1 no stores, 2 no MFMAs, 4 no activation / residual, 8 no epilogue, 16 no DMA
// IL: the DMA pieces sit between the MFMA groups (1) or in front of them (0)
// Tried and dropped (round 5, profiles/r5_gemm_h1_perf3.txt): requesting every row's NEXT 128-byte line into L2 together with the DMA of an
// even K tile (a plain 4-byte load per row whose result is never used, so that the memory side sees 256 contiguous bytes per row): 1-7 %
// slower at every shape, also with the MFMAs switched off - the activation stream is not bound by DRAM page locality.
template <int ABL, bool IL>
__global__ void __launch_bounds__(256, 2) gemm_h1_kernel(ConvParams p, int ntn, int ntiles, unsigned long long* trace) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // 0..3: rows 64 * wave .. + 63 of the tile
    const int l31 = lane & 31, lhi = lane >> 5;
    const int KT = p.K / HK, KS = 2 * KT;                           // K tiles (32 wide), half steps (16 wide)
    const unsigned char* w1 = reinterpret_cast<const unsigned char*>(p.w1);

    // ---- DMA addressing.  A: piece pc (0..7) of a wavefront = tile rows 64 wave + 8 pc .. + 7, lane -> row + lane / 8, 16-byte chunk
    // position lane % 8 holding SOURCE chunk (lane % 8) ^ ((row >> 1) & 7); rows past M re-read the last row (never stored).
    // B: half step s of N tile nt = 8 fragments of 1 KB at ((nt * KS + s) * 8 + f) * 1024, wavefront w copies f = 2 w, 2 w + 1.
    int m0 = 0, n0 = 0;
    unsigned voffA[8];
    h1_rsrc ra, rb;
    const unsigned voffB = (unsigned)lane * 16u;
    auto setup_tile = [&](int v) {
        const int w = h1_tile_of(v, ntiles);
        const int tile_m = w / ntn, tile_n = w - tile_m * ntn;
        m0 = tile_m * HM;
        n0 = tile_n * HN;
        ra = h1_make_rsrc(p.x + (size_t)m0 * p.xld);
        rb = h1_make_rsrc(w1 + (size_t)tile_n * KS * H_BH);
        const int mlast = p.M - 1 - m0;
#pragma unroll
        for (int pc = 0; pc < 8; ++pc) {
            const int row = 64 * wave + 8 * pc + (lane >> 3);
            voffA[pc] = (unsigned)min(row, mlast) * (unsigned)p.xld * 4u + (unsigned)(((lane & 7) ^ ((row >> 1) & 7)) << 4);
        }
    };
    auto issue_A = [&](int t, int buf, int pc) {
        if constexpr (!(ABL & 16)) h1_dma16(ra, voffA[pc], (unsigned)t * (HK * 4), (unsigned)buf * H_A + (unsigned)(8 * wave + pc) * 1024u, smem);
    };
    auto issue_B = [&](int s, int slot, int q) {
        const unsigned f = 2u * wave + q;
        if constexpr (!(ABL & 16)) h1_dma16(rb, voffB, ((unsigned)s * 8u + f) * 1024u, 2u * H_A + (unsigned)slot * H_BH + f * 1024u, smem);
    };

    // ---- fragment addressing.  A (inside a buffer): row R = 64 wave + 32 i + l31, (R >> 1) & 7 = (l31 >> 1) & 7 whatever i and wave
    // are: k step ks flips bit 2 of the chunk index (address ^ 64), the second chunk of the 8 values is address ^ 16, row block i is
    // + 4096.  B: fragment f = 2 j + plane at slot + f * 1024 + 16 lane.
    const int a_base = (64 * wave + l31) * 128 + (((2 * lhi) ^ ((l31 >> 1) & 7)) << 4);
    int b_base_ = 2 * H_A + lane * 16;
    asm volatile("" : "+v"(b_base_));      // one address register + immediate fragment offsets (the compiler kept sixteen addresses otherwise)
    const unsigned char* const bp = smem + b_base_;
    float neg1 = -1.f;
    asm volatile("" : "+s"(neg1));

    unsigned emax = 0;
    // TRACE (ABL bit 5, developer: RD_GEMM1_DBG=32, tools/mb_gemm_h1.py trace): wavefront 0 of every workgroup sums, in shader cycles
    // (s_memtime), what it spends in the two wait + barrier points of a K tile, in the epilogue and in total; written out once at the end
    constexpr bool TRACE = (ABL & 32) != 0;
    unsigned long long tw0 = 0, tw1 = 0, tep = 0, tstart = 0, tmark = 0;
    auto now = [&]() {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        return t_;
    };
    if constexpr (TRACE) tstart = now();
    int v = blockIdx.x;
    if (v >= ntiles) return;
    int ab = 0;                      // activation buffer of the current K tile (runs on across output tiles)
    setup_tile(v);
#pragma unroll
    for (int pc = 0; pc < 8; ++pc) issue_A(0, 0, pc);
    issue_B(0, 0, 0);
    issue_B(0, 0, 1);

    for (;;) {
        f32x16 acc[2][4];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        const int vnext = v + (int)gridDim.x;
        const bool has_next = vnext < ntiles;
        const int em0 = m0, en0 = n0;

        // One K tile = two half steps, ONE loop body for every K tile (the last K tile as a separate copy made the register allocator
        // rename - and spill - the accumulators).  DMA stream of a wavefront, in issue order:
        //   step (t, 0): B(t).h1 x 2 -> slot 1, A(t + 1)[0..7] -> buffer ab ^ 1;   step (t, 1): B(t + 1).h0 x 2 -> slot 0
        // (round-5 first version: half of A(t + 1) in step (t, 1), i.e. half a step before it is needed - shorter than the HBM latency)
        // where, in the last K tile, "t + 1" is the FIRST K tile of the workgroup's next output tile (the descriptors switch after the
        // B(t).h1 pieces) - or, in its very last tile, a harmless re-read of this tile's first pieces, so that the counted waits never
        // change.  Top of (t, 0) waits for everything (A(t), B(t).h0, and after an epilogue its stores), top of (t, 1) for all but the
        // eight newest pieces.  The barrier behind each wait also says that every wavefront is past its reads of what the step's pieces
        // overwrite (buffer ab ^ 1 and slot 1: read during K tile t - 1; slot 0: read during step (t, 0)).
        for (int t = 0; t < KT; ++t) {
            const bool last = t + 1 == KT;
            const int tA = last ? 0 : t + 1, sB = last ? 0 : 2 * t + 2;
            const unsigned char* sa = smem + ab * H_A;
            // ---------------- step (t, 0)
            if constexpr (TRACE) tmark = now();
            asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if constexpr (TRACE) tw0 += now() - tmark;
            // the k-step-0 activations are read and split first; the k-step-1 reads reuse their registers
            f16x8 ah[2], al[2];
            f16x8 bh[4], bl[4];
            {
                f32x4 x0[2][2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    x0[i][0] = *reinterpret_cast<const f32x4*>(sa + a_base + i * 4096);
                    x0[i][1] = *reinterpret_cast<const f32x4*>(sa + ((a_base + i * 4096) ^ 16));
                }
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8*>(bp + (2 * j) * 1024);
                    bl[j] = *reinterpret_cast<const f16x8*>(bp + (2 * j + 1) * 1024);
                }
                h1_split8(x0[0][0], x0[0][1], neg1, ah[0], al[0]);
                h1_split8(x0[1][0], x0[1][1], neg1, ah[1], al[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            f32x4 x1[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                x1[i][0] = *reinterpret_cast<const f32x4*>(sa + ((a_base + i * 4096) ^ 64));
                x1[i][1] = *reinterpret_cast<const f32x4*>(sa + ((a_base + i * 4096) ^ 64 ^ 16));
            }
            auto piece0 = [&](int g) {      // DMA pieces behind MFMA group g of step (t, 0): B(t).h1, then ALL of A(t + 1) - a whole K tile ahead
                if (g == 0) {
                    issue_B(2 * t + 1, 1, 0);
                    issue_B(2 * t + 1, 1, 1);
                } else if (g < 5) {
                    if (g == 1 && last && has_next) setup_tile(vnext);      // (em0 / en0 keep this tile's origin for the epilogue)
                    issue_A(tA, ab ^ 1, 2 * g - 2);
                    issue_A(tA, ab ^ 1, 2 * g - 1);
                }
            };
            auto piece1 = [&](int g) {      // ... of step (t, 1): the next half step of the weights (L2 resident, half a step ahead is enough)
                if (g < 2) issue_B(sB, 0, g);
            };
            // six groups of four MFMAs; the three terms of a product are eight MFMAs apart (no back-to-back accumulator chain)
            auto mfma_step = [&](const f16x8 (&xh)[2], const f16x8 (&xl)[2], const f16x8 (&wh)[4], const f16x8 (&wl)[4], auto piece, auto between) {
                if constexpr (!IL) {
#pragma unroll
                    for (int g = 0; g < 6; ++g) piece(g);
                }
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    const int term = g >> 1, i = g & 1;
                    if constexpr (!(ABL & 2)) {
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(term == 2 ? xl[i] : xh[i], term == 1 ? wl[j] : wh[j], acc[i][j], 0, 0, 0);
                    } else {
                        acc[i][0][0] += (float)xh[i][0] + (float)xl[i][1] + (float)wh[g & 3][0] + (float)wl[g & 3][1];
                    }
                    if constexpr (IL) {
                        __builtin_amdgcn_sched_barrier(0);
                        piece(g);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    between(g);
                }
            };
            f16x8 ah1[2], al1[2];
            mfma_step(ah, al, bh, bl, piece0, [&](int g) {
                // the second k step's split rides under the MFMAs of the first
                if (g == 2) h1_split8(x1[0][0], x1[0][1], neg1, ah1[0], al1[0]);
                if (g == 3) h1_split8(x1[1][0], x1[1][1], neg1, ah1[1], al1[1]);
            });
            // ---------------- step (t, 1)
            if constexpr (TRACE) tmark = now();
            asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            if constexpr (TRACE) tw1 += now() - tmark;
            f16x8 ch[4], cl[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                ch[j] = *reinterpret_cast<const f16x8*>(bp + H_BH + (2 * j) * 1024);
                cl[j] = *reinterpret_cast<const f16x8*>(bp + H_BH + (2 * j + 1) * 1024);
            }
            mfma_step(ah1, al1, ch, cl, piece1, [&](int) {});
            ab ^= 1;
        }
        // ---- epilogue (the next tile's first pieces are in flight)
        if constexpr (TRACE) tmark = now();
        if constexpr ((ABL & 8) != 0) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) emax = max(emax, __float_as_uint(acc[i][j][0] + acc[i][j][7]) & 0x7fffffffu);
        } else {
            const float sinv = p.w1_inv;
            const bool interior = em0 + HM <= p.M && en0 + HN <= p.Ng;        // (wave-uniform)
            if (interior) {
                // Interior tile (all but the last row / column of tiles): no bounds test, and every load / store is "uniform row base
                // (SGPR pair) + this lane's 32-bit offset + immediate column block": the general path below spends a 64-bit multiply-add,
                // a 64-bit shift-add, a compare and an EXEC save / branch on each of its 128 four-byte stores.
                // (buffer addressing: descriptor at the tile's first row, lane offset in the vector operand, row offset in the scalar one,
                //  column block as immediate - zero vector instructions per access; aux 2 = nt, like the general path's stores)
                // (the lane's offsets are re-derived here on purpose: computed from an opaque copy of the lane id they cannot be hoisted out
                //  of the tile loop, where they would sit in registers through the K loop - the kernel has none to spare)
                int lane_e = lane;
                asm volatile("" : "+v"(lane_e));
                const int l31e = lane_e & 31, lhie = lane_e >> 5;
                const unsigned yoff = ((unsigned)(4 * lhie) * (unsigned)p.yld + (unsigned)(en0 + l31e)) * 4u;
                const unsigned roff = ((unsigned)(4 * lhie) * (unsigned)p.rld + (unsigned)(en0 + l31e)) * 4u;
                const h1_rsrc ry = h1_make_rsrc(p.y + (size_t)(em0 + 64 * wave) * p.yld);
                const h1_rsrc rr = h1_make_rsrc((p.res ? p.res : p.y) + (size_t)(em0 + 64 * wave) * (p.res ? p.rld : p.yld));
                float bv[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) bv[j] = p.bias ? p.bias[en0 + 32 * j + l31e] : 0.f;
#pragma unroll
                for (int i = 0; i < 2; ++i) {
#pragma unroll
                    for (int r4 = 0; r4 < 16; r4 += 2) {          // two rows x four column blocks at a time (four rows: two spilled registers)
                        float o[2][4];
#pragma unroll
                        for (int q = 0; q < 2; ++q)
#pragma unroll
                            for (int j = 0; j < 4; ++j) o[q][j] = fmaf(acc[i][j][r4 + q], sinv, bv[j]);
#pragma unroll
                        for (int q = 0; q < 2; ++q) emax = max(emax, __float_as_uint(o[q][0]) & 0x7fffffffu);       // column block 0 carries the range check
                        if constexpr ((ABL & 4) != 0) {
                        } else if (p.act == ACT_GELU) {
                            if constexpr ((ABL & 64) != 0) {          // A/B (RD_GEMM1_DBG=64, valid results): two GELUs per packed-fp32 instruction
#pragma unroll
                                for (int q = 0; q < 2; ++q)
#pragma unroll
                                    for (int j = 0; j < 4; j += 2) {
                                        const f32x2 g = rd_gelu2(f32x2{o[q][j], o[q][j + 1]});
                                        o[q][j] = g[0];
                                        o[q][j + 1] = g[1];
                                    }
                            } else {
#pragma unroll
                                for (int q = 0; q < 2; ++q)
#pragma unroll
                                    for (int j = 0; j < 4; ++j) o[q][j] = rd_gelu(o[q][j]);
                            }
                        } else if (p.act == ACT_RELU) {
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[q][j] = fmaxf(o[q][j], 0.f);
                        } else if (p.act != ACT_NONE) {
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[q][j] = rd_act(o[q][j], p.act);
                        }
                        const int row0 = 32 * i + 8 * (r4 >> 2) + (r4 & 3);       // + q: wave-uniform row inside the wavefront's 64 (lhi sits in the lane offsets)
                        if (p.res && !(ABL & 4)) {
                            float rs[2][4];
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const unsigned so = (unsigned)(row0 + q) * (unsigned)p.rld * 4u;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    rs[q][j] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (int)(roff + 128u * j), (int)so, 0));
                            }
#pragma unroll
                            for (int q = 0; q < 2; ++q)
#pragma unroll
                                for (int j = 0; j < 4; ++j) o[q][j] += rs[q][j];
                        }
                        if (!((ABL & 1) && o[0][0] != 12345.678f)) {
#pragma unroll
                            for (int q = 0; q < 2; ++q) {
                                const unsigned so = (unsigned)(row0 + q) * (unsigned)p.yld * 4u;
#pragma unroll
                                for (int j = 0; j < 4; ++j)
                                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[q][j]), ry, (int)(yoff + 128u * j), (int)so, 2);
                            }
                        }
                    }
                }
            } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = en0 + j * 32 + l31;
                const bool live = n < p.Ng;
                if (j > 0 && !live) continue;       // (block 0 always runs: it carries the range check of the wavefront's 64 rows)
                const float bv = (p.bias && live) ? p.bias[n] : 0.f;
                const int nn = live ? n : en0;      // dead columns of block 0: rows checked, nothing stored (mb = M)
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int mb = live ? em0 + 64 * wave + 32 * i + 4 * lhi : p.M;
                    if (j == 0) h1_finish_block<ABL, true>(p, acc[i][j], mb, nn, sinv, bv, emax);
                    else h1_finish_block<ABL, false>(p, acc[i][j], mb, nn, sinv, bv, emax);
                }
            }
            }
        }
        if constexpr (TRACE) {
            asm volatile("s_nop 0" ::: "memory");
            tep += now() - tmark;
        }
        if (!has_next) break;
        v = vnext;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // (the last tile's re-read pieces)
    if constexpr (TRACE) {
        if (trace && wave == 0 && lane == 0) {
            unsigned long long* o = trace + (size_t)blockIdx.x * 4;
            o[0] = now() - tstart; o[1] = tw0; o[2] = tw1; o[3] = tep;
        }
    }
    if (emax >= 0x7f800000u && p.range_flag) rd_raise_flag(p.range_flag);
}

// ------------------------------------------------------------------------------------------------------------------
// Host side: the weight image.  ONE power-of-two scale per matrix: e with max |w| * 2^e in [2^13, 2^14) (all zero: e = 0),
// w' = w * 2^e split into hi = fp16(w'), lo = fp16(w' - hi); returns 2^-e (exact), which the epilogue multiplies the sums by.  lo is a
// normal fp16 number for weights down to 2^-17 of the matrix' largest and a subnormal with absolute error 2^-25 (2^-39 of the largest)
// below.  Image = [N tiles of 128][K / 16 half steps][column block j = 0..3][plane hi, lo][lane = 32 lhi + l31][8 halfs]: the 16 bytes
// lane (l31, lhi) feeds to v_mfma_f32_32x32x16_f16 as column 128 nt + 32 j + l31 of its second operand, k = 16 s + 8 lhi .. + 7.
// Channels past N are zero.
float prepare_gemm_h1_weights(const float* w, int N, int K, std::vector<uint16_t>& img) {
    const int KS = K / 16, ntn = (N + HN - 1) / HN;
    img.assign((size_t)ntn * KS * (H_BH / 2), 0);
    float mx = 0.f;
    for (size_t i = 0; i < (size_t)N * K; ++i) mx = std::fmax(mx, std::fabs(w[i]));
    int ex = 0;
    if (mx > 0.f && std::isfinite(mx)) {
        int x = 0;
        (void)std::frexp(mx, &x);           // mx = m * 2^x, m in [0.5, 1)
        ex = 14 - x;
        ex = ex > 100 ? 100 : ex < -100 ? -100 : ex;
    }
    for (int nt = 0; nt < ntn; ++nt)
        for (int s = 0; s < KS; ++s)
            for (int j = 0; j < 4; ++j)
                for (int lane = 0; lane < 64; ++lane) {
                    const int n = nt * HN + j * 32 + (lane & 31);
                    if (n >= N) continue;
                    for (int e8 = 0; e8 < 8; ++e8) {
                        const int k = 16 * s + 8 * (lane >> 5) + e8;
                        const float vs = std::ldexp(w[(size_t)n * K + k], ex);
                        const _Float16 hh = (_Float16)vs;
                        const _Float16 ll = (_Float16)(vs - (float)hh);
                        uint16_t hb, lb;
                        __builtin_memcpy(&hb, &hh, 2);
                        __builtin_memcpy(&lb, &ll, 2);
                        const size_t base = (((size_t)nt * KS + s) * 8 + 2 * j) * 512 + (size_t)lane * 8 + e8;
                        img[base] = hb;
                        img[base + 512] = lb;
                    }
                }
    return std::ldexp(1.f, -ex);
}

bool gemm_h1_shape_ok(int K, int cout) { return K % HK == 0 && K >= 2 * HK && cout >= 96; }

bool gemm_h1_applies(const ConvParams& p) {
    static const bool off = [] { const char* e = getenv("RD_GEMM_H1"); return e && e[0] == '0'; }();
    return !off && p.w1 && p.w1_inv > 0.f && p.KH == 1 && p.KW == 1 && p.SH == 1 && p.SW == 1 && p.PT == 0 && p.PL == 0 && p.OH == p.H && p.OW == p.W &&
           p.out_mode == OUT_NHWC && !p.ascale && gemm_h1_shape_ok(p.K, p.Ng) && (p.xld % 4) == 0 &&
           (unsigned long long)HM * (unsigned long long)p.xld * 4ull + (unsigned long long)p.K * 4ull < (1ull << 31);
}

void launch_gemm_h1(const ConvParams& p, hipStream_t s) {
    const int ntm = (p.M + HM - 1) / HM, ntn = (p.Ng + HN - 1) / HN, ntiles = ntm * ntn;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return rd_cu_budget(n > 0 ? n : 256);
    }();
    static const int per_cu = [] { const char* e = getenv("RD_GEMM1_WGS"); return e ? atoi(e) : 2; }();     // developer A/B: 1 = one workgroup per CU
    static const int dbg = [] { const char* e = getenv("RD_GEMM1_DBG"); return e ? atoi(e) : 0; }();
    // IL (the DMA pieces between the MFMA groups instead of in front of them): level within +-3 % at the step's shapes, ahead at M = 131072 /
    // K = 768 (283 vs 291 us), behind at M = 65536 (118 vs 111), K = 384 / N = 384 (136 vs 129), K = 192 (99 vs 92); off by default
    static const int il = [] { const char* e = getenv("RD_GEMM1_IL"); return e ? atoi(e) : 0; }();
    const int grid = ntiles < per_cu * n_cu ? ntiles : per_cu * n_cu;
#define RD_H1(A, I)                                                                                                   \
    do {                                                                                                              \
        static unsigned long long ok_ = 0;                                                                            \
        rd_allow_dynamic_lds((const void*)gemm_h1_kernel<A, I>, H_LDS, ok_);                                          \
        hipLaunchKernelGGL((gemm_h1_kernel<A, I>), dim3(grid), dim3(256), H_LDS, s, p, ntn, ntiles, (unsigned long long*)nullptr); \
    } while (0)
    if (dbg == 0 && il) { RD_H1(0, true); return; }
    if (dbg == 0) { RD_H1(0, false); return; }
    switch (dbg) {
        case 1: RD_H1(1, false); break;
        case 2: RD_H1(2, false); break;
        case 8: RD_H1(8, false); break;
        case 16: RD_H1(16, false); break;
        case 64: RD_H1(64, false); break;
        case 32: case 34: {      // phase cycles of every workgroup's wavefront 0 (one synchronous launch per call; stderr)
            static unsigned long long* tbuf = nullptr;
            if (!tbuf) (void)hipMalloc(&tbuf, 4096 * 4 * sizeof(unsigned long long));
            (void)hipMemsetAsync(tbuf, 0, 4096 * 4 * sizeof(unsigned long long), s);
            static unsigned long long ok32 = 0, ok34 = 0;
            if (dbg == 32) {
                rd_allow_dynamic_lds((const void*)gemm_h1_kernel<32, false>, H_LDS, ok32);
                hipLaunchKernelGGL((gemm_h1_kernel<32, false>), dim3(grid), dim3(256), H_LDS, s, p, ntn, ntiles, tbuf);
            } else {
                rd_allow_dynamic_lds((const void*)gemm_h1_kernel<34, false>, H_LDS, ok34);
                hipLaunchKernelGGL((gemm_h1_kernel<34, false>), dim3(grid), dim3(256), H_LDS, s, p, ntn, ntiles, tbuf);
            }
            static int printed = 0;
            if (printed < 24 && (++printed % 3) == 0) {      // every third launch of a shape list, warmed up
                (void)hipStreamSynchronize(s);
                std::vector<unsigned long long> h((size_t)grid * 4);
                (void)hipMemcpy(h.data(), tbuf, h.size() * 8, hipMemcpyDeviceToHost);
                double sum[4] = {0, 0, 0, 0}, mx = 0;
                for (int b = 0; b < grid; ++b) {
                    for (int k = 0; k < 4; ++k) sum[k] += (double)h[(size_t)b * 4 + k];
                    mx = std::fmax(mx, (double)h[(size_t)b * 4]);
                }
                const double tiles_per_wg = (double)ntiles / grid, kt = p.K / HK;
                fprintf(stderr, "h1 trace M=%d K=%d N=%d dbg=%d: per workgroup (mean of %d, shader cycles): total %.0f (max %.0f) = %.1f tiles x [ %d K tiles x "
                        "( wait A+B %.0f + wait B %.0f + rest %.0f ) + epilogue %.0f ]\n", p.M, p.K, p.Ng, dbg, grid, sum[0] / grid, mx, tiles_per_wg, (int)kt,
                        sum[1] / grid / tiles_per_wg / kt, sum[2] / grid / tiles_per_wg / kt,
                        (sum[0] - sum[1] - sum[2] - sum[3]) / grid / tiles_per_wg / kt, sum[3] / grid / tiles_per_wg);
            }
            break;
        }
        default: RD_H1(0, false); break;
    }
#undef RD_H1
}

}  // namespace rd
