// Weight-streaming fused PPLCNetV4 channel mixer ("ws") on the split-fp16 matrix cores - round-2 replacement of
// lc_mixer_h3_kernel for C = 96 / 192 (rec_lcnetv4.py:226-236:  Y = X' + W2 . GELU(W1 . X' + b1) + b2,  X' = X * gate).
//
// Why a new structure: the round-1 kernel kept a 128-pixel X tile in LDS (98 KB at C = 192), which left room
// for ONE 4-wavefront workgroup per CU at 456 VGPRs: one wavefront per SIMD, every phase (tile load, GEMM1, GELU, GEMM2,
// store) serialised, matrix pipe 22 % busy.  Here the ACTIVATIONS LIVE IN REGISTERS and LDS holds nothing but the weight
// stream:
//   * a wavefront owns 16 pixels.  Both GEMMs are computed transposed on v_mfma_f32_16x16x32_f16 with the weights as the A
//     operand (rows = hidden units / output channels, straight from LDS) and the activations as the B operand (columns =
//     pixels, from registers):  H^T[32 x 16] = W1c . X^T,   Y^T[C x 16] += W2c . GELU(H)^T.
//     X^T as B fragments is C/4 VGPRs (48 at C = 192), Y^T accumulators C/4 x 2, so a wavefront needs < 256 VGPRs and
//     EIGHT wavefronts (two per SIMD) share a CU: one wavefront's VALU (GELU, split) and waits hide under the other's MFMAs.
//   * the k index of an MFMA is free as long as A and B agree, so W1's input channels are PERMUTED at weight-preparation
//     time such that the 8 k-slots a lane feeds to GEMM1 are the same 2 x 4 contiguous channels it owns in the C/D layout
//     of Y^T: one float4 load per 16-channel block serves the split, the same addresses serve the residual and the store.
//     The C/D registers of H^T are directly the B fragment of GEMM2 (W2's hidden columns permuted to match), as before.
//   * the weights are pre-arranged on the host as a stream of 1-KB MFMA fragments in exactly the order they are consumed:
//     an LDS stage is filled by a LINEAR global_load_lds copy (no swizzle, no address arithmetic) and every fragment read
//     is ds_read_b128 at stage + fragment * 1024 + lane * 16: conflict-free by construction.
//   * persistent workgroups (one per CU) loop over pixel tiles while the weight stream cycles through a 3-stage LDS ring
//     (2 stages = 96 KB in flight at C = 192): the pipeline never drains between tiles.  Stage q holds
//     [W1 chunk q+1 | W2 chunk q]: a step runs GELU(chunk q) beside GEMM1(chunk q+1), then GEMM2(chunk q) - the MFMAs of the
//     next chunk cover the VALU of this one inside a wavefront too.
//   * PER-WAVEFRONT PHASES: the weight stream is cyclic, so a wavefront may start a tile at any chunk.  Wavefront w of the
//     grid starts its tiles at phase (5 w) mod NC of the weight cycle.  Measured reason (ablations of the lock-step
//     version, M = 105 600): with every wavefront switching tiles at the same step the 243 MB of tile traffic arrives in
//     four bursts during which nothing computes (40 of 168 us); spread over all NC phases the memory system sees a uniform
//     ~2 TB/s and a wavefront's tile switch (epilogue, residual re-read, next X tile) runs under the GEMMs of the other
//     wavefront of its SIMD.  Tiles are dealt statically: T steps give a wavefront of phase ph floor((T - 1 - ph) / NC)
//     tiles, the host picks the smallest T that covers M (no more steps than the lock-step version needs).
// Arithmetic ("scaled split", one accumulator): an operand v is carried as hi = fp16(v * s), lo = fp16(v * s - hi) with a
// POWER-OF-TWO scale s, and a product is three MFMAs (hi.hi + hi.lo + lo.hi) into ONE fp32 accumulator that is multiplied by
// the exact inverse scales in the epilogue.  The gfx950 fp16 matrix cores keep subnormal inputs (tools/probe_mfma.hip), so lo
// needs no 2^11 pre-scale and the second accumulator of the round-1 kernels (half of the accumulator registers) goes away.
// lo is a normal fp16 number - i.e. the operand keeps 22 significant bits - whenever |v * s| >= 2^-3, otherwise its absolute
// error is <= 2^-25 / s.  Scales: weights per matrix, max |w| * s in [2^13, 2^14) (fixed at load time; elements down to
// 2^-17 of the largest keep 22 bits); X' PER PIXEL, max over the pixel's channels scaled into [2^13, 2^14) (block floating
// point: no activation of any magnitude can leave the fp16 range, and precision is relative to the pixel's own largest
// channel); GELU outputs by a fixed 2^4 (guarded: |h| >= 4094 raises the range flag -> fp32 re-run, as in round 1).
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "rd_device.h"

namespace rd {

static constexpr int WS_PX = 16;      // pixels per wavefront
static constexpr int WS_WAVES = 8;    // wavefronts per workgroup
static constexpr int WS_HC = 32;      // hidden units per chunk
static constexpr int WS_NSTAGE = 3;

template <int C>
struct WsGeom {
    static constexpr int KS = C / 32;              // k-steps of GEMM1 (32 input channels each)
    static constexpr int NB = C / 16;              // 16-channel output blocks of GEMM2
    static constexpr int NC = 2 * C / WS_HC;       // hidden chunks = steps per tile
    static constexpr int W1_FRAGS = 2 * KS * 2;    // (k-step s, hidden block b, plane)
    static constexpr int W2_FRAGS = NB * 2;        // (output block n, plane)
    static constexpr int STAGE_FRAGS = W1_FRAGS + W2_FRAGS;
    static constexpr int STAGE_BYTES = STAGE_FRAGS * 1024;
    static constexpr int PIECES = STAGE_FRAGS / WS_WAVES;   // DMA instructions per wavefront and stage
    static_assert(STAGE_FRAGS % WS_WAVES == 0, "stage must split evenly over the wavefronts");
    static constexpr size_t LDS_BYTES = (size_t)WS_NSTAGE * STAGE_BYTES + 3 * C * sizeof(float);
};

// one wait + workgroup barrier per step: my DMA pieces of the stage about to be read have landed (the PIECES issued
// last step may still be in flight), then everybody's have, and everybody is done reading the stage about to be refilled
template <int PIECES>
__device__ __forceinline__ void ws_step_sync(bool next_in_flight) {
    // (loads complete in issue order, so "at most PIECES operations outstanding" implies the older stage has landed
    //  whatever the stores of an epilogue in between do; on the last step nothing younger is in flight: wait for all)
    if (!next_in_flight) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (PIECES == 6) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    else if constexpr (PIECES == 3) asm volatile("s_waitcnt vmcnt(3)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
}

// the same with an allowance of `extra` (0 / 12 / 24) younger operations: loads or stores the previous step issued BEHIND its DMA
// pieces (C = 192, PIECES = 6)
// (in the PF kernel every step issues its six DMA pieces, also the last two whose stages nobody reads: the count never has a
// special case, and the code between a register load and its wait stays free of branches)
template <int EXTRA, bool BARRIER = true>
__device__ __forceinline__ void ws_step_sync_pf() {
    static_assert(EXTRA == 0 || EXTRA == 12 || EXTRA == 24, "");
    if constexpr (!BARRIER) {      // (ablation)
        if constexpr (EXTRA == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (EXTRA == 12) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(30)" ::: "memory");
    } else if constexpr (EXTRA == 0) asm volatile("s_waitcnt vmcnt(6)\n\ts_barrier" ::: "memory");
    else if constexpr (EXTRA == 12) asm volatile("s_waitcnt vmcnt(18)\n\ts_barrier" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(30)\n\ts_barrier" ::: "memory");
}

static constexpr float WS_SH = 16.f;   // fixed scale of the hidden activations (GELU outputs)

// v -> (hi, lo) of v * s
__device__ __forceinline__ void ws_split(float v, float s, _Float16& hi, _Float16& lo) {
    const float vs = v * s;
    hi = (_Float16)vs;
    lo = (_Float16)(vs - (float)hi);   // exact difference, rounded once (only in the fp16 subnormal range)
}

// inv1 = 1 / scale(W1), inv2 = 1 / (scale(W2) * WS_SH): exact powers of two
// ABL: microbenchmark ablation bits (results are garbage): 1 no weight DMA after the first two stages, 2 no MFMAs,
// 4 no fragment reads and no MFMAs, 8 no GELU, 16 no workgroup barrier
// PF (round 3): no synchronous memory round trip at a tile switch.  Ablations of the round-2 kernel at M = 105 600 ADD UP (tile IO
// alone 38 us, + MFMAs and fragment reads 60, + GELU 27, + barriers 33, + weight DMA 14 = 172): all eight wavefronts of a CU switch
// tiles in the same step and sit through two HBM round trips (residual re-read, next X tile) plus the in-order retirement of the
// stores in between.  Here the X registers themselves are the landing buffer: GEMM1 of a tile's last chunk is the last reader of
// the split X fragments, so from the middle of step NC-1 to the switch those 48 VGPRs are free.  They first receive the raw fp32 tile
// again (issued behind the last GEMM1, landing under GEMM2 of that step) which is FOLDED INTO THE ACCUMULATORS at the start of the
// switch step (Y^T += x' * scale(W2) * WS_SH, exact power of two; only the last chunk's three products are accumulated on top of the
// large term, so the rounding stays at the fp32 level), then the NEXT tile (issued right after the fold, landing under the last
// chunk's GELU + GEMM2), which is split in place after the epilogue's stores have been issued.  No load is ever waited for behind a
// store: the step barrier's vmcnt allows for what the previous step issued behind its DMA pieces.
// Measured and dropped on top of this form: GELU(q) issued instruction by instruction in the shadow of GEMM1(q + 1) (one MFMA, then
// 1 - 3 VALU instructions of the four value pairs' GELU pipelines, fenced with sched_barrier so the order holds in the ISA):
// 151 - 154 us against 149 - the VALU time of a SIMD adds to its MFMA time however it is arranged (probe_mfma_valu).
template <int C, bool GATED, bool KEEPX, int ABL = 0, bool PF = false>
__global__ void __launch_bounds__(512, 2) lc_mixer_ws_kernel(MixerParams p, const unsigned char* __restrict__ wimg, int n_tiles,
                                                             int T_total, int ph_mul, int ph_unit, float inv1, float inv2) {
    using G = WsGeom<C>;
    constexpr int KS = G::KS, NB = G::NB, NC = G::NC;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    float* B1s = reinterpret_cast<float*>(lds + WS_NSTAGE * G::STAGE_BYTES);   // [2C] hidden biases, [C] output biases
    float* B2s = B1s + 2 * C;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int px = lane & 15, g = lane >> 4;

    for (int i = tid; i < 2 * C; i += 512) B1s[i] = p.b1[i];
    for (int i = tid; i < C; i += 512) B2s[i] = p.b2[i];
    __syncthreads();

    // this wavefront's phase in the weight cycle and its (contiguous) run of 16-pixel tiles: a wavefront of phase ph' fits
    // floor((T - 1 - ph') / NC) tiles into T steps; phases repeat every NC wavefronts, so the run start is a closed form
    // dealing unit: a wavefront (ph_unit = 1) or a whole workgroup (ph_unit = 8: its wavefronts switch tiles together, so a
    // switch stalls nobody else at the step barrier, while different workgroups - nothing couples them - switch at different
    // steps and the tile traffic of the chip is spread over the weight cycle instead of arriving in bursts)
    const int du = ((int)blockIdx.x * WS_WAVES + wave) / ph_unit, sub = ((int)blockIdx.x * WS_WAVES + wave) % ph_unit;
    constexpr int TAIL = 1;                     // steps a tile needs beyond its NC chunk steps
    auto n_of = [&](int w) { return max(0, (T_total - TAIL - (w * ph_mul) % NC) / NC); };
    const int ph = (du * ph_mul) % NC;
    int per_cycle = 0, before = 0;
    for (int j = 0; j < NC; ++j) {
        per_cycle += n_of(j);
        before += j < du % NC ? n_of(j) : 0;
    }
    const int tile_base = (du / NC) * per_cycle + before;            // in unit tiles (ph_unit wavefront tiles each)
    const int n_unit_tiles = (n_tiles + ph_unit - 1) / ph_unit;
    const int R = max(0, min(n_of(du % NC), n_unit_tiles - tile_base));    // tiles of this wavefront

    // ---- weight stream: stage q of the image = [W1 fragments of chunk (q+1) % NC | W2 fragments of chunk q]
    // Every CU streams the SAME bytes at about the same time; walking them in the same order would make all 32 CUs of an
    // XCD hit the same L2 channel at the same moment.  Each workgroup therefore starts its walk at a different fragment.
    const int rot = (int)((blockIdx.x * 7u) % (unsigned)G::STAGE_FRAGS);
    auto issue_stage = [&](int t) {
        const int q = (t + NC - 1) % NC;
        const unsigned char* src = wimg + (size_t)q * G::STAGE_BYTES + lane * 16;
        unsigned char* dst = lds + (t % WS_NSTAGE) * G::STAGE_BYTES;
#pragma unroll
        for (int u = 0; u < G::PIECES; ++u) {
            int f = wave + u * WS_WAVES + rot;
            f = f >= G::STAGE_FRAGS ? f - G::STAGE_FRAGS : f;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
        }
    };
    issue_stage(0);
    issue_stage(1);

    f16x8 xh[KS], xl[KS];            // X^T as B fragments of GEMM1 (hi / lo of x' * sx)
    f32x4 xr[KEEPX ? NB : 1];        // KEEPX: the fp32 tile itself stays in registers for the residual
    f32x4 y[NB];                     // Y^T accumulators
    f32x4 hc[2];                     // H^T of the chunk being finished (GELU -> GEMM2)
    f32x4 hn[2];                     // H^T of the next chunk (GEMM1 running)
    float hfac = 0.f;                // inv1 / sx of this lane's pixel (set by load_x): H = acc * hfac + b1
    float hfac_cur = 0.f;            // the factor that belongs to hc (the tile whose chunk is being finished)
#pragma unroll
    for (int n = 0; n < NB; ++n) y[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int n = 0; n < (KEEPX ? NB : 1); ++n) xr[n] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 2; ++b) hc[b] = hn[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < KS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) xh[s][e] = xl[s][e] = (_Float16)0.f;
    float amax = 0.f;                // largest |GELU output| this lane split; X' can only fail by being non-finite
    bool bad = false;

    auto tile_m = [&](int r) { return ((tile_base + r) * ph_unit + sub) * WS_PX + px; };

    // X tile -> (gate) -> per-pixel scale -> split into the B fragments.  Lane (px, g) owns channels 16n + 4g .. + 4 of every
    // block n; the four lanes px, px + 16, px + 32, px + 48 hold one pixel.
    auto load_x = [&](int r) {
        const int m = min(tile_m(r), p.M - 1);      // rows past M re-read the last row and are never stored
        const float* xp = p.x + (size_t)m * p.xld + 4 * g;
        const float* gp = GATED ? p.gate + (size_t)(m / p.HW) * C + 4 * g : nullptr;
        f32x4 v[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) v[n] = *reinterpret_cast<const f32x4*>(xp + 16 * n);
        if (GATED) {
#pragma unroll
            for (int n = 0; n < NB; ++n) v[n] *= *reinterpret_cast<const f32x4*>(gp + 16 * n);
        }
        float mx = 0.f;
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[n][e]));
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        bad = bad || !(mx < INFINITY);               // inf / NaN input: let the fp32 path produce what fp32 produces
        // sx = 2^(13 - floor(log2 mx)): mx * sx in [2^13, 2^14).  Exponent arithmetic on the bits; tiny / zero pixels clamp
        // to 2^100 (everything then lands in the fp16 subnormal range or is 0: absolute error 2^-125, irrelevant)
        const int ex = max((int)((__float_as_uint(mx) >> 23) & 0xffu), 40);
        const float sx = __uint_as_float((unsigned)(267 - ex) << 23);
        hfac = __uint_as_float((unsigned)(ex - 13) << 23) * inv1;      // (1 / sx) * inv1
#pragma unroll
        for (int n = 0; n < NB; ++n) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 a, b;
                ws_split(v[n][e], sx, a, b);
                xh[n >> 1][(n & 1) * 4 + e] = a;
                xl[n >> 1][(n & 1) * 4 + e] = b;
            }
            if constexpr (KEEPX) xr[n] = v[n];
        }
    };

    auto frag = [&](const unsigned char* stage, int f) {
        return *reinterpret_cast<const f16x8*>(stage + f * 1024 + lane * 16);
    };

    // ---- MFMA "units".  A stage is NU units of 4 fragments (4 KB): units 0 .. KS-1 are GEMM1 k-steps
    // [hidden block 0 hi, lo, hidden block 1 hi, lo], units KS .. NU-1 GEMM2 output-block pairs [n hi, lo, n+1 hi, lo].
    // Fragments run through a 3-deep register ring: the reads of unit i+2 are issued before the MFMAs of unit i, so an LDS
    // round trip has two units (12 MFMAs) of cover, and a ring slot is rewritten a whole unit after its last reader (no
    // MFMA-source hazard nops).
    constexpr int NU = KS + NB / 2;
#ifndef RD_WS_RING
#define RD_WS_RING 4          // fragment ring depth of the common steps: reads three units ahead (round 4: +1-2 % over depth 3, -DRD_WS_RING=3; 216 VGPRs)
#endif
    constexpr int RG = RD_WS_RING;
    f16x8 fr[RG > 3 ? RG : 3][4] = {};
    int frag_once = 0;
    auto unit_load = [&](const unsigned char* stage, int i, int slot) {
        if constexpr (ABL & 4) return;
        if constexpr (ABL & 64) {      // MFMAs on fragments that are read once: what the fragment reads cost
            if (frag_once) return;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) fr[slot][j] = frag(stage, 4 * i + j);
    };
    // GEMM1 (transposed): hn += W1c . X^T   (A = weights, B = X fragments in registers)
    auto unit_g1 = [&](int s, int slot) {
        if constexpr (ABL & 6) return;
        hn[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][0], xh[s], hn[0], 0, 0, 0);
        hn[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][2], xh[s], hn[1], 0, 0, 0);
        hn[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][0], xl[s], hn[0], 0, 0, 0);
        hn[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][2], xl[s], hn[1], 0, 0, 0);
        hn[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][1], xh[s], hn[0], 0, 0, 0);
        hn[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][3], xh[s], hn[1], 0, 0, 0);
    };
    // GEMM2 (transposed): Y^T += W2c . H^T
    auto unit_g2 = [&](int n, int slot, const f16x8& hh, const f16x8& hl) {
        if constexpr (ABL & 6) return;
        y[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][0], hh, y[n], 0, 0, 0);
        y[n + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][2], hh, y[n + 1], 0, 0, 0);
        y[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][0], hl, y[n], 0, 0, 0);
        y[n + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][2], hl, y[n + 1], 0, 0, 0);
        y[n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][1], hh, y[n], 0, 0, 0);
        y[n + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[slot][3], hh, y[n + 1], 0, 0, 0);
    };
    // pin the issue order of one pipeline iteration: 4 fragment reads (unit i+2), then the 6 MFMAs of unit i
    auto pin_unit = [&](bool has_load) {
        if (has_load) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);
    };
    // un-scale + bias + GELU in fp32 on the finished chunk, split (x 2^4) into the B fragment of GEMM2: C/D register r of
    // hidden block b is hidden 16b + 4g + r = k-slot 4b + r (W2's hidden columns are permuted to this order on the host)
    auto gelu_split = [&](int q, f16x8& hh, f16x8& hl) {
        if constexpr (ABL & 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) hh[e] = hl[e] = (_Float16)hc[e >> 2][e & 3];
            return;
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&B1s[q * WS_HC + 16 * b + 4 * g]);
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const f32x2 v = rd_gelu2(f32x2{fmaf(hc[b][r], hfac_cur, bv[r]), fmaf(hc[b][r + 1], hfac_cur, bv[r + 1])});
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    _Float16 a, c;
                    ws_split(v[u], WS_SH, a, c);
                    amax = fmaxf(amax, fabsf(v[u]));
                    hh[4 * b + r + u] = a;
                    hl[4 * b + r + u] = c;
                }
            }
        }
    };
    // un-scale + b2 + residual (exact fp32: kept in registers, or re-read from the same float4 addresses as the tile load),
    // store, clear the accumulators
    auto epilogue = [&](int r) {
        const int mm = tile_m(r);
        const int m = min(mm, p.M - 1);
        f32x4 v[NB];
        if constexpr (KEEPX) {
#pragma unroll
            for (int n = 0; n < NB; ++n) v[n] = xr[n];
        } else {
            const float* xp = p.x + (size_t)m * p.xld + 4 * g;
            const float* gp = GATED ? p.gate + (size_t)(m / p.HW) * C + 4 * g : nullptr;
#pragma unroll
            for (int n = 0; n < NB; ++n) v[n] = *reinterpret_cast<const f32x4*>(xp + 16 * n);
            if (GATED) {
#pragma unroll
                for (int n = 0; n < NB; ++n) v[n] *= *reinterpret_cast<const f32x4*>(gp + 16 * n);
            }
        }
        float* yp = p.y + (size_t)m * p.yld + 4 * g;
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&B2s[16 * n + 4 * g]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = fmaf(y[n][e], inv2, bv[e]) + v[n][e];
            // NaN anywhere upstream (X, the gate, a hidden unit) reaches the output through the MFMAs / the residual: the running
            // maxima above are blind to it (v_max_f32 returns the non-NaN operand), this sum is not
            bad = bad || !(fabsf(o[0]) + fabsf(o[1]) + fabsf(o[2]) + fabsf(o[3]) < INFINITY);
            if (mm < p.M) __builtin_nontemporal_store(o, reinterpret_cast<f32x4*>(yp + 16 * n));
            y[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
    };

    if (wave < WS_WAVES / 2) __builtin_amdgcn_s_setprio(1);   // static: the first wavefront of every SIMD wins arbitration
    if constexpr (PF) {
        static_assert(!KEEPX && G::PIECES == 6, "PF: C = 192, residual folded");
        const float s2h = 1.f / inv2;                         // scale(W2) * WS_SH: an exact power of two
        // raw tile -> the X registers (bit patterns): block n = channels 16n + 4g .. + 4 lands in xh[n / 2] (n even) / xl[n / 2] (n odd).
        // The loads, the stores and the waits between them are inline assembly: the compiler's own wait insertion cannot express
        // "the loads have landed, the younger stores need not have" once a store sits under an exec mask (it falls back to
        // vmcnt(0)), and a wait it places for anything drains the weight DMA queue as well.  What is hidden from it only makes the
        // waits it does place (gate loads) more conservative; the counts used here are stated where they are used.
        auto issue_raw = [&](int r) {
            const int m = min(tile_m(r), p.M - 1);
            const float* xp = p.x + (size_t)m * p.xld + 4 * g;
#pragma unroll
            for (int sblk = 0; sblk < KS; ++sblk) {
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xh[sblk]) : "v"(xp), "n"(64 * (2 * sblk)));
                asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(xl[sblk]) : "v"(xp), "n"(64 * (2 * sblk + 1)));
            }
        };
        // the raw tile has landed: at most `newer` younger vector-memory operations are still outstanding (operations complete
        // in issue order).  Ties the twelve registers to the wait so that no reader can be scheduled above it.
        static_assert(KS == 6, "raw_landed lists twelve registers");
        auto raw_landed = [&](auto newer) {
            constexpr int N = decltype(newer)::value;
            if constexpr (N == 6)
                asm volatile("s_waitcnt vmcnt(6)" : "+v"(xh[0]), "+v"(xl[0]), "+v"(xh[1]), "+v"(xl[1]), "+v"(xh[2]), "+v"(xl[2]), "+v"(xh[3]),
                             "+v"(xl[3]), "+v"(xh[4]), "+v"(xl[4]), "+v"(xh[5]), "+v"(xl[5]));
            else if constexpr (N == 12)
                asm volatile("s_waitcnt vmcnt(12)" : "+v"(xh[0]), "+v"(xl[0]), "+v"(xh[1]), "+v"(xl[1]), "+v"(xh[2]), "+v"(xl[2]), "+v"(xh[3]),
                             "+v"(xl[3]), "+v"(xh[4]), "+v"(xl[4]), "+v"(xh[5]), "+v"(xl[5]));
            else if constexpr (N == 18)
                asm volatile("s_waitcnt vmcnt(18)" : "+v"(xh[0]), "+v"(xl[0]), "+v"(xh[1]), "+v"(xl[1]), "+v"(xh[2]), "+v"(xl[2]), "+v"(xh[3]),
                             "+v"(xl[3]), "+v"(xh[4]), "+v"(xl[4]), "+v"(xh[5]), "+v"(xl[5]));
            else
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(xh[0]), "+v"(xl[0]), "+v"(xh[1]), "+v"(xl[1]), "+v"(xh[2]), "+v"(xl[2]), "+v"(xh[3]),
                             "+v"(xl[3]), "+v"(xh[4]), "+v"(xl[4]), "+v"(xh[5]), "+v"(xl[5]));
        };
        auto raw_block = [&](int n) { return __builtin_bit_cast(f32x4, (n & 1) ? xl[n >> 1] : xh[n >> 1]); };
        auto gate_block = [&](int r, int n) {
            const int m = min(tile_m(r), p.M - 1);
            return *reinterpret_cast<const f32x4*>(p.gate + (size_t)(m / p.HW) * C + 4 * g + 16 * n);
        };
        // the residual of tile r: Y^T += x' * s2h
        auto fold_raw = [&](int r) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                f32x4 v = raw_block(n);
                if (GATED) v *= gate_block(r, n);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[n][e] = fmaf(v[e], s2h, y[n][e]);
            }
        };
        // RS (ABL bit 9, round 5): the residual WITHOUT a second read of the tile.  The X registers still hold x' * sx as (hi, lo): hi + lo is
        // exact in fp32 and equals x' * sx to 2^-22 relative (lo = fp16(x' sx - hi) rounds a 13-bit remainder to 11 bits; it is normal for every
        // channel within 2^-3 of the pixel's largest after the per-pixel scale sx, the pixel's smaller channels keep an ABSOLUTE error of
        // 2^-25 / sx), so  Y^T += (hi + lo) * (s2h / sx)  - exact powers of two - puts the residual into the accumulator with an error of
        // <= 2^-22 |x'| per element: up to 4 fp32 ulps per block where the re-read residual had none (bounded in
        // tests/test_gpu_parity.py::test_mixer_residual_from_split_fragments_small_inputs).
        // The counters said what the re-read cost: 370 MB fetched + written per launch against 249 MB algorithmic (1.49x, VERDICT r4
        // weak #9).  It runs right behind the last reader of the fragments (GEMM1 of the tile's last chunk); the NEXT tile is then
        // requested into the freed registers a whole step earlier than before.
        constexpr bool RS = (ABL & 512) != 0;
        auto fold_split = [&]() {
            const float f = hfac * (1.f / inv1) * s2h;          // (1 / sx) * scale(W2) * WS_SH
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const f16x8& h = xh[n >> 1];
                const f16x8& l = xl[n >> 1];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (float)h[(n & 1) * 4 + e] + (float)l[(n & 1) * 4 + e];
                    y[n][e] = fmaf(v, f, y[n][e]);
                }
            }
        };
        // raw tile r in the X registers -> (gate) -> per-pixel scale -> split, in place (load_x without the loads)
        auto split_raw = [&](int r) {
            if (GATED) {
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const f32x4 v = raw_block(n) * gate_block(r, n);
                    if (n & 1) xl[n >> 1] = __builtin_bit_cast(f16x8, v);
                    else xh[n >> 1] = __builtin_bit_cast(f16x8, v);
                }
            }
            float mx = 0.f;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const f32x4 v = raw_block(n);
#pragma unroll
                for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(v[e]));
            }
            mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            // rows past M (the clamped "next" tile behind the last one; a tile whose masked stores did not count in vmcnt may even
            // be split before it has landed) never reach memory and must not raise the flag either
            bad = bad || (!(mx < INFINITY) && tile_m(r) < p.M);
            const int ex = max((int)((__float_as_uint(mx) >> 23) & 0xffu), 40);
            const float sx = __uint_as_float((unsigned)(267 - ex) << 23);
            hfac = __uint_as_float((unsigned)(ex - 13) << 23) * inv1;
#pragma unroll
            for (int sblk = 0; sblk < KS; ++sblk) {           // 8 raw registers -> the same 8 registers as (hi, lo) fragments
                const f32x4 v0 = raw_block(2 * sblk), v1 = raw_block(2 * sblk + 1);
                f16x8 hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    _Float16 a, b;
                    ws_split(v0[e], sx, a, b);
                    hi[e] = a; lo[e] = b;
                    ws_split(v1[e], sx, a, b);
                    hi[4 + e] = a; lo[4 + e] = b;
                }
                xh[sblk] = hi;
                xl[sblk] = lo;
            }
        };
        bool stores_counted = false;      // the last epilogue's stores were issued with lanes enabled (they count in vmcnt for sure)
        // un-scale + b2 (the residual is in the accumulator), store, clear
        auto epilogue_pf = [&](int r) {
            const int mm = tile_m(r);
            float* yp = p.y + (size_t)min(mm, p.M - 1) * p.yld + 4 * g;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(&B2s[16 * n + 4 * g]);
#pragma unroll
                for (int e = 0; e < 4; ++e) y[n][e] = fmaf(y[n][e], inv2, bv[e]);
                bad = bad || !(fabsf(y[n][0]) + fabsf(y[n][1]) + fabsf(y[n][2]) + fabsf(y[n][3]) < INFINITY);
            }
            // twelve 16-byte stores, rows past M masked off by EXEC - always ISSUED, so that the counts of the waits that follow hold
            // (a wavefront whose whole tile lies past M does not count on them: see `stores_counted`)
            static_assert(NB == 12, "twelve stores");
            const unsigned long long ok = __ballot(mm < p.M);
            unsigned long long saved;
            asm volatile(
                "s_and_saveexec_b64 %[sv], %[ok]\n\t"
                "global_store_dwordx4 %[p], %[d0], off nt\n\t"
                "global_store_dwordx4 %[p], %[d1], off offset:64 nt\n\t"
                "global_store_dwordx4 %[p], %[d2], off offset:128 nt\n\t"
                "global_store_dwordx4 %[p], %[d3], off offset:192 nt\n\t"
                "global_store_dwordx4 %[p], %[d4], off offset:256 nt\n\t"
                "global_store_dwordx4 %[p], %[d5], off offset:320 nt\n\t"
                "global_store_dwordx4 %[p], %[d6], off offset:384 nt\n\t"
                "global_store_dwordx4 %[p], %[d7], off offset:448 nt\n\t"
                "global_store_dwordx4 %[p], %[d8], off offset:512 nt\n\t"
                "global_store_dwordx4 %[p], %[d9], off offset:576 nt\n\t"
                "global_store_dwordx4 %[p], %[d10], off offset:640 nt\n\t"
                "global_store_dwordx4 %[p], %[d11], off offset:704 nt\n\t"
                "s_mov_b64 exec, %[sv]"
                : [sv] "=&s"(saved)
                : [ok] "s"(ok), [p] "v"(yp), [d0] "v"(y[0]), [d1] "v"(y[1]), [d2] "v"(y[2]), [d3] "v"(y[3]), [d4] "v"(y[4]), [d5] "v"(y[5]),
                  [d6] "v"(y[6]), [d7] "v"(y[7]), [d8] "v"(y[8]), [d9] "v"(y[9]), [d10] "v"(y[10]), [d11] "v"(y[11])
                : "memory");
            stores_counted = ok != 0ull;
#pragma unroll
            for (int n = 0; n < NB; ++n) y[n] = f32x4{0.f, 0.f, 0.f, 0.f};
        };
        // ---- the schedule of this wavefront, written out tile by tile: ph idle steps, [first X tile], R x (NC - 1 common steps +
        // switch step), idle steps to T_total.  Every step begins with the workgroup's wait + barrier + DMA of the stage two steps
        // ahead.  BETWEEN A REGISTER LOAD AND ITS WAIT THE CODE IS STRAIGHT-LINE (no branch, so no block boundary at which the
        // compiler could place copies of registers whose loads are still in flight; tests/test_isa_resources.py checks the ISA for
        // exactly that): the step that follows a re-fetch knows its allowance at compile time, the switch step requests, splits and
        // starts the "next" tile also behind the last one (clamped rows, results never used).
        int t = 0;
        auto step_begin = [&](auto extra) {
            ws_step_sync_pf<decltype(extra)::value, !(ABL & 16)>();
            issue_stage(t + 2);        // (no "no DMA" ablation here: the wait counts below rely on the pieces)
        };
        // IL (ABL bit 7, round 4; what the engine dispatches): in the COMMON steps the six DMA pieces of stage t + 2 are issued between the
        // MFMA units instead of in a clump behind the barrier.  A piece costs its wavefront 80-125 issue cycles (phase stamps of the DMA
        // GEMM, tools/mb_gemm_trace.py) and both wavefronts of a SIMD paid all six with the matrix pipe idle; between units they issue
        // while the pipe works.  Same pieces, same order relative to the re-fetch loads, so the counted waits are unchanged.
        constexpr bool IL = (ABL & 128) != 0;
        auto step_sync = [&](auto extra) { ws_step_sync_pf<decltype(extra)::value, !(ABL & 16)>(); };
        auto issue_piece = [&](const unsigned char* src, unsigned char* dst, int u) {
            int f = wave + u * WS_WAVES + rot;
            f = f >= G::STAGE_FRAGS ? f - G::STAGE_FRAGS : f;
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + f * 1024),
                                             (__attribute__((address_space(3))) void*)(dst + f * 1024), 16, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        };
        using E0 = std::integral_constant<int, 0>;
        using E12 = std::integral_constant<int, 12>;
        auto stage_of = [&]() { return lds + (t % WS_NSTAGE) * G::STAGE_BYTES; };
        auto gemm1_only = [&](const unsigned char* stage) {
#pragma unroll
            for (int b = 0; b < 2; ++b) hn[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            unit_load(stage, 0, 0);
            unit_load(stage, 1, 1);
#pragma unroll
            for (int i = 0; i < KS; ++i) {
                if (i + 2 < KS) unit_load(stage, i + 2, (i + 2) % 3);
                unit_g1(i, i % 3);
                pin_unit(i + 2 < KS);
            }
        };
        auto roll = [&]() {
#pragma unroll
            for (int b = 0; b < 2; ++b) hc[b] = hn[b];
            hfac_cur = hfac;
        };
        // common step: GELU of the finished chunk, GEMM1 of the next one, GEMM2 of the finished one; REFETCH: behind the last
        // reader of the X fragments (GEMM1) the raw tile is requested again into the same registers
        auto common_step = [&](int r, auto refetch) {
            const unsigned char* stage = stage_of();
            const int q = (t + NC - 1) % NC;
            const unsigned char* il_src = wimg + (size_t)((t + 2 + NC - 1) % NC) * G::STAGE_BYTES + lane * 16;      // stage t + 2 (IL)
            unsigned char* il_dst = lds + ((t + 2) % WS_NSTAGE) * G::STAGE_BYTES;
            f16x8 hh, hl;
#pragma unroll
            for (int b = 0; b < 2; ++b) hn[b] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int u = 0; u < RG - 1; ++u) unit_load(stage, u, u);
            // units [lo, hi) of the step's MFMA pipeline (compile-time bounds: the loop is unrolled, the ring slots are constants)
            auto units = [&](auto lo_c, auto hi_c) {          // (the ring reads ahead inside [lo, hi) only)
                constexpr int LO = decltype(lo_c)::value, HI = decltype(hi_c)::value;
#pragma unroll
                for (int i = LO; i < HI; ++i) {
                    if (i + RG - 1 < HI) unit_load(stage, i + RG - 1, (i + RG - 1) % RG);
                    if (i < KS) unit_g1(i, i % RG);
                    else unit_g2(2 * (i - KS), i % RG, hh, hl);
                    pin_unit(i + RG - 1 < HI);
                    if constexpr (IL) {
                        if (i < G::PIECES) issue_piece(il_src, il_dst, i);
                    }
                    if constexpr (decltype(refetch)::value) {
                        if (i == KS - 1) {
                            __builtin_amdgcn_sched_barrier(0);
                            if constexpr (RS) {
                                fold_split();              // residual from the fragments GEMM1 has just read for the last time ...
                                __builtin_amdgcn_sched_barrier(0);
                                issue_raw(r + 1);          // ... whose registers then receive the NEXT tile
                            } else {
                                issue_raw(r);
                            }
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                }
            };
            using U0 = std::integral_constant<int, 0>;
            using UK = std::integral_constant<int, KS>;
            using UN = std::integral_constant<int, NU>;
            // GO (ABL bit 8, round 4): GEMM1 of the next chunk (which does not need this chunk's GELU) runs BEFORE the GELU, GEMM2 after it.
            // With the GELU first, the two wavefronts of a SIMD - released together by the step's barrier - did their GELUs at the same
            // time with the matrix pipe idle (ablation: GELU = 27 of 153 us).  With the GELU in the middle the wavefront that wins the
            // matrix pipe (static s_setprio) reaches its GELU while the other still issues GEMM1, and starts GEMM2 while the other is in its
            // GELU: one code path, the skew comes from the arbitration.
            if constexpr ((ABL & 256) != 0) {
                units(U0{}, UK{});
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int u = 0; u < RG - 1; ++u) unit_load(stage, KS + u, (KS + u) % RG);      // GEMM2's first fragments land under the GELU
                gelu_split(q, hh, hl);
                __builtin_amdgcn_sched_barrier(0);
                units(UK{}, UN{});
            } else {
                gelu_split(q, hh, hl);
                __builtin_amdgcn_sched_barrier(0);
                units(U0{}, UN{});
            }
            roll();
        };
        bool after_switch = false;
        auto step_begin_rt = [&]() {                                           // allowance known at run time only (no load in flight)
            if (!after_switch) step_begin(E0{});
            else step_begin(E12{});
            after_switch = false;
        };
        // the same in front of a common step: under IL only the wait + barrier - common_step issues the pieces itself.  (NOT for the idle
        // steps: an idle wavefront still owes its six fragments of every stage to the wavefronts that compute.)
        auto step_begin_common_rt = [&]() {
            if constexpr (IL) {
                if (!after_switch) step_sync(E0{});
                else step_sync(E12{});
                after_switch = false;
            } else {
                step_begin_rt();
            }
        };
        for (; t < T_total && (R == 0 || t < ph); ++t) step_begin(E0{});       // idle head (all of it for a wavefront without tiles)
        if (R > 0) {
            step_begin(E0{});                                                  // first tile: nothing to hide its load under
            issue_raw(0);
            raw_landed(E0{});
            split_raw(0);
            gemm1_only(stage_of());
            frag_once = 1;
            roll();
            ++t;
            for (int r = 0; r < R; ++r) {
                for (int j = 1; j < NC - 1; ++j) {
                    step_begin_common_rt();
                    common_step(r, std::false_type{});
                    ++t;
                }
                if constexpr (IL) step_sync(E0{});
                else step_begin(E0{});
                common_step(r, std::true_type{});
                ++t;
                // switch step: residual into the accumulator, next tile requested, last chunk finished, epilogue, next tile split
                step_begin(E12{});                    // the twelve re-fetch loads sit behind the previous step's DMA pieces
                {
                    const unsigned char* stage = stage_of();
                    const int q = (t + NC - 1) % NC;
                    if constexpr (!RS) {
                        raw_landed(std::integral_constant<int, 6>{});             // younger: this step's six DMA pieces
                        fold_raw(r);
                        __builtin_amdgcn_sched_barrier(0);
                        issue_raw(r + 1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    f16x8 hh, hl;
                    unit_load(stage, KS, 0);
                    unit_load(stage, KS + 1, 1);
                    gelu_split(q, hh, hl);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int i = KS; i < NU; ++i) {
                        if (i + 2 < NU) unit_load(stage, i + 2, (i + 2 - KS) % 3);
                        unit_g2(2 * (i - KS), (i - KS) % 3, hh, hl);
                        pin_unit(i + 2 < NU);
                    }
                    epilogue_pf(r);
                    __builtin_amdgcn_sched_barrier(0);
                    if constexpr (RS) raw_landed(std::integral_constant<int, 18>{});     // younger: this step's six DMA pieces + the twelve stores
                    else raw_landed(E12{});           // younger: the twelve stores (a tile wholly past M: see split_raw)
                    split_raw(r + 1);
                    gemm1_only(stage);
                    roll();
                    after_switch = stores_counted;    // the stores are young (the next tile's loads have been waited for); stores that
                                                      // may not have counted get no allowance
                }
                ++t;
            }
        }
        for (; t < T_total; ++t) step_begin_rt();                              // idle tail
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                       // the DMA pieces of the two stages nobody reads
    } else
    for (int t = 0; t < T_total; ++t) {
        if constexpr (ABL & 16) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else ws_step_sync<G::PIECES>(t + 1 < T_total);
        if (t + 2 < T_total && !(ABL & 1)) issue_stage(t + 2);
        const int u = t - ph;
        if (u < 0 || u > R * NC) continue;               // (idle head / tail of this phase; barriers and DMA above still run)
        const unsigned char* stage = lds + (t % WS_NSTAGE) * G::STAGE_BYTES;
        const int q = (t + NC - 1) % NC;                  // weight chunk finished this step
        const bool has_cur = u >= 1;
        const bool tile_end = has_cur && (u % NC == 0);
        if (has_cur && !tile_end) {
            // common step: GELU of this chunk (VALU), then one MFMA pipeline over GEMM1 of the next chunk and GEMM2 of this
            // one.  The two wavefronts of a SIMD are kept out of phase by static priority (s_setprio above): the favoured
            // one wins the matrix pipe, so the other's GELU falls under its MFMAs and vice versa.
            f16x8 hh, hl;
#pragma unroll
            for (int b = 0; b < 2; ++b) hn[b] = f32x4{0.f, 0.f, 0.f, 0.f};
            unit_load(stage, 0, 0);
            unit_load(stage, 1, 1);
            gelu_split(q, hh, hl);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < NU; ++i) {
                if (i + 2 < NU) unit_load(stage, i + 2, (i + 2) % 3);
                if (i < KS) unit_g1(i, i % 3);
                else unit_g2(2 * (i - KS), i % 3, hh, hl);
                pin_unit(i + 2 < NU);
            }
        } else {
            if (has_cur) {                                // last chunk of a tile: GELU, GEMM2, epilogue
                f16x8 hh, hl;
                unit_load(stage, KS, 0);
                unit_load(stage, KS + 1, 1);
                gelu_split(q, hh, hl);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = KS; i < NU; ++i) {
                    if (i + 2 < NU) unit_load(stage, i + 2, (i + 2 - KS) % 3);
                    unit_g2(2 * (i - KS), (i - KS) % 3, hh, hl);
                    pin_unit(i + 2 < NU);
                }
                epilogue(u / NC - 1);
            }
            if (u / NC < R) {                             // tile start (u % NC == 0 here): X tile, GEMM1 of its first chunk
                load_x(u / NC);
#pragma unroll
                for (int b = 0; b < 2; ++b) hn[b] = f32x4{0.f, 0.f, 0.f, 0.f};
                unit_load(stage, 0, 0);
                unit_load(stage, 1, 1);
#pragma unroll
                for (int i = 0; i < KS; ++i) {
                    if (i + 2 < KS) unit_load(stage, i + 2, (i + 2) % 3);
                    unit_g1(i, i % 3);
                    pin_unit(i + 2 < KS);
                }
            }
        }
#pragma unroll
        for (int b = 0; b < 2; ++b) hc[b] = hn[b];
        hfac_cur = hfac;
    }
    if ((bad || !(amax * WS_SH < 65504.f)) && p.range_flag) rd_raise_flag(p.range_flag);   // NaN: `bad` (epilogue), not amax
}

// C = 96 builds and passes the same tests, but measures level with the round-1 kernel (99 vs 97 us at M = 211 200): the
// engine only routes C = 192 here
bool mixer_ws_supported(int C) { return C == 96 || C == 192; }
bool mixer_ws_prefetch() {
    static const bool on = [] { const char* e = getenv("RD_WS_PF"); return e ? e[0] == '1' : true; }();
    return on;
}
bool mixer_ws_preferred(int C) { return C == 192; }

// largest power of two s with max|w| * s < 2^14 (1 for an all-zero matrix)
static float ws_weight_scale(const float* w, size_t n) {
    float mx = 0.f;
    for (size_t i = 0; i < n; ++i) mx = std::fmax(mx, std::fabs(w[i]));
    if (!(mx > 0.f) || !(mx < INFINITY)) return 1.f;
    int e = 0;
    (void)std::frexp(mx, &e);            // mx = f * 2^e, f in [0.5, 1)  ->  mx * 2^(14 - e) in [2^13, 2^14)
    return std::ldexp(1.f, 14 - e);
}

// host: the weight stream image.  w1 [2C][C], w2 [C][2C] fp32 (BN folded).  img: NC stages of STAGE_BYTES;
// inv[0] = 1 / scale(W1), inv[1] = 1 / (scale(W2) * WS_SH) - what the kernel multiplies its accumulators by.
void prepare_mixer_weights_ws(const float* w1, const float* w2, int C, std::vector<uint16_t>& img, float inv[2]) {
    const int KS = C / 32, NB = C / 16, NC = 2 * C / WS_HC, H2 = 2 * C;
    const int w1_frags = 2 * KS * 2, w2_frags = NB * 2, stage_halfs = (w1_frags + w2_frags) * 512;
    img.assign((size_t)NC * stage_halfs, 0);
    const float s1 = ws_weight_scale(w1, (size_t)H2 * C), s2 = ws_weight_scale(w2, (size_t)H2 * C);
    inv[0] = 1.f / s1;
    inv[1] = 1.f / (s2 * WS_SH);
    auto put = [](float v, float sc, uint16_t& hb, uint16_t& lb) {
        const float vs = v * sc;
        const _Float16 h = (_Float16)vs;
        const _Float16 l = (_Float16)(vs - (float)h);
        __builtin_memcpy(&hb, &h, 2);
        __builtin_memcpy(&lb, &l, 2);
    };
    for (int q = 0; q < NC; ++q) {
        uint16_t* st = img.data() + (size_t)q * stage_halfs;
        const int c1 = (q + 1) % NC;   // W1 chunk of this stage
        for (int b = 0; b < 2; ++b)
            for (int s = 0; s < KS; ++s)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 8; ++e) {
                        const int m = l & 15, g = l >> 4;
                        const int hid = 32 * c1 + 16 * b + m, ch = 16 * (2 * s + e / 4) + 4 * g + (e & 3);
                        const size_t f = (size_t)((s * 2 + b) * 2) * 512 + l * 8 + e;   // unit s = [b0 hi, b0 lo, b1 hi, b1 lo]
                        put(w1[(size_t)hid * C + ch], s1, st[f], st[f + 512]);
                    }
        uint16_t* st2 = st + (size_t)w1_frags * 512;
        const int c2 = q;              // W2 chunk of this stage
        for (int n = 0; n < NB; ++n)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int m = l & 15, g = l >> 4;
                    const int out = 16 * n + m, hid = 32 * c2 + 16 * (e / 4) + 4 * g + (e & 3);
                    const size_t f = (size_t)(2 * n) * 512 + l * 8 + e;
                    put(w2[(size_t)out * H2 + hid], s2, st2[f], st2[f + 512]);
                }
    }
}

// smallest number of steps in which `waves` wavefronts with phases (w * ph_mul) mod NC cover n_tiles tiles
static int ws_total_steps(int n_tiles, int waves, int NC, int ph_mul, int tail = 1) {
    for (int T = NC + tail;; ++T) {
        long cap = 0;
        for (int j = 0; j < NC; ++j) {
            const long n = (T - tail - (j * ph_mul) % NC) / NC;
            if (n <= 0) continue;
            cap += n * (waves / NC + (j < waves % NC ? 1 : 0));
        }
        if (cap >= n_tiles) return T;
    }
}

template <int C, bool GATED, bool KEEPX, int ABL = 0, bool PF = false>
static void launch_ws(const MixerParams& p, const unsigned char* wimg, int n_tiles, int grid, int ph_mul, int ph_unit, hipStream_t s) {
    static unsigned long long lds_ok = 0;
    rd_allow_dynamic_lds((const void*)lc_mixer_ws_kernel<C, GATED, KEEPX, ABL, PF>, WsGeom<C>::LDS_BYTES, lds_ok);
    const int T = ws_total_steps((n_tiles + ph_unit - 1) / ph_unit, grid * WS_WAVES / ph_unit, WsGeom<C>::NC, ph_mul, 1);
    hipLaunchKernelGGL((lc_mixer_ws_kernel<C, GATED, KEEPX, ABL, PF>), dim3(grid), dim3(512), WsGeom<C>::LDS_BYTES, s, p, wimg, n_tiles, T,
                       ph_mul, ph_unit, p.ws_inv1, p.ws_inv2);
}

// p.w1h carries the weight stream image, p.ws_inv1 / ws_inv2 its inverse scales (prepare_mixer_weights_ws);
// p.dbg (microbenchmark A/B): bit 0 force lock step, bit 2 force per-wavefront phases, bit 3 force per-workgroup phases, bit 1 flip the residual policy
// (default: kept in registers at C = 96, re-read at C = 192 where keeping it spills); bits 8.. ablations (C = 192, garbage)
void launch_mixer_fused_ws(const MixerParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        (void)hipGetDevice(&dev);
        n_cu = rd_cu_budget((hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256);
    }
    const int n_tiles = (p.M + WS_PX - 1) / WS_PX;
    const int n_wg = (n_tiles + WS_WAVES - 1) / WS_WAVES;
    // persistent grid: as many rounds as a full machine needs, on the FEWEST workgroups that still manage in that many
    // rounds (825 workgroup tiles on 256 CUs are 4 rounds either way; 207 workgroups leave 49 CUs to the kernels of the
    // other streams instead of idling through the fourth round)
    const int rounds = (n_wg + n_cu - 1) / n_cu;
    const int grid = (n_wg + rounds - 1) / rounds;
    const unsigned char* img = reinterpret_cast<const unsigned char*>(p.w1h);
    // Phases.  Lock step (every wavefront of the chip switches tiles at the same step) puts the tile traffic into bursts
    // during which nothing computes (ablation: 40 of 168 us at M = 105 600); per-WAVEFRONT phases spread it, but with a barrier
    // every step a wavefront in its tile switch holds up its whole workgroup (175 vs 169 us); per-WORKGROUP phases spread the
    // traffic over the chip without that coupling.  Phases cost up to NC - 1 extra steps when the tiles divide evenly, so the
    // launcher takes them when the schedule grows by at most 1/8.
    const int NC = 2 * p.C / WS_HC;
    const bool pf = p.ws_pf && p.C == 192;
    const int tail = 1;
    const int t_lock = ws_total_steps(n_tiles, grid * WS_WAVES, NC, 0, tail);
    const int t_wg = ws_total_steps((n_tiles + WS_WAVES - 1) / WS_WAVES, grid, NC, 5, tail);     // 5: coprime to NC = 6 and 12
    const int t_wave = ws_total_steps(n_tiles, grid * WS_WAVES, NC, 5, tail);
    int ph_mul = 0, ph_unit = 1;
    if (t_wave < t_lock) { ph_mul = 5; ph_unit = 1; }
    else if (t_wg * 8 <= t_lock * 9) { ph_mul = 5; ph_unit = WS_WAVES; }
    if (p.dbg & 1) { ph_mul = 0; ph_unit = 1; }
    if (p.dbg & 4) { ph_mul = 5; ph_unit = 1; }
    if (p.dbg & 8) { ph_mul = 5; ph_unit = WS_WAVES; }
    const bool gated = p.gate != nullptr;
    if (p.dbg >> 8) {
        switch (p.dbg >> 8) {
            case 1: launch_ws<192, false, false, 1>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 4: launch_ws<192, false, false, 4>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 8: launch_ws<192, false, false, 8>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 12: launch_ws<192, false, false, 12>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 13: launch_ws<192, false, false, 13>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 16: launch_ws<192, false, false, 16>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 29: launch_ws<192, false, false, 29>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 2: launch_ws<192, false, false, 2>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 17: launch_ws<192, false, false, 17>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 25: launch_ws<192, false, false, 25>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 27: launch_ws<192, false, false, 27>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 0: launch_ws<192, false, false, 0, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;      // 32 +: the PF form
            case 32 + 4: launch_ws<192, false, false, 4, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 8: launch_ws<192, false, false, 8, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 12: launch_ws<192, false, false, 12, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 28: launch_ws<192, false, false, 28, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 64: launch_ws<192, false, false, 64, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 72: launch_ws<192, false, false, 72, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 88: launch_ws<192, false, false, 88, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            case 32 + 16: launch_ws<192, false, false, 16, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); break;
            default: break;
        }
        return;
    }
    if (pf) {
        static const bool il = [] { const char* e = getenv("RD_WS_IL"); return !(e && e[0] == '0'); }();     // A/B switch: RD_WS_IL=0 = the round-3 form
        static const bool go = [] { const char* e = getenv("RD_WS_GO"); return !(e && e[0] == '0'); }();    // A/B switch: RD_WS_GO=0 = GELU first (round 3)
        static const bool rs = [] { const char* e = getenv("RD_WS_RS"); return !(e && e[0] == '0'); }();    // A/B switch: RD_WS_RS=0 = residual re-read (rounds 3-4)
        if (il && go && rs) {
            if (gated) launch_ws<192, true, false, 896, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
            else launch_ws<192, false, false, 896, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
            return;
        }
        if (il && go) {
            if (gated) launch_ws<192, true, false, 384, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
            else launch_ws<192, false, false, 384, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
            return;
        }
        if (il) {
            if (gated) launch_ws<192, true, false, 128, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
            else launch_ws<192, false, false, 128, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
            return;
        }
        if (gated) launch_ws<192, true, false, 0, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
        else launch_ws<192, false, false, 0, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s);
        return;
    }
    const bool keepx = (p.C == 96) != ((p.dbg & 2) != 0);
#define RD_WS(CC)                                                                                                \
    do {                                                                                                         \
        if (gated) { if (keepx) launch_ws<CC, true, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); else launch_ws<CC, true, false>(p, img, n_tiles, grid, ph_mul, ph_unit, s); } \
        else { if (keepx) launch_ws<CC, false, true>(p, img, n_tiles, grid, ph_mul, ph_unit, s); else launch_ws<CC, false, false>(p, img, n_tiles, grid, ph_mul, ph_unit, s); } \
    } while (0)
    if (p.C == 192) RD_WS(192);
    else if (p.C == 96) RD_WS(96);
#undef RD_WS
}

}  // namespace rd
