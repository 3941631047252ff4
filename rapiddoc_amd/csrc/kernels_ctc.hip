// Fused CTC head for PP-OCRv6 rec:  Linear(120 -> 18710) + per-time-step argmax + softmax max-probability.
//
// Replaces `torch.softmax(ctc_logits).cpu().numpy()` (reference rapid_doc/model/ocr/torch.py:186-192, 3 MB of
// probabilities per 48x320 crop) followed by rapidocr's CTCLabelDecode argmax/max on the host: the
// [B*T, 18710] logits never leave the chip, only (idx, prob) per time step do.
//
// GEMM is computed transposed on the fp32 matrix cores, D[class][token] = W[class][:] . X[token][:], so a
// lane's 16 accumulator registers of a 32x32 tile are 16 CLASSES of ONE token: the running
// (max, argmax, sum-exp) update is register-local VALU work, one v_permlane/shuffle (xor 32) merges the two
// class halves of a wavefront, and a tiny second kernel merges the class splits.
//   * the bias rides in the padded K column (K=120 -> 128: X[:,120] = 1, W'[:,120] = bias)
//   * class ranges are split across blocks so that split s runs on XCD s%8: each XCD keeps its ~1.2 MB slice
//     of W' resident in its private L2 while the token tiles stream past
#include <cstdlib>
#include <type_traits>

#include "rd_device.h"

namespace rd {


static constexpr int CT_TOK = 128;   // tokens per block (4 waves x 32)
static constexpr int CT_CLS = 128;   // classes per iteration (4 MFMA tiles)
static constexpr int CT_K = 128;     // padded K
static constexpr int CT_LD = 132;    // LDS row stride (floats): conflict-free ds_read_b128

__global__ void __launch_bounds__(256) ctc_head_kernel(CtcParams p, int cls_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;
    float* Ws = smem + CT_TOK * CT_LD;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int split = blockIdx.x % p.nsplit, ttile = blockIdx.x / p.nsplit;
    const int tok0 = ttile * CT_TOK;
    const int c_begin = split * cls_per_split;
    const int c_end = min(p.C, c_begin + cls_per_split);
    const int nit = (c_end - c_begin + CT_CLS - 1) / CT_CLS;
    const int lrow = tid >> 5, lkq = tid & 31;  // 8 rows x 32 float4 per pass
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- X tile (tokens), bias column appended
#pragma unroll 4
    for (int r = lrow; r < CT_TOK; r += 8) {
        const int tok = tok0 + r, k = 4 * lkq;
        f32x4 v = zero4;
        if (tok < p.M) {
            if (k < p.K) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)tok * p.xld + k);
            else if (k == p.K) v[0] = 1.f;
        }
        *reinterpret_cast<f32x4*>(&Xs[r * CT_LD + k]) = v;
    }
    f32x4 wreg[16];
    auto load_w = [&](int it) {
        const int cb = c_begin + it * CT_CLS;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int c = cb + lrow + 8 * i;
            wreg[i] = (c < c_end) ? *reinterpret_cast<const f32x4*>(p.w + (size_t)c * CT_K + 4 * lkq) : zero4;
        }
    };
    auto store_w = [&]() {
#pragma unroll
        for (int i = 0; i < 16; ++i) *reinterpret_cast<f32x4*>(&Ws[(lrow + 8 * i) * CT_LD + 4 * lkq]) = wreg[i];
    };
    load_w(0);
    store_w();
    __syncthreads();

    float m_run = -INFINITY, s_run = 0.f;
    int i_run = 0;
    const float* xp = &Xs[(wave * 32 + l31) * CT_LD + 4 * lhi];
    const float* wp = &Ws[l31 * CT_LD + 4 * lhi];
    for (int it = 0; it < nit; ++it) {
        const bool more = it + 1 < nit;
        if (more) load_w(it + 1);
        f32x16 acc[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ct][r] = 0.f;
#pragma unroll 4
        for (int g = 0; g < CT_K / 8; ++g) {
            const f32x4 bf = *reinterpret_cast<const f32x4*>(xp + g * 8);
            f32x4 af[4];
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) af[ct] = *reinterpret_cast<const f32x4*>(wp + ct * 32 * CT_LD + g * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ct = 0; ct < 4; ++ct)
                    acc[ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[ct][e], bf[e], acc[ct], 0, 0, 0);
        }
        // ---- register-local running (max, argmax, sum-exp) over this lane's 64 classes of its token
        const int cb = c_begin + it * CT_CLS + 4 * lhi;
        float lm = -INFINITY;
        int li = 0;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cb + ct * 32 + (r & 3) + 8 * (r >> 2);
                const float v = (c < c_end) ? acc[ct][r] : -INFINITY;
                acc[ct][r] = v;
                if (v > lm) { lm = v; li = c; }
            }
        const float m_new = fmaxf(m_run, lm);
        if (m_new > -INFINITY) {
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __expf(acc[ct][r] - m_new);
            s_run = s_run * __expf(m_run - m_new) + s;
            if (lm > m_run) i_run = li;
            m_run = m_new;
        }
        __syncthreads();
        if (more) {
            store_w();
            __syncthreads();
        }
    }
    // ---- merge the two class halves of the wavefront (lane l and l^32 hold the same token)
    const float om = __shfl_xor(m_run, 32, 64), os = __shfl_xor(s_run, 32, 64);
    const int oi = __shfl_xor(i_run, 32, 64);
    const float mm = fmaxf(m_run, om);
    // (a half that saw no class keeps m = -inf, s = 0: exp(-inf - -inf) would be NaN)
    const float ss = (m_run > -INFINITY ? s_run * __expf(m_run - mm) : 0.f) + (om > -INFINITY ? os * __expf(om - mm) : 0.f);
    const int ii = (om > m_run || (om == m_run && oi < i_run)) ? oi : i_run;
    const int tok = tok0 + wave * 32 + l31;
    if (lhi == 0 && tok < p.M) {
        float* o = p.part + ((size_t)tok * p.nsplit + split) * 4;
        o[0] = mm;
        o[1] = ss;
        o[2] = __int_as_float(ii);
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Split-fp16 variant (precision "auto" / "h3"): same split and statistics, but the products run on the fp16 matrix cores with
// (hi, lo) operands - 3 x v_mfma_f32_32x32x16_f16 per 16-wide k-step instead of 8 fp32 MFMAs, fp32 accumulate (arithmetic as in
// kernels_conv_h3.hip).  The weights are split once at load time.
//
// Round 2 restructuring.  The round-1 kernel (128 tokens x 128 classes per step, 4 wavefronts, token tile and one class tile
// in LDS) measured 204-270 us per launch against ~30 us of MFMA issue.  Ablations of an intermediate version showed its phases
// simply ADD UP per class tile: MFMA 3070 cycles + statistics ~3800 cycles of VALU (merge, max / argmax, 64 exp per lane) +
// ~1500 cycles of LDS-DMA issue and wait: with one wavefront per SIMD nothing runs under anything else unless a single
// instruction stream interleaves it, and the compiler does not (sched_group_barrier pipelines were ignored, it clusters the
// MFMAs of a k-step and then runs the VALU work with the matrix pipe idle).  So the overlap is left to the hardware:
//   * 8 wavefronts x 32 tokens per workgroup, TWO per SIMD, 64 classes per step; the token tile lives in REGISTERS as the MFMA
//     B operand (8 k-steps x (hi, lo) x f16x8 = 64 VGPRs, split once);
//   * wavefronts 0-3 run [MFMAs of tile s, statistics of tile s], wavefronts 4-7 run [statistics of tile s-1, MFMAs of tile s]
//     between the same two barriers: each SIMD hosts one wavefront of either kind, so its matrix pipe and its VALU are busy
//     with different wavefronts at the same time;
//   * class tiles (32 KB: hi and lo planes of 64 classes) are double-buffered by LDS-DMA, four 1-KB pieces per wavefront and
//     step, requested one full step ahead; one barrier per step; a 256-token workgroup also halves the L2 -> LDS weight stream;
//   * statistics in slices of 8 classes with an online softmax (one extra exp per slice), exp as one fma + v_exp, class masking
//     only in a split's last tile, the running argmax as selects of uniform class codes (no divergent branch);
//   * blocks are dealt to XCDs in contiguous runs of the (class split, token tile) list, so an XCD's L2 holds the one or two
//     class slices it works on whatever the split count is, and
//   * the split count is chosen to minimise rounds x steps (ctc_head_nsplit): 34 token tiles x 8 splits = 272 blocks used to take
//     two rounds on 256 CUs with the second one 6 % full.
static constexpr int CH_TOK = 256;   // tokens per block (8 wavefronts x 32)
static constexpr int CH_CLS = 64;    // classes per step (2 MFMA tiles)

template <int MODE>   // developer variants (RD_CTC_MODE): 1 pair by wave parity, 2 no skew, 3 s_setprio around the MFMAs
__global__ void __launch_bounds__(512) ctc_head_h3_kernel(CtcParams p, int cls_per_split, int tiles, int per_xcd) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Wbuf = reinterpret_cast<_Float16*>(smem);   // 2 x [hi plane 64 x 128 | lo plane]: filled by LDS-DMA, 16-byte chunk c of
                                                          // row r sits at chunk c ^ (r & 15) (rows are 256 B = a whole bank sweep apart)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool late = MODE == 2 ? false : MODE == 1 ? (wave & 1) != 0 : wave >= 4;   // statistics one tile late (see above)
    const int l31 = lane & 31, lhi = lane >> 5;
    // workgroup b runs on XCD b % 8: XCD x takes entries [x * per_xcd, (x + 1) * per_xcd) of the split-major block list
    const int lin = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if ((blockIdx.x >> 3) >= per_xcd || lin >= tiles * p.nsplit) return;
    const int split = lin / tiles, ttile = lin - split * tiles;
    const int tok0 = ttile * CH_TOK;
    const int c_begin = split * cls_per_split;
    const int c_end = min(p.C, c_begin + cls_per_split);
    const int nit = (c_end - c_begin + CH_CLS - 1) / CH_CLS;
    const _Float16* wh_g = reinterpret_cast<const _Float16*>(p.wh);
    const _Float16* wl_g = reinterpret_cast<const _Float16*>(p.wl);

    // W tile by LDS-DMA: 32 KB = 32 instructions of 1 KB, 4 per wavefront (no staging registers, no ds_write).  Instruction
    // q = wave + 8 u covers bytes [q*1024, +1024) of [hi plane | lo plane] (u >= 2: lo plane): lane -> row rbase + 32 (u & 1),
    // 16-byte chunk position lane & 15, which holds source chunk (lane & 15) ^ (row & 15) = a constant.
    unsigned char* lds_bytes = reinterpret_cast<unsigned char*>(smem);
    const int rbase = 4 * wave + (lane >> 4);
    const int koff = 8 * ((lane & 15) ^ (rbase & 15));
    auto dma_tile = [&](int tile) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            // classes past the end of this split re-read its last class; their logits are masked to -inf in the statistics
            const int cls = min(c_begin + tile * CH_CLS + rbase + 32 * (u & 1), c_end - 1);
            const unsigned off = (unsigned)cls * CT_K + (unsigned)koff;
            const _Float16* src = ((u >= 2) ? wl_g : wh_g) + off;
            const unsigned dst = (unsigned)(tile & 1) * (unsigned)(2 * CH_CLS * CT_K * 2) + (unsigned)(wave + 8 * u) * 1024u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds_bytes + dst), 16, 0, 0);
        }
    };
    dma_tile(0);

    // ---- this wavefront's 32 tokens as B operands: lane (l31, lhi) holds X[token l31][16 ks + 8 lhi .. + 8]; bias column appended
    float amax = 0.f;
    f16x8 xh[CT_K / 16], xl[CT_K / 16];
    {
        const int tok = min(tok0 + wave * 32 + l31, p.M - 1);
        const float* xr = p.x + (size_t)tok * p.xld;
#pragma unroll
        for (int ks = 0; ks < CT_K / 16; ++ks)
#pragma unroll
            for (int hq = 0; hq < 2; ++hq) {
                const int k = ks * 16 + 8 * lhi + 4 * hq;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (k < p.K) v = *reinterpret_cast<const f32x4*>(xr + k);
                else if (k == p.K) v[0] = 1.f;
                f16x4 h4, l4;
                rd_split4(v, h4, l4);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[ks][4 * hq + e] = h4[e];
                    xl[ks][4 * hq + e] = l4[e];
                    amax = (v[e] != v[e]) ? INFINITY : fmaxf(amax, fabsf(v[e]));   // fmaxf alone would drop a NaN token
                }
            }
    }

    float m_run = -INFINITY, s_run = 0.f;
    int i_run = 0;                                   // class code: class - c_begin - 4 lhi
    const int wrow = l31 * CT_K, wkey = l31 & 15;
    constexpr float LOG2E = 1.4426950408889634f;
    f32x16 zero16;
#pragma unroll
    for (int r = 0; r < 16; ++r) zero16[r] = 0.f;
    float vals[32];                                  // merged logits of one tile: this lane's 32 classes of its token

    auto mfma_tile = [&](int s) {
        const _Float16* Wh = Wbuf + (size_t)(s & 1) * (2 * CH_CLS * CT_K);
        const _Float16* Wl = Wh + CH_CLS * CT_K;
        f32x16 acc1[2], acc2[2];
#pragma unroll
        for (int ks = 0; ks < CT_K / 16; ++ks)
#pragma unroll
            for (int ct = 0; ct < 2; ++ct) {
                const int wa = ct * 32 * CT_K + wrow + (((2 * ks + lhi) ^ wkey) << 3);
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&Wh[wa]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&Wl[wa]);
                // (the first k-step starts from a constant zero C operand instead of zero-filled accumulators)
                acc1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xh[ks], ks ? acc1[ct] : zero16, 0, 0, 0);
                acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, xl[ks], ks ? acc2[ct] : zero16, 0, 0, 0);
                acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, xh[ks], acc2[ct], 0, 0, 0);
            }
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) vals[ct * 16 + r] = fmaf(acc2[ct][r], 1.f / 2048.f, acc1[ct][r]);
    };
    // running (max, argmax, sum-exp) over `vals` = tile t; TAIL: the tile may be partial (a split's last tile)
    auto stats = [&](auto tail_t, int t) {
        constexpr bool TAIL = decltype(tail_t)::value;
        const int cb = c_begin + t * CH_CLS + 4 * lhi;
#pragma unroll
        for (int sl = 0; sl < 4; ++sl) {
            // slice sl: this lane's classes cb + ct * 32 + {0..3, 8..11} (+16 for the odd slice)
            const int ct = sl >> 1, r0 = (sl & 1) * 8;
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                v[j] = vals[ct * 16 + r0 + j];
                if (TAIL && cb + ct * 32 + ((r0 + j) & 3) + 8 * ((r0 + j) >> 2) >= c_end) v[j] = -INFINITY;
            }
            const float lm = fmaxf(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])), fmaxf(fmaxf(v[4], v[5]), fmaxf(v[6], v[7])));
            // argmax: lowest class of the slice that attains its maximum, taken if that maximum beats the running one (what a
            // strict `>` scan in class order finds).  Eight selects of UNIFORM class codes straight into the running index (a
            // select chain feeding one final select is turned into a divergent branch by the compiler)
            const bool beats = lm > m_run;
#pragma unroll
            for (int j = 7; j >= 0; --j)
                i_run = (beats && v[j] == lm) ? t * CH_CLS + ct * 32 + ((r0 + j) & 3) + 8 * ((r0 + j) >> 2) : i_run;
            const float m_new = fmaxf(m_run, lm);
            // (TAIL: a masked slice in front of any real class has m_new = -inf, exp(-inf - -inf) would be NaN)
            const float nm = (TAIL && !(m_new > -INFINITY)) ? 0.f : -m_new * LOG2E;
            float sum = 0.f;
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += __builtin_amdgcn_exp2f(fmaf(v[j], LOG2E, nm));
            s_run = fmaf(s_run, __builtin_amdgcn_exp2f(fmaf(m_run, LOG2E, nm)), sum);
            m_run = m_new;
        }
    };
    using T = std::true_type;
    using F = std::false_type;
    for (int s = 0; s < nit; ++s) {
        // class tile s has landed (every wavefront's pieces) and every wavefront is past the MFMAs of step s-1, which read the
        // buffer that tile s+1 goes to
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
        if (s + 1 < nit) dma_tile(s + 1);
        if (late && s > 0) stats(F{}, s - 1);
        if (MODE == 3) __builtin_amdgcn_s_setprio(1);
        mfma_tile(s);
        if (MODE == 3) __builtin_amdgcn_s_setprio(0);
        if (!late) {
            if (s + 1 < nit) stats(F{}, s);
            else stats(T{}, s);
        }
    }
    if (late) stats(T{}, nit - 1);

    if (!(amax < 65504.f) && p.range_flag) rd_raise_flag(p.range_flag);
    i_run += c_begin + 4 * lhi;        // class code -> class
    const float om = __shfl_xor(m_run, 32, 64), os = __shfl_xor(s_run, 32, 64);
    const int oi = __shfl_xor(i_run, 32, 64);
    const float mm = fmaxf(m_run, om);
    // (a half that saw no class keeps m = -inf, s = 0: exp(-inf - -inf) would be NaN)
    const float ss = (m_run > -INFINITY ? s_run * __expf(m_run - mm) : 0.f) + (om > -INFINITY ? os * __expf(om - mm) : 0.f);
    const int ii = (om > m_run || (om == m_run && oi < i_run)) ? oi : i_run;
    const int tok = tok0 + wave * 32 + l31;
    if (lhi == 0 && tok < p.M) {
        float* o = p.part + ((size_t)tok * p.nsplit + split) * 4;
        o[0] = mm;
        o[1] = ss;
        o[2] = __int_as_float(ii);
    }
}

__global__ void __launch_bounds__(256) ctc_merge_kernel(const float* part, int M, int nsplit, int32_t* idx, float* prob) {
    const int tok = blockIdx.x * 256 + threadIdx.x;
    if (tok >= M) return;
    const float* pr = part + (size_t)tok * nsplit * 4;
    float m = -INFINITY;
    int best = 0;
    for (int s = 0; s < nsplit; ++s) {
        const float v = pr[s * 4];
        if (v > m) { m = v; best = __float_as_int(pr[s * 4 + 2]); }
    }
    float sum = 0.f;
    for (int s = 0; s < nsplit; ++s)
        if (pr[s * 4] > -INFINITY) sum += pr[s * 4 + 1] * __expf(pr[s * 4] - m);   // an empty class split carries (-inf, 0)
    idx[tok] = best;
    prob[tok] = 1.f / sum;
}

// class splits: one block = (split, token tile).  The split count is a function of the DICTIONARY only (about 1280 classes = 20 / 10
// steps per split: 15 splits for PP-OCRv6's 18710 classes), never of the token count M: a token's sum of exponentials is formed split by
// split and merged in split order (ctc_merge_kernel), so a split count that followed M (rounds 1-5 minimised rounds x steps per launch)
// made the last bits of a line's probabilities - and with them its mean confidence - depend on how many lines shared its launch.  The
// price is one partly filled last round of ~23 steps per launch (a launch of the bench runs 9-16 rounds).  Every split owns at least
// one class (classes per split are rounded up to a multiple of 4).
static int ctc_pick_nsplit(int C) {
    int ns = (C + 1279) / 1280;
    if (ns > 64) ns = 64;
    while (ns > 1 && (long)(((C + ns - 1) / ns + 3) / 4 * 4) * (ns - 1) >= C) --ns;
    return ns < 1 ? 1 : ns;
}
int ctc_head_nsplit(int M, int C, bool split_fp16) {
    (void)M; (void)split_fp16;
    return ctc_pick_nsplit(C);
}

void launch_ctc_head(const CtcParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    int cps = (p.C + p.nsplit - 1) / p.nsplit;
    cps = (cps + 3) / 4 * 4;
    static unsigned long long lds_ok = 0, lds_ok3 = 0;
    if (p.wh) {
        const int tiles = (p.M + CH_TOK - 1) / CH_TOK;
        const size_t sh3 = (size_t)(2 * 2 * CH_CLS * CT_K) * sizeof(_Float16);   // two (hi, lo) class tiles
        const int per_xcd = (tiles * p.nsplit + 7) / 8;
        static const int mode = std::getenv("RD_CTC_MODE") ? atoi(std::getenv("RD_CTC_MODE")) : 0;
        static unsigned long long okm[4] = {};
#define RD_CTC_CASE(A) case A: rd_allow_dynamic_lds((const void*)ctc_head_h3_kernel<A>, sh3, okm[A]); \
        hipLaunchKernelGGL(ctc_head_h3_kernel<A>, dim3(per_xcd * 8), dim3(512), sh3, s, p, cps, tiles, per_xcd); break;
        switch (mode) { RD_CTC_CASE(1) RD_CTC_CASE(2) RD_CTC_CASE(3) default: RD_CTC_CASE(0) }
#undef RD_CTC_CASE
    } else {
        const int tiles = (p.M + CT_TOK - 1) / CT_TOK;
        const size_t sh = (size_t)(CT_TOK + CT_CLS) * CT_LD * sizeof(float);
        rd_allow_dynamic_lds((const void*)ctc_head_kernel, sh, lds_ok);
        hipLaunchKernelGGL(ctc_head_kernel, dim3(tiles * p.nsplit), dim3(256), sh, s, p, cps);
    }
    hipLaunchKernelGGL(ctc_merge_kernel, dim3((p.M + 255) / 256), dim3(256), 0, s, p.part, p.M, p.nsplit, p.idx, p.prob);
}

}  // namespace rd
