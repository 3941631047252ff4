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
// Split-fp16 variant (precision "auto" / "h3"): same tiling, split and statistics, but the products run on the fp16 matrix
// cores with (hi, lo) operands - 3 x v_mfma_f32_32x32x16_f16 per 16-wide k-step instead of 8 fp32 MFMAs, fp32 accumulate
// (arithmetic as in kernels_conv_h3.hip).  The token tile is split once per block, the weights once at load time.
static constexpr int CH_LD = CT_K + 8;   // LDS row stride in halfs (272 B): conflict-free ds_read_b128

__global__ void __launch_bounds__(256) ctc_head_h3_kernel(CtcParams p, int cls_per_split) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    _Float16* Xh = reinterpret_cast<_Float16*>(smem);
    _Float16* Xl = Xh + CT_TOK * CH_LD;
    _Float16* Wh = Xl + CT_TOK * CH_LD;         // [128 classes][128] UNPADDED: filled by LDS-DMA, 16-byte chunk c of row r
    _Float16* Wl = Wh + CT_CLS * CT_K;          // sits at chunk c ^ (r & 15) (rows are 256 B = a whole bank sweep apart)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int split = blockIdx.x % p.nsplit, ttile = blockIdx.x / p.nsplit;
    const int tok0 = ttile * CT_TOK;
    const int c_begin = split * cls_per_split;
    const int c_end = min(p.C, c_begin + cls_per_split);
    const int nit = (c_end - c_begin + CT_CLS - 1) / CT_CLS;
    const _Float16* wh_g = reinterpret_cast<const _Float16*>(p.wh);
    const _Float16* wl_g = reinterpret_cast<const _Float16*>(p.wl);

    // ---- X tile (tokens): bias column appended, split once
    float amax = 0.f;
    {
        const int lrow = tid >> 5, lkq = tid & 31;
#pragma unroll 4
        for (int r = lrow; r < CT_TOK; r += 8) {
            const int tok = min(tok0 + r, p.M - 1), k = 4 * lkq;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (k < p.K) v = *reinterpret_cast<const f32x4*>(p.x + (size_t)tok * p.xld + k);
            else if (k == p.K) v[0] = 1.f;
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                _Float16 h, l;
                rd_split(v[e], h, l);
                hi[e] = h;
                lo[e] = l;
                amax = fmaxf(amax, fabsf(v[e]));
            }
            *reinterpret_cast<f16x4*>(&Xh[r * CH_LD + k]) = hi;
            *reinterpret_cast<f16x4*>(&Xl[r * CH_LD + k]) = lo;
        }
    }
    // W tile by LDS-DMA: 64 KB = 64 instructions of 1 KB per iteration, 16 per wavefront (no staging registers, no
    // ds_write).  Instruction q covers bytes [q*1024, +1024) of [hi plane | lo plane]; lane -> (row, chunk position).
    unsigned char* lds_bytes = reinterpret_cast<unsigned char*>(smem);
    const unsigned w_base = (unsigned)((Wh - Xh) * 2);
    int woff[16];            // element offset of this lane's source chunk inside a 128-class block of W
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const int q = wave + 4 * u;                       // 0..63
        const int o = (q & 31) * 1024 + lane * 16;        // byte offset inside the plane
        const int row = o >> 8, cp = (o & 255) >> 4;
        woff[u] = row * CT_K + 8 * (cp ^ (row & 15));
    }
    auto issue_w = [&](int it) {
        const int cb = c_begin + it * CT_CLS;
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int q = wave + 4 * u;
            const int row = woff[u] / CT_K;
            // classes past the end of this split re-read its last class; their logits are masked to -inf below
            const int cls = min(cb + row, c_end - 1);
            const _Float16* src = ((q >> 5) ? wl_g : wh_g) + (size_t)cls * CT_K + (woff[u] - row * CT_K);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                             (__attribute__((address_space(3))) void*)(lds_bytes + w_base + (unsigned)q * 1024u), 16, 0, 0);
        }
    };
    issue_w(0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

    float m_run = -INFINITY, s_run = 0.f;
    int i_run = 0;
    const int xo = (wave * 32 + l31) * CH_LD + 8 * lhi;
    const int wrow = l31 * CT_K, wkey = l31 & 15;
    for (int it = 0; it < nit; ++it) {
        const bool more = it + 1 < nit;
        f32x16 acc1[4], acc2[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[ct][r] = acc2[ct][r] = 0.f;
#pragma unroll 2
        for (int ks = 0; ks < CT_K / 16; ++ks) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(&Xh[xo + ks * 16]);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(&Xl[xo + ks * 16]);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
                const int wa = ct * 32 * CT_K + wrow + (((2 * ks + lhi) ^ wkey) << 3);
                const f16x8 ah = *reinterpret_cast<const f16x8*>(&Wh[wa]);
                const f16x8 al = *reinterpret_cast<const f16x8*>(&Wl[wa]);
                acc1[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[ct], 0, 0, 0);
                acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[ct], 0, 0, 0);
                acc2[ct] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[ct], 0, 0, 0);
            }
        }
        // every wavefront is done with this W tile: request the next one, it lands under the statistics below
        asm volatile("s_barrier" ::: "memory");
        if (more) issue_w(it + 1);
        const int cb = c_begin + it * CT_CLS + 4 * lhi;
        float lm = -INFINITY;
        int li = 0;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = cb + ct * 32 + (r & 3) + 8 * (r >> 2);
                const float v = (c < c_end) ? fmaf(acc2[ct][r], 1.f / 2048.f, acc1[ct][r]) : -INFINITY;
                acc1[ct][r] = v;
                if (v > lm) { lm = v; li = c; }
            }
        const float m_new = fmaxf(m_run, lm);
        if (m_new > -INFINITY) {
            float s = 0.f;
#pragma unroll
            for (int ct = 0; ct < 4; ++ct)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += __expf(acc1[ct][r] - m_new);
            s_run = s_run * __expf(m_run - m_new) + s;
            if (lm > m_run) i_run = li;
            m_run = m_new;
        }
        asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (!(amax < 65504.f) && p.range_flag) atomicOr(p.range_flag, 1u);
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

int ctc_head_nsplit(int M, int C) {
    const int tiles = (M + CT_TOK - 1) / CT_TOK;
    int ns = 8;
    while (tiles * ns < 256 && ns < 64 && (C / (ns * 2)) >= CT_CLS) ns *= 2;
    // small dictionaries: every split must own at least one class (classes per split are rounded up to a multiple of 4)
    while (ns > 1 && (((C + ns - 1) / ns + 3) / 4 * 4) * (ns - 1) >= C) --ns;
    return ns;
}

void launch_ctc_head(const CtcParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    const int tiles = (p.M + CT_TOK - 1) / CT_TOK;
    int cps = (p.C + p.nsplit - 1) / p.nsplit;
    cps = (cps + 3) / 4 * 4;
    const size_t sh = (size_t)(CT_TOK + CT_CLS) * CT_LD * sizeof(float);
    const size_t sh3 = (size_t)(2 * CT_TOK * CH_LD + 2 * CT_CLS * CT_K) * sizeof(_Float16);
    static unsigned long long lds_ok = 0, lds_ok3 = 0;
    rd_allow_dynamic_lds((const void*)ctc_head_kernel, sh, lds_ok);
    rd_allow_dynamic_lds((const void*)ctc_head_h3_kernel, sh3, lds_ok3);
    if (p.wh) hipLaunchKernelGGL(ctc_head_h3_kernel, dim3(tiles * p.nsplit), dim3(256), sh3, s, p, cps);
    else hipLaunchKernelGGL(ctc_head_kernel, dim3(tiles * p.nsplit), dim3(256), sh, s, p, cps);
    hipLaunchKernelGGL(ctc_merge_kernel, dim3((p.M + 255) / 256), dim3(256), 0, s, p.part, p.M, p.nsplit, p.idx, p.prob);
}

}  // namespace rd
