// PP-FormulaNet_plus decoder head on the GPU: enc_to_dec_proj + 6-layer MBart decoder + lm_head, greedy decode with a
// KV cache, batched over all formulas of a page batch.
//
// Reference (rapid_doc/model/formula/rapid_formula_self/networks/heads/):
//   rec_ppformulanet_head.py:1054-1176 generate_export, :919-962 generate_single_iter, :407-630 CustomMBartDecoder
//   rec_unimernet_head.py:502-628 MBartAttention, :635-746 MBartDecoderLayer, :440-456 positional offset 2,
//   :1545-1572 ForcedEOSTokenLogitsProcessor (max_length 1537)
// Differences in HOW (not WHAT): the reference re-applies enc_to_dec_proj to all encoder tokens and re-derives the
// cross-attention K/V every step (:934-936); here they are computed once per batch.  q scaling (head_dim^-0.5) and the
// embedding scale sqrt(d_model) are folded into the weights at load time.  The step index lives in device memory so
// every launch of a step has constant arguments (hipGraph-ready).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "engine.h"
#include "rd_device.h"

namespace rd {

static constexpr int D = 512, HEADS = 16, HD = 32, FFN = 2048;
static constexpr int EOS_ID = 2, PAD_ID = 1, START_ID = 0, FORCED_EOS_LEN = 1537;

struct DecState {  // device-resident scalars
    int step;          // number of tokens generated so far (position of the token being consumed)
    int n_unfinished;
    int arrived;       // select-kernel arrival ticket
};

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// x[b] = LayerNorm(emb[ids[b][t]] + pos[t + 2])   (emb pre-multiplied by sqrt(d_model)); one wavefront per sequence
__global__ void __launch_bounds__(64) dec_embed_ln_kernel(const float* emb, const float* pos, const long long* ids, int ids_ld,
                                                          const DecState* st, const float* g, const float* b, float* x) {
    const int bi = blockIdx.x, lane = threadIdx.x, t = st->step;
    const long long tok = ids[(size_t)bi * ids_ld + t];
    float v[8];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        v[i] = emb[(size_t)tok * D + c] + pos[(size_t)(t + 2) * D + c];
        s += v[i];
    }
    const float mean = wsum(s) / D;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) q += (v[i] - mean) * (v[i] - mean);
    const float rstd = rsqrtf(wsum(q) / D + 1e-5f);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int c = lane + 64 * i;
        x[(size_t)bi * D + c] = (v[i] - mean) * rstd * g[c] + b[c];
    }
}

// Single-query attention for one (sequence, head): keys/values from a cache [B][T][ld] (+ optionally the current step's
// k, v which are also appended to the cache), softmax(q.k) v.  q is already scaled.
struct AttnDecParams {
    const float* q; int ldq;          // [B][ldq], head h at q + h*HD
    const float* kc; const float* vc; int ldkv; long long seq_stride;   // cache rows
    const float* kcur; const float* vcur; int ldcur;                   // current k, v (self-attention) or nullptr
    float* kw; float* vw;             // where to append (self-attention) or nullptr
    const DecState* st; int fixed_T;  // T = fixed_T (cross) or st->step (self, then + current)
    float* out; int ldo;
};
// Round 3: the loop runs ~50 dependent launches of a few microseconds each, so this kernel is a latency chain.  Every row of K is
// read with eight 16-byte loads requested before the first is used (it was 32 scalar loads per key), V by (16 key groups x 8 dim
// quads) with four rows in flight per thread, the block reductions are wavefront shuffles + one LDS exchange (they were 7-step LDS
// trees with a barrier per step).  11 -> ~7 us per launch (12 launches per token).
__global__ void __launch_bounds__(128) dec_attention_kernel(AttnDecParams p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ float sm[];  // scores[Tmax] + red[128]
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int Tc = p.kcur ? p.st->step : p.fixed_T;   // cached keys
    const int T = Tc + (p.kcur ? 1 : 0);
    float* sc = sm;
    float* red = sm + ((T + 3) & ~3);
    static_assert(HD == 32, "eight float4 per head row");
    f4 q[8];
    const f4* qp = reinterpret_cast<const f4*>(p.q + (size_t)b * p.ldq + h * HD);
#pragma unroll
    for (int d = 0; d < 8; ++d) q[d] = qp[d];
    if (p.kcur && tid < HD) {  // append this step's k, v for head h
        p.kw[(size_t)b * p.seq_stride + (size_t)Tc * p.ldkv + h * HD + tid] = p.kcur[(size_t)b * p.ldcur + h * HD + tid];
        p.vw[(size_t)b * p.seq_stride + (size_t)Tc * p.ldkv + h * HD + tid] = p.vcur[(size_t)b * p.ldcur + h * HD + tid];
    }
    auto krow = [&](int j) { return (j < Tc) ? p.kc + (size_t)b * p.seq_stride + (size_t)j * p.ldkv + h * HD : p.kcur + (size_t)b * p.ldcur + h * HD; };
    auto vrow = [&](int j) { return (j < Tc) ? p.vc + (size_t)b * p.seq_stride + (size_t)j * p.ldkv + h * HD : p.vcur + (size_t)b * p.ldcur + h * HD; };
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 128) {
        const f4* kr = reinterpret_cast<const f4*>(krow(j));
        f4 k[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) k[d] = kr[d];
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(q[d][e], k[d][e], s);       // (same order as the scalar loop: d * 4 + e ascending)
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
    // block max: butterfly inside the wavefront, one exchange between the two
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    float sum = 0.f;
    for (int j = tid; j < T; j += 128) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    __syncthreads();                       // (red[0 / 1] read by everybody; sc[] complete)
    if (lane == 0) red[2 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[2] + red[3]);
    // output: thread (g = tid / 8: one of 16 key groups, dq = tid % 8: dims 4 dq .. + 3) sums its keys, four rows in flight
    const int dq = tid & 7, g = tid >> 3;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    int j = g;
    for (; j + 48 < T; j += 64) {
        const f4 v0 = reinterpret_cast<const f4*>(vrow(j))[dq], v1 = reinterpret_cast<const f4*>(vrow(j + 16))[dq];
        const f4 v2 = reinterpret_cast<const f4*>(vrow(j + 32))[dq], v3 = reinterpret_cast<const f4*>(vrow(j + 48))[dq];
        acc += v0 * sc[j];
        acc += v1 * sc[j + 16];
        acc += v2 * sc[j + 32];
        acc += v3 * sc[j + 48];
    }
    for (; j < T; j += 16) acc += reinterpret_cast<const f4*>(vrow(j))[dq] * sc[j];
    __syncthreads();                       // red[] is reused below
    // sum over the 16 key groups: groups g and g ^ 1 .. sit 8 lanes apart inside a wavefront (8 groups per wavefront)
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (lane < 8) *reinterpret_cast<f4*>(&red[wave * 32 + 4 * lane]) = acc;
    __syncthreads();
    if (tid < 32) p.out[(size_t)b * p.ldo + h * HD + tid] = (red[tid] + red[32 + tid]) * inv;
}

// Round 4: the projection that feeds an attention call, fused into it.  One decode step was 52 dependent launches of 5-15 us (the loop
// is a latency chain, profiles/r3_formula_decode.txt); q|k|v (self) and q (cross) were skinny GEMMs of ~9 us each whose only consumer is
// the attention launch behind them.  Here the (sequence, head) workgroup computes ITS 32 (x 3) outputs itself: LayerNorm of the row
// (two-pass, as layernorm_kernel), then a 512-long dot product per output - eight lanes per output, each reading every eighth float4 of
// the weight row so that a load instruction covers 128 contiguous bytes per row - and goes straight on to the scores.  The weight slab
// of a head (64 KB per matrix) is read by the B workgroups of that head: from HBM once, from L2 after that.  52 -> 40 launches per
// token.  Used for B <= 32 (beyond that a real GEMM reads the weights once instead of B times).
struct AttnFusedParams {
    const float* x;                           // [B][D] the layer's input row (before its LayerNorm)
    const float* ln_g; const float* ln_b;
    const float* w; const float* bias;        // SELF: q | k | v rows [3 D][D] (q pre-scaled); cross: q rows [D][D]
    const float* kc; const float* vc; int ldkv; long long seq_stride;
    float* kw; float* vw;                     // SELF: where this step's k, v are appended
    const DecState* st; int fixed_T;
    float* out; int ldo;
};
template <bool SELF>
__global__ void __launch_bounds__(256) dec_attn_fused_kernel(AttnFusedParams p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float* xn = smf;                   // [512] normalised row
    float* qs = smf + D;               // [32] q | [32] k | [32] v of this step
    float* red = qs + 3 * HD;          // [256]
    float* sc = red + 256;             // [T] scores
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(D == 512 && HD == 32, "two values per thread, eight float4 per head row");
    // ---- LayerNorm of row b
    {
        const float v0 = p.x[(size_t)b * D + tid], v1 = p.x[(size_t)b * D + 256 + tid];
        float s = wsum(v0 + v1);
        if (lane == 0) red[wave] = s;
        __syncthreads();
        const float mean = (red[0] + red[1] + red[2] + red[3]) / D;
        const float d0 = v0 - mean, d1 = v1 - mean;
        float q = wsum(d0 * d0 + d1 * d1);
        if (lane == 0) red[4 + wave] = q;
        __syncthreads();
        const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / D + 1e-5f);
        xn[tid] = d0 * rstd * p.ln_g[tid] + p.ln_b[tid];
        xn[256 + tid] = d1 * rstd * p.ln_g[256 + tid] + p.ln_b[256 + tid];
    }
    __syncthreads();
    // ---- this head's projections: output o = tid / 8, its eight lanes take float4 number 8 i + part of the row
    const int Tc = SELF ? p.st->step : p.fixed_T;
    {
        const int o = tid >> 3, part = tid & 7;
        f4 xr[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) xr[i] = reinterpret_cast<const f4*>(xn)[8 * i + part];
#pragma unroll
        for (int m = 0; m < (SELF ? 3 : 1); ++m) {
            const int row = m * D + h * HD + o;
            const f4* wr = reinterpret_cast<const f4*>(p.w + (size_t)row * D);
            f4 wv[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) wv[i] = wr[8 * i + part];
            float acc = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = fmaf(xr[i][e], wv[i][e], acc);
            acc += __shfl_xor(acc, 1, 64);
            acc += __shfl_xor(acc, 2, 64);
            acc += __shfl_xor(acc, 4, 64);
            if (part == 0) {
                const float v = acc + p.bias[row];
                qs[m * HD + o] = v;
                if (SELF && m == 1) p.kw[(size_t)b * p.seq_stride + (size_t)Tc * p.ldkv + h * HD + o] = v;
                if (SELF && m == 2) p.vw[(size_t)b * p.seq_stride + (size_t)Tc * p.ldkv + h * HD + o] = v;
            }
        }
    }
    __syncthreads();
    // ---- single-query attention (dec_attention_kernel with 256 threads; this step's k, v come from LDS)
    const int T = Tc + (SELF ? 1 : 0);
    f4 q[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) q[d] = reinterpret_cast<const f4*>(qs)[d];
    auto krow = [&](int j) -> const float* { return (!SELF || j < Tc) ? p.kc + (size_t)b * p.seq_stride + (size_t)j * p.ldkv + h * HD : qs + HD; };
    auto vrow = [&](int j) -> const float* { return (!SELF || j < Tc) ? p.vc + (size_t)b * p.seq_stride + (size_t)j * p.ldkv + h * HD : qs + 2 * HD; };
    float mx = -INFINITY;
    for (int j = tid; j < T; j += 256) {
        const f4* kr = reinterpret_cast<const f4*>(krow(j));
        f4 k[8];
#pragma unroll
        for (int d = 0; d < 8; ++d) k[d] = kr[d];
        float s = 0.f;
#pragma unroll
        for (int d = 0; d < 8; ++d)
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(q[d][e], k[d][e], s);
        sc[j] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
    if (lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    for (int j = tid; j < T; j += 256) {
        const float e = __expf(sc[j] - mx);
        sc[j] = e;
        sum += e;
    }
    sum = wsum(sum);
    __syncthreads();                       // (red[0..3] read by everybody; sc[] complete)
    if (lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    // output: thread (g = tid / 8: one of 32 key groups, dq = tid % 8: dims 4 dq .. + 3), four rows in flight
    const int dq = tid & 7, g = tid >> 3;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    int j = g;
    for (; j + 96 < T; j += 128) {
        const f4 v0 = reinterpret_cast<const f4*>(vrow(j))[dq], v1 = reinterpret_cast<const f4*>(vrow(j + 32))[dq];
        const f4 v2 = reinterpret_cast<const f4*>(vrow(j + 64))[dq], v3 = reinterpret_cast<const f4*>(vrow(j + 96))[dq];
        acc += v0 * sc[j];
        acc += v1 * sc[j + 32];
        acc += v2 * sc[j + 64];
        acc += v3 * sc[j + 96];
    }
    for (; j < T; j += 32) acc += reinterpret_cast<const f4*>(vrow(j))[dq] * sc[j];
    __syncthreads();                       // red[] is reused below
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o, 64);
    }
    if (lane < 8) *reinterpret_cast<f4*>(&red[wave * 32 + 4 * lane]) = acc;
    __syncthreads();
    if (tid < 32) p.out[(size_t)b * p.ldo + h * HD + tid] = (red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]) * inv;
}

// Round 6: dec_attn_fused_kernel with its latency chain folded.  The round-4 kernel ran LayerNorm -> barrier -> (q, then k, then v: three
// serial rounds of sixteen 16-byte weight loads per thread) -> barrier -> K rows -> scores: 9.6 us (self) / 7.0 us (cross) per launch, twelve
// launches per token.  Nothing in the weight rows or in the cached K rows depends on x, so here
//   * every thread requests its weight row FIRST (self: 768 threads, one of q | k | v each, so the three projections are one round), and
//     the thread's first cached K row right behind it; LayerNorm of the row runs under those loads;
//   * the attention itself is the first 256 threads, unchanged.
// Per output the same eight lanes read the same float4s and sum in the same order, LayerNorm and the softmax are the same code on the same
// 256 threads: results are bit-identical to dec_attn_fused_kernel (RD_DEC_ATTN2=0 selects it for A/B).
template <bool SELF>
__global__ void __launch_bounds__(SELF ? 768 : 256) dec_attn_fused2_kernel(AttnFusedParams p) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    extern __shared__ __attribute__((aligned(16))) float smf[];
    float* xn = smf;                   // [512] normalised row
    float* qs = smf + D;               // [32] q | [32] k | [32] v of this step
    float* red = qs + 3 * HD;          // [256]
    float* sc = red + 256;             // [T] scores
    const int b = blockIdx.x, h = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    static_assert(D == 512 && HD == 32, "two values per thread, eight float4 per head row");
    const bool att = !SELF || tid < 256;                  // the 256 threads of LayerNorm and of the attention
    const int t256 = SELF ? (tid & 255) : tid, pm = SELF ? (tid >> 8) : 0;       // projection: matrix pm, output o, part
    const int o = t256 >> 3, part = t256 & 7;
    // ---- requested first: this thread's weight row (its eighth of it), its first cached K row
    const int row = pm * D + h * HD + o;
    f4 wv[16];
    {
        const f4* wr = reinterpret_cast<const f4*>(p.w + (size_t)row * D);
#pragma unroll
        for (int i = 0; i < 16; ++i) wv[i] = wr[8 * i + part];
    }
    const int Tc = SELF ? p.st->step : p.fixed_T;
    const bool kpre = att && tid < Tc;
    f4 k0[8];
    {
        const f4* kr = reinterpret_cast<const f4*>(p.kc + (size_t)b * p.seq_stride + (size_t)(kpre ? tid : 0) * p.ldkv + h * HD);
#pragma unroll
        for (int d = 0; d < 8; ++d) k0[d] = kr[d];       // (row 0 of the cache buffer exists for every step; unused when !kpre)
    }
    // ---- LayerNorm of row b (threads 0 .. 255)
    {
        float v0 = 0.f, v1 = 0.f;
        if (att) { v0 = p.x[(size_t)b * D + tid]; v1 = p.x[(size_t)b * D + 256 + tid]; }
        float s = wsum(v0 + v1);
        if (att && lane == 0) red[wave] = s;
        __syncthreads();
        const float mean = (red[0] + red[1] + red[2] + red[3]) / D;
        const float d0 = v0 - mean, d1 = v1 - mean;
        float q = wsum(d0 * d0 + d1 * d1);
        if (att && lane == 0) red[4 + wave] = q;
        __syncthreads();
        const float rstd = rsqrtf((red[4] + red[5] + red[6] + red[7]) / D + 1e-5f);
        if (att) {
            xn[tid] = d0 * rstd * p.ln_g[tid] + p.ln_b[tid];
            xn[256 + tid] = d1 * rstd * p.ln_g[256 + tid] + p.ln_b[256 + tid];
        }
    }
    __syncthreads();
    // ---- this head's projections
    {
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const f4 xr = reinterpret_cast<const f4*>(xn)[8 * i + part];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = fmaf(xr[e], wv[i][e], acc);
        }
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        acc += __shfl_xor(acc, 4, 64);
        if (part == 0) {
            const float v = acc + p.bias[row];
            qs[pm * HD + o] = v;
            if (SELF && pm == 1) p.kw[(size_t)b * p.seq_stride + (size_t)Tc * p.ldkv + h * HD + o] = v;
            if (SELF && pm == 2) p.vw[(size_t)b * p.seq_stride + (size_t)Tc * p.ldkv + h * HD + o] = v;
        }
    }
    __syncthreads();
    // ---- single-query attention (threads 0 .. 255; this step's k, v come from LDS)
    const int T = Tc + (SELF ? 1 : 0);
    f4 q[8];
#pragma unroll
    for (int d = 0; d < 8; ++d) q[d] = reinterpret_cast<const f4*>(qs)[d];
    auto krow = [&](int j) -> const float* { return (!SELF || j < Tc) ? p.kc + (size_t)b * p.seq_stride + (size_t)j * p.ldkv + h * HD : qs + HD; };
    auto vrow = [&](int j) -> const float* { return (!SELF || j < Tc) ? p.vc + (size_t)b * p.seq_stride + (size_t)j * p.ldkv + h * HD : qs + 2 * HD; };
    float mx = -INFINITY;
    if (att) {
        for (int j = tid; j < T; j += 256) {
            f4 k[8];
            if (j == tid && kpre) {
#pragma unroll
                for (int d = 0; d < 8; ++d) k[d] = k0[d];
            } else {
                const f4* kr = reinterpret_cast<const f4*>(krow(j));
#pragma unroll
                for (int d = 0; d < 8; ++d) k[d] = kr[d];
            }
            float s = 0.f;
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int e = 0; e < 4; ++e) s = fmaf(q[d][e], k[d][e], s);
            sc[j] = s;
            mx = fmaxf(mx, s);
        }
    }
#pragma unroll
    for (int o2 = 32; o2 > 0; o2 >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o2, 64));
    if (att && lane == 0) red[wave] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.f;
    if (att) {
        for (int j = tid; j < T; j += 256) {
            const float e = __expf(sc[j] - mx);
            sc[j] = e;
            sum += e;
        }
    }
    sum = wsum(sum);
    __syncthreads();                       // (red[0..3] read by everybody; sc[] complete)
    if (att && lane == 0) red[4 + wave] = sum;
    __syncthreads();
    const float inv = 1.f / (red[4] + red[5] + red[6] + red[7]);
    // output: thread (g = tid / 8: one of 32 key groups, dq = tid % 8: dims 4 dq .. + 3), four rows in flight
    const int dq = tid & 7, g = tid >> 3;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    if (att) {
        int j = g;
        for (; j + 96 < T; j += 128) {
            const f4 v0 = reinterpret_cast<const f4*>(vrow(j))[dq], v1 = reinterpret_cast<const f4*>(vrow(j + 32))[dq];
            const f4 v2 = reinterpret_cast<const f4*>(vrow(j + 64))[dq], v3 = reinterpret_cast<const f4*>(vrow(j + 96))[dq];
            acc += v0 * sc[j];
            acc += v1 * sc[j + 32];
            acc += v2 * sc[j + 64];
            acc += v3 * sc[j + 96];
        }
        for (; j < T; j += 32) acc += reinterpret_cast<const f4*>(vrow(j))[dq] * sc[j];
    }
    __syncthreads();                       // red[] is reused below
#pragma unroll
    for (int o2 = 8; o2 < 64; o2 <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] += __shfl_xor(acc[e], o2, 64);
    }
    if (att && lane < 8) *reinterpret_cast<f4*>(&red[wave * 32 + 4 * lane]) = acc;
    __syncthreads();
    if (tid < 32) p.out[(size_t)b * p.ldo + h * HD + tid] = (red[tid] + red[32 + tid] + red[64 + tid] + red[96 + tid]) * inv;
}

// Round 6: the decode step's linears (M = the decode batch, K = 512 / 2048) as weight-streaming GEMVs.  profiles/r4_formula_decode.txt: 25 of a
// step's 39 launches were skinny2_gemm_kernel at 5 - 35 us each for 1 - 4 MB of weights (fc2: 4 MB in 15 us = 0.28 TB/s) - 32 workgroups of
// four wavefronts, every wavefront four columns x two serial K passes.  The guide's price list puts a dependent kernel boundary at 1.2 - 1.9 us
// and a 256-workgroup grid barrier at 4 - 5 us, so folding the step into ONE persistent kernel trades every boundary for
// something dearer; what a launch costs here is its own latency chain, and that is what this kernel shortens:
//   * ONE column group (CW columns x all of K) per wavefront and pass, K in a single pass (KPL float4 per lane and column): every byte of a
//     wavefront's weights is requested by its first instructions, before X is staged and normalised, and N = 512 fills 128 workgroups
//     instead of 32;
//   * wide layers (lm_head: 50 000 columns, 102 MB) loop over column groups with the NEXT group's weights requested before this group's
//     arithmetic (register double buffer): the stream never drains, X is staged (and normalised: the final LayerNorm rides here instead of
//     in a launch of its own) once per workgroup;
//   * arithmetic identical to skinny2_gemm_kernel (kernels_conv.hip), bit for bit: a lane owns k = 4 lane + 256 h, h ascending, the same
//     product expression, the same halving butterfly (sums over lanes in the order xor 32, 16, .. 1), the same LayerNorm - the decoded ids
//     cannot move (tests/test_gpu_round3.py::test_formula_decoder_300_tokens_equals_reference; RD_DEC_GEMV=0 keeps the old launches for A/B).
struct GemvParams {
    const float* x; int M, K, N;      // x [M][K]
    const float* w; const float* bias; // w [N][K]
    const float* ln_g; const float* ln_b;
    const float* res;                 // [M][N] or nullptr, added after the activation
    float* y;                         // [M][N]
    int act;
};
// DX (so / co / fc2: no LayerNorm in front, M <= 8): X is not staged at all - a lane only ever multiplies the X values of ITS k positions,
// so it reads those float4 (L2 hits: the same 16 / 64 KB for every workgroup) straight into registers next to its weight loads - at most 32
// at a time, i.e. fc2 (K = 2048) in two batches of four rows: no LDS write, no barrier.  Same products in the same order per output.
template <int MT, int CW, int KPL, bool DB, bool DX = false>
__global__ void __launch_bounds__(256) dec_gemv_kernel(GemvParams p, int n_groups) {
    static_assert(!DX || !DB, "direct X: one column group per wavefront");
    constexpr int XR = !DX ? 1 : (MT * KPL <= 16 ? MT : 32 / KPL);     // DX: rows whose X values sit in registers at a time (<= 32 float4)
    static_assert(!DX || (MT % XR == 0 && CW == 1), "direct X: whole row batches, one column per wavefront");
    typedef float f4 __attribute__((ext_vector_type(4)));
    constexpr int NV = MT * CW, KP = 256 * KPL;       // K == KP (host)
    extern __shared__ __attribute__((aligned(16))) float gx[];          // [MT][KP]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gw = (int)blockIdx.x * 4 + wave, GW = (int)gridDim.x * 4;
    f4 wa[CW][KPL], wb[DB ? CW : 1][DB ? KPL : 1];      // DB: two register sets alternate (wide layers)
    auto load_w = [&](f4 (&wr)[CW][KPL], int g) {
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const int n = min(g * CW + c, p.N - 1);            // clamped: columns past N are computed on the last column and never stored
            const f4* wp = reinterpret_cast<const f4*>(p.w + (size_t)n * KP) + lane;
#pragma unroll
            for (int h = 0; h < KPL; ++h) wr[c][h] = wp[64 * h];
        }
    };
    if (gw < n_groups) load_w(wa, gw);
    f4 xd[DX ? XR : 1][DX ? KPL : 1];
    auto load_x = [&](int m0) {       // rows m0 .. m0 + XR of X, this lane's k positions
        if constexpr (DX) {
#pragma unroll
            for (int m = 0; m < XR; ++m)
#pragma unroll
                for (int h = 0; h < KPL; ++h)
                    xd[m][h] = m0 + m < p.M ? reinterpret_cast<const f4*>(p.x + (size_t)(m0 + m) * KP)[lane + 64 * h] : f4{0.f, 0.f, 0.f, 0.f};
        }
    };
    load_x(0);
    // X -> LDS, fused pre-LayerNorm (wave w normalises rows w, w + 4, ...): skinny2_gemm_kernel's code
    if constexpr (!DX)
    for (int i = tid; i < MT * (KP / 4); i += 256) {
        const int m = i / (KP / 4), k = 4 * (i - m * (KP / 4));
        *reinterpret_cast<f4*>(&gx[m * KP + k]) = m < p.M ? *reinterpret_cast<const f4*>(p.x + (size_t)m * KP + k) : f4{0.f, 0.f, 0.f, 0.f};
    }
    if constexpr (!DX) __syncthreads();
    if (!DX && p.ln_g) {
        if constexpr (KP == 512) {
            for (int m = wave; m < MT; m += 4) {
                float v[8], s1 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    v[i] = gx[m * KP + lane + 64 * i];
                    s1 += v[i];
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
                const float mean = s1 / KP;
                float s2 = 0.f;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const float d = v[i] - mean;
                    s2 += d * d;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
                const float rstd = rsqrtf(s2 / KP + 1e-5f);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int k = lane + 64 * i;
                    gx[m * KP + k] = (v[i] - mean) * rstd * p.ln_g[k] + p.ln_b[k];
                }
            }
        }
        __syncthreads();
    }
    auto finish = [&](const f4 (&wr)[CW][KPL], int g) {
        float acc[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) acc[i] = 0.f;
        if constexpr (DX) {
            // row batch by row batch (a value acc[m] only ever sees its own row: per (column, row) the products still arrive in h order)
#pragma unroll
            for (int m0 = 0; m0 < MT; m0 += XR) {
                if (m0) load_x(m0);
#pragma unroll
                for (int h = 0; h < KPL; ++h)
#pragma unroll
                    for (int m = 0; m < XR; ++m) {
                        const f4 xv = xd[m][h];
                        acc[m0 + m] += xv[0] * wr[0][h][0] + xv[1] * wr[0][h][1] + xv[2] * wr[0][h][2] + xv[3] * wr[0][h][3];
                    }
            }
        } else {
#pragma unroll
        for (int h = 0; h < KPL; ++h) {
            const int kq = h * 256 + 4 * lane;
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                const f4 xv = *reinterpret_cast<const f4*>(&gx[m * KP + kq]);
#pragma unroll
                for (int c = 0; c < CW; ++c)
                    acc[c * MT + m] += xv[0] * wr[c][h][0] + xv[1] * wr[c][h][1] + xv[2] * wr[c][h][2] + xv[3] * wr[c][h][3];
            }
            if constexpr (KPL > 2) __builtin_amdgcn_sched_barrier(0);      // keeps the X reads of later h out of this one's registers
        }
        }
        // halving butterfly (skinny2_gemm_kernel): a lane whose bit o is set keeps the upper half of what it carries
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
        constexpr int LOW = NV >= 64 ? 0 : (64 / NV - 1);
        const int c = idx / MT, m = idx - c * MT, n = g * CW + c;
        if ((lane & LOW) == 0 && m < p.M && n < p.N) {
            float v = rd_act(acc[0] + (p.bias ? p.bias[n] : 0.f), p.act);
            if (p.res) v += p.res[(size_t)m * p.N + n];
            p.y[(size_t)m * p.N + n] = v;
        }
    };
    // column groups gw, gw + GW, ...: neighbouring wavefronts stream neighbouring rows of W
    if constexpr (DB) {
        for (int g = gw; g < n_groups; g += 2 * GW) {
            if (g + GW < n_groups) load_w(wb, g + GW);
            finish(wa, g);
            if (g + GW < n_groups) {
                if (g + 2 * GW < n_groups) load_w(wa, g + 2 * GW);
                finish(wb, g + GW);
            }
        }
    } else {
        for (int g = gw; g < n_groups; g += GW) {
            if (g != gw) load_w(wa, g);
            finish(wa, g);
        }
    }
}

// next token: argmax of the logits row (forced EOS at the length limit), pad for finished sequences, append.  The last
// block to finish advances the step counter (arrival ticket), so no separate launch is needed.
// Round 6: the NEXT step's embedding + LayerNorm (dec_embed_ln_kernel's arithmetic, one wavefront) rides at the end of this kernel when
// `emb` is given: one launch fewer per token.
struct NextEmbed { const float* emb; const float* pos; const float* g; const float* b; float* x; int max_new; };
__global__ void __launch_bounds__(1024) dec_select_kernel(const float* logits, int V, long long* ids, int ids_ld, int* unfinished,
                                                          DecState* st, int B, NextEmbed ne) {
    const int b = blockIdx.x, tid = threadIdx.x, t = st->step;
    const float* z = logits + (size_t)b * V;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    if ((V & 3) == 0 && V <= 65536) {
        // 16-byte loads, every load of the thread requested before the first compare (the scalar loop below ran its 49 loads per
        // thread through a compare-and-select chain: 18 us for 200 KB); indices still visited in increasing order per thread
        constexpr int MAXQ = 16;                     // V <= 65536
        const int nq = V >> 2;
        typedef float f4 __attribute__((ext_vector_type(4)));
        f4 q[MAXQ];
#pragma unroll
        for (int it = 0; it < MAXQ; ++it) q[it] = *reinterpret_cast<const f4*>(z + 4 * (size_t)min(tid + 1024 * it, nq - 1));
#pragma unroll
        for (int it = 0; it < MAXQ; ++it) {
            const int c0 = 4 * (tid + 1024 * it);
            if (tid + 1024 * it < nq) {
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (q[it][e] > mx) { mx = q[it][e]; mi = c0 + e; }
            }
        }
    } else
    for (int c = tid; c < V; c += 1024) {
        const float v = z[c];
        if (v > mx) { mx = v; mi = c; }
    }
    // (value, index) pairs are totally ordered - larger value first, then smaller index - so any reduction tree gives the pair the ten-level
    // LDS tree of rounds 1-5 gave: butterfly inside the wavefront, one exchange between the sixteen (round 6: two barriers instead of eleven)
    __shared__ float smx[16];
    __shared__ int smi[16];
    auto better = [](float c, int ic, float a, int ia) { return c > a || (c == a && ic < ia); };
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float c = __shfl_xor(mx, o, 64);
        const int ic = __shfl_xor(mi, o, 64);
        if (better(c, ic, mx, mi)) { mx = c; mi = ic; }
    }
    if ((tid & 63) == 0) { smx[tid >> 6] = mx; smi[tid >> 6] = mi; }
    __syncthreads();
    __shared__ int s_tok;
    if (tid == 0) {
        float a = smx[0];
        int ia = smi[0];
        for (int w = 1; w < 16; ++w)
            if (better(smx[w], smi[w], a, ia)) { a = smx[w]; ia = smi[w]; }
        int tok = ia;
        if (t + 1 == FORCED_EOS_LEN - 1) tok = EOS_ID;          // input length == max_length - 1 -> only EOS survives
        const int unf = unfinished[b];
        tok = unf ? tok : PAD_ID;
        ids[(size_t)b * ids_ld + t + 1] = tok;
        s_tok = tok;
        if (unf && tok == EOS_ID) {
            unfinished[b] = 0;
            atomicSub(&st->n_unfinished, 1);
        }
        __threadfence();
        if (atomicAdd(&st->arrived, 1) == B - 1) {   // every block has read `step` (t) before its ticket: safe to advance
            st->arrived = 0;
            st->step = t + 1;
        }
    }
    if (!ne.emb || t + 1 >= ne.max_new) return;      // (uniform over the block)
    __syncthreads();
    if (tid < 64) {    // x[b] = LayerNorm(emb[tok] + pos[(t + 1) + 2]): the input row of the next step
        const long long tok = s_tok;
        const int lane = tid;
        float v[8];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            v[i] = ne.emb[(size_t)tok * D + c] + ne.pos[(size_t)(t + 1 + 2) * D + c];
            s += v[i];
        }
        const float mean = wsum(s) / D;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) q += (v[i] - mean) * (v[i] - mean);
        const float rstd = rsqrtf(wsum(q) / D + 1e-5f);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int c = lane + 64 * i;
            ne.x[(size_t)b * D + c] = (v[i] - mean) * rstd * ne.g[c] + ne.b[c];
        }
    }
}
__global__ void dec_advance_kernel(DecState* st) { st->step += 1; }
__global__ void dec_init_kernel(DecState* st, int* unfinished, long long* ids, int ids_ld, int B) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) { st->step = 0; st->n_unfinished = B; st->arrived = 0; }
    if (i < B) unfinished[i] = 1;
    for (long long k = i; k < (long long)B * ids_ld; k += (long long)gridDim.x * blockDim.x) ids[k] = (k % ids_ld == 0) ? START_ID : PAD_ID;
}

// ------------------------------------------------------------------------------------------------------------------
class FormulaDecoder {
   public:
    explicit FormulaDecoder(int device) : device_(device) {}
    ~FormulaDecoder() {
        (void)hipSetDevice(device_);
        if (buf_) (void)hipFree(buf_);
    }
    void load(const WeightStore& ws);
    // enc [B,S,2048] (device) -> ids [B][max_new+1] (device, int64); returns the number of columns the reference returns
    int decode(const float* enc, int B, int S, int max_new, long long* ids_out, hipStream_t s);
    int max_positions() const { return n_pos_ - 2; }

   private:
    void gemm(const float* x, int M, int K, const std::string& key, int N, float* y, int act, const float* res, hipStream_t s,
              const std::string& ln_key = "", float* ln_tmp = nullptr);
    // the decode step's linears (M <= 32 rows, K = 512 / 2048) through dec_gemv_kernel; false = shape not covered, nothing launched
    bool gemv(const float* x, int M, int K, const std::string& key, int N, float* y, int act, const float* res, hipStream_t s,
              const std::string& ln_key = "");
    void ln(const float* x, const std::string& key, float* y, int M, hipStream_t s) {
        launch_layernorm(x, D, y, D, params_.ptr(key + ".weight"), params_.ptr(key + ".bias"), M, D, 1e-5f, s);
    }
    int device_;
    int n_layers_ = 0, vocab_ = 0, n_pos_ = 0, enc_dim_ = 0;
    ParamBlock params_;
    uint8_t* buf_ = nullptr;
    size_t buf_bytes_ = 0;
};

static std::vector<float> vec_of(const HostTensor& t) { return std::vector<float>(t.f32(), t.f32() + t.numel()); }

void FormulaDecoder::load(const WeightStore& ws) {
    const std::string P = "head.decoder.model.decoder.";
    const HostTensor& emb = ws.get(P + "embed_tokens.weight");
    vocab_ = (int)emb.shape[0];
    RD_CHECK((int)emb.shape[1] == D, "formula decoder: d_model must be 512");
    {
        std::vector<float> e = vec_of(emb);
        const float sc = std::sqrt((float)D);  // scale_embedding=True (rec_ppformulanet_head.py:769)
        for (auto& v : e) v *= sc;
        params_.add("emb", e);
    }
    const HostTensor& pos = ws.get(P + "embed_positions.weight");
    n_pos_ = (int)pos.shape[0];
    params_.add("pos", vec_of(pos));
    auto add_ln = [&](const std::string& name, const std::string& key) {
        params_.add(key + ".weight", vec_of(ws.get(name + ".weight")));
        params_.add(key + ".bias", vec_of(ws.get(name + ".bias")));
    };
    add_ln(P + "layernorm_embedding", "ln_emb");
    add_ln(P + "layer_norm", "ln_out");
    const float qs = 1.0f / std::sqrt((float)HD);
    auto lin = [&](const std::string& name, const std::string& key, float scale) {
        std::vector<float> w = vec_of(ws.get(name + ".weight"));
        for (auto& v : w) v *= scale;
        params_.add(key + "#w", w);
        if (ws.has(name + ".bias")) {
            std::vector<float> b = vec_of(ws.get(name + ".bias"));
            for (auto& v : b) v *= scale;
            params_.add(key + "#b", b);
        }
    };
    n_layers_ = 0;
    while (ws.has(P + "layers." + std::to_string(n_layers_) + ".fc1.weight")) ++n_layers_;
    RD_CHECK(n_layers_ > 0, "formula decoder: no layers found");
    for (int l = 0; l < n_layers_; ++l) {
        const std::string L = P + "layers." + std::to_string(l) + ".", K = "l" + std::to_string(l) + ".";
        add_ln(L + "self_attn_layer_norm", K + "ln1");
        add_ln(L + "encoder_attn_layer_norm", K + "ln2");
        add_ln(L + "final_layer_norm", K + "ln3");
        // fused q|k|v projection [1536][512]; q rows carry the head_dim^-0.5 scaling (rec_unimernet_head.py:553)
        std::vector<float> w, bvec;
        for (const char* nm : {"q_proj", "k_proj", "v_proj"}) {
            std::vector<float> a = vec_of(ws.get(L + "self_attn." + nm + ".weight")), bb = vec_of(ws.get(L + "self_attn." + nm + ".bias"));
            const float sc = std::string(nm) == "q_proj" ? qs : 1.f;
            for (auto& v : a) v *= sc;
            for (auto& v : bb) v *= sc;
            w.insert(w.end(), a.begin(), a.end());
            bvec.insert(bvec.end(), bb.begin(), bb.end());
        }
        params_.add(K + "qkv#w", w);
        params_.add(K + "qkv#b", bvec);
        lin(L + "self_attn.out_proj", K + "so", 1.f);
        lin(L + "encoder_attn.q_proj", K + "cq", qs);
        w.clear(); bvec.clear();
        for (const char* nm : {"k_proj", "v_proj"}) {
            std::vector<float> a = vec_of(ws.get(L + "encoder_attn." + nm + ".weight")), bb = vec_of(ws.get(L + "encoder_attn." + nm + ".bias"));
            w.insert(w.end(), a.begin(), a.end());
            bvec.insert(bvec.end(), bb.begin(), bb.end());
        }
        params_.add(K + "ckv#w", w);
        params_.add(K + "ckv#b", bvec);
        lin(L + "encoder_attn.out_proj", K + "co", 1.f);
        lin(L + "fc1", K + "fc1", 1.f);
        lin(L + "fc2", K + "fc2", 1.f);
    }
    lin("head.decoder.lm_head", "lm", 1.f);
    if (ws.has("head.enc_to_dec_proj.weight")) {
        enc_dim_ = (int)ws.get("head.enc_to_dec_proj.weight").shape[1];
        lin("head.enc_to_dec_proj", "encp", 1.f);
    } else {
        enc_dim_ = D;
    }
    params_.upload();
}

// y = act(LN?(x) . W^T + b) (+ res).  With `ln_key` the pre-LayerNorm is fused into the small-M GEMM when it applies, otherwise
// it runs as its own kernel through `ln_tmp`.
void FormulaDecoder::gemm(const float* x, int M, int K, const std::string& key, int N, float* y, int act, const float* res, hipStream_t s,
                          const std::string& ln_key, float* ln_tmp) {
    ConvParams p{};
    if (!ln_key.empty()) {
        if (skinny_gemm_applies(M, K) && N <= 4096) {   // every workgroup re-normalises X: only worth it for few workgroups
            p.ln_g = params_.ptr(ln_key + ".weight");
            p.ln_b = params_.ptr(ln_key + ".bias");
        } else {
            ln(x, ln_key, ln_tmp, M, s);
            x = ln_tmp;
        }
    }
    p.x = x; p.xld = K; p.N = 1; p.H = 1; p.W = M; p.Cin = K;
    p.w = params_.ptr(key + "#w");
    p.bias = params_.has(key + "#b") ? params_.ptr(key + "#b") : nullptr;
    p.y = y; p.yld = N; p.OH = 1; p.OW = M; p.Cout = N;
    p.KH = p.KW = p.SH = p.SW = 1;
    p.res = res; p.rld = N;
    p.act = act; p.out_mode = OUT_NHWC;
    p.M = M; p.K = K; p.Ng = N;
    p.allow_skinny = 1;      // M is the decode batch
    launch_conv_igemm(p, s);
}

template <int MT, int CW, int KPL, bool DB, bool DX = false>
static void launch_gemv(const GemvParams& p, int max_wgs, hipStream_t s) {
    const int n_groups = (p.N + CW - 1) / CW;
    const size_t lds = DX ? 0 : (size_t)MT * 256 * KPL * sizeof(float);
    static unsigned long long lds_ok = 0;
    rd_allow_dynamic_lds((const void*)dec_gemv_kernel<MT, CW, KPL, DB, DX>, lds, lds_ok);
    const int grid = std::min((n_groups + 3) / 4, max_wgs);
    hipLaunchKernelGGL((dec_gemv_kernel<MT, CW, KPL, DB, DX>), dim3(grid), dim3(256), lds, s, p, n_groups);
}

bool FormulaDecoder::gemv(const float* x, int M, int K, const std::string& key, int N, float* y, int act, const float* res, hipStream_t s,
                          const std::string& ln_key) {
    static const bool on = [] { const char* e = getenv("RD_DEC_GEMV"); return !(e && e[0] == '0'); }();
    // developer knobs (tools/bench_formula.py sweeps): workgroups of a wide layer's loop, columns per wavefront of a wide layer
    static const int wide_wgs = [] { const char* e = getenv("RD_DEC_GEMV_WGS"); return e ? atoi(e) : 512; }();
    if (!on || M > 32 || (K != 512 && K != 2048) || (K == 2048 && (M > 16 || !ln_key.empty()))) return false;
    GemvParams p{};
    p.x = x; p.M = M; p.K = K; p.N = N;
    p.w = params_.ptr(key + "#w");
    p.bias = params_.has(key + "#b") ? params_.ptr(key + "#b") : nullptr;
    if (!ln_key.empty()) {
        p.ln_g = params_.ptr(ln_key + ".weight");
        p.ln_b = params_.ptr(ln_key + ".bias");
    }
    p.res = res; p.y = y; p.act = act;
    const bool wide = N > 4096;          // loops over column groups with the next group's weights in flight
    if (K == 2048) {
        static const bool dx8 = [] { const char* e = getenv("RD_DEC_GEMV_DX"); return !(e && e[0] == '0'); }();     // A/B
        if (M <= 8 && dx8) launch_gemv<8, 1, 8, false, true>(p, 1024, s);
        else if (M <= 8) launch_gemv<8, 1, 8, false>(p, 1024, s);
        else launch_gemv<16, 1, 8, false>(p, 1024, s);
    } else if (M <= 8) {
        static const bool dx = [] { const char* e = getenv("RD_DEC_GEMV_DX"); return !(e && e[0] == '0'); }();     // A/B
        if (wide) launch_gemv<8, 4, 2, true>(p, wide_wgs, s);
        else if (dx && ln_key.empty()) launch_gemv<8, 1, 2, false, true>(p, 1024, s);
        else launch_gemv<8, 1, 2, false>(p, 1024, s);
    } else if (M <= 16) {
        if (wide) launch_gemv<16, 2, 2, true>(p, wide_wgs, s);
        else launch_gemv<16, 1, 2, false>(p, 1024, s);
    } else {
        if (wide) launch_gemv<32, 1, 2, true>(p, wide_wgs, s);
        else launch_gemv<32, 1, 2, false>(p, 1024, s);
    }
    return true;
}

int FormulaDecoder::decode(const float* enc, int B, int S, int max_new, long long* ids_out, hipStream_t s) {
    RD_HIP(hipSetDevice(device_));
    RD_CHECK(B > 0 && S > 0 && max_new > 0, "formula decode: empty batch");
    RD_CHECK(max_new + 2 <= n_pos_, "formula decode: max_new_tokens exceeds the positional table of these weights");
    const int Tmax = max_new + 1, ids_ld = max_new + 1;
    auto al = [](size_t v) { return (v + 255) & ~size_t(255); };
    const size_t f = sizeof(float);
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al(bytes); return o; };
    const size_t o_state = take(sizeof(DecState)), o_unf = take(B * sizeof(int));
    const size_t o_encp = take((size_t)B * S * D * f);
    const size_t o_ckv = take((size_t)n_layers_ * B * S * 2 * D * f);
    const size_t o_kc = take((size_t)n_layers_ * B * Tmax * D * f), o_vc = take((size_t)n_layers_ * B * Tmax * D * f);
    const size_t o_x = take((size_t)B * D * f), o_x2 = take((size_t)B * D * f), o_h = take((size_t)B * D * f);
    const size_t o_qkv = take((size_t)B * 3 * D * f), o_a = take((size_t)B * D * f), o_f = take((size_t)B * FFN * f);
    const size_t o_lg = take((size_t)B * vocab_ * f);
    if (off > buf_bytes_) {
        RD_HIP(hipStreamSynchronize(s));
        if (buf_) RD_HIP(hipFree(buf_));
        buf_ = nullptr;
        RD_HIP(hipMalloc((void**)&buf_, off));
        buf_bytes_ = off;
    }
    DecState* st = reinterpret_cast<DecState*>(buf_ + o_state);
    int* unf = reinterpret_cast<int*>(buf_ + o_unf);
    float* encp = reinterpret_cast<float*>(buf_ + o_encp);
    float* ckv = reinterpret_cast<float*>(buf_ + o_ckv);
    float* kc = reinterpret_cast<float*>(buf_ + o_kc);
    float* vc = reinterpret_cast<float*>(buf_ + o_vc);
    float* x = reinterpret_cast<float*>(buf_ + o_x);
    float* x2 = reinterpret_cast<float*>(buf_ + o_x2);
    float* h = reinterpret_cast<float*>(buf_ + o_h);
    float* qkv = reinterpret_cast<float*>(buf_ + o_qkv);
    float* a = reinterpret_cast<float*>(buf_ + o_a);
    float* ff = reinterpret_cast<float*>(buf_ + o_f);
    float* lg = reinterpret_cast<float*>(buf_ + o_lg);

    hipLaunchKernelGGL(dec_init_kernel, dim3(64), dim3(256), 0, s, st, unf, ids_out, ids_ld, B);
    // once per batch: project the encoder states and derive every layer's cross-attention K | V
    const float* enc_d = enc;
    if (params_.has("encp#w")) {
        gemm(enc, B * S, enc_dim_, "encp", D, encp, ACT_NONE, nullptr, s);
        enc_d = encp;
    }
    for (int l = 0; l < n_layers_; ++l)
        gemm(enc_d, B * S, D, "l" + std::to_string(l) + ".ckv", 2 * D, ckv + (size_t)l * B * S * 2 * D, ACT_NONE, nullptr, s);

    const size_t attn_sh_self = (size_t)(((Tmax + 3) & ~3) + 128) * f, attn_sh_cross = (size_t)(((S + 3) & ~3) + 128) * f;
    // fused projection + attention launches (dec_attn_fused_kernel): RD_DEC_FUSED=0 keeps the round-3 form (A/B, parity tests)
    static const bool fused_env = [] { const char* e = getenv("RD_DEC_FUSED"); return !(e && e[0] == '0'); }();
    const bool fused = fused_env && B <= 32;
    static const bool attn2 = [] { const char* e = getenv("RD_DEC_ATTN2"); return !(e && e[0] == '0'); }();
    const size_t fsh_self = (size_t)(D + 3 * HD + 256 + ((Tmax + 3) & ~3)) * f, fsh_cross = (size_t)(D + 3 * HD + 256 + ((S + 3) & ~3)) * f;
    // One decode step = ~70 dependent launches whose arguments never change (the step index lives in device memory), so
    // the step is captured once into a hipGraph and replayed: the host cost per step drops from ~70 launches to one.
    // a linear of the step: the weight-streaming GEMV where it covers the shape, the round-3 skinny GEMM otherwise
    auto lin = [&](const float* xin, int K, const std::string& key, int N, float* y, int act, const float* res, const std::string& ln_key = "",
                   float* ln_tmp = nullptr) {
        if (!gemv(xin, B, K, key, N, y, act, res, s, ln_key)) gemm(xin, B, K, key, N, y, act, res, s, ln_key, ln_tmp);
    };
    // the embedding of step t + 1 rides in step t's select launch (RD_DEC_EMBED_IN_SELECT=0: its own launch, as rounds 1-5); the first
    // step's is enqueued once, ahead of the loop
    static const bool embed_in_select = [] { const char* e = getenv("RD_DEC_EMBED_IN_SELECT"); return !(e && e[0] == '0'); }();
    auto enqueue_embed = [&]() {
        hipLaunchKernelGGL(dec_embed_ln_kernel, dim3(B), dim3(64), 0, s, params_.ptr("emb"), params_.ptr("pos"), ids_out, ids_ld, st,
                           params_.ptr("ln_emb.weight"), params_.ptr("ln_emb.bias"), x);
    };
    auto enqueue_step = [&]() {
        if (!embed_in_select) enqueue_embed();
        float* cur = x;
        float* nxt = x2;
        for (int l = 0; l < n_layers_; ++l) {
            const std::string K = "l" + std::to_string(l) + ".";
            if (fused) {
                AttnFusedParams sp{};
                sp.x = cur; sp.ln_g = params_.ptr(K + "ln1.weight"); sp.ln_b = params_.ptr(K + "ln1.bias");
                sp.w = params_.ptr(K + "qkv#w"); sp.bias = params_.ptr(K + "qkv#b");
                sp.kc = kc + (size_t)l * B * Tmax * D; sp.vc = vc + (size_t)l * B * Tmax * D; sp.ldkv = D; sp.seq_stride = (long long)Tmax * D;
                sp.kw = kc + (size_t)l * B * Tmax * D; sp.vw = vc + (size_t)l * B * Tmax * D;
                sp.st = st; sp.fixed_T = 0; sp.out = a; sp.ldo = D;
                if (attn2) hipLaunchKernelGGL((dec_attn_fused2_kernel<true>), dim3(B, HEADS), dim3(768), fsh_self, s, sp);
                else hipLaunchKernelGGL((dec_attn_fused_kernel<true>), dim3(B, HEADS), dim3(256), fsh_self, s, sp);
                lin(a, D, K + "so", D, nxt, ACT_NONE, cur);
                std::swap(cur, nxt);
                AttnFusedParams xp{};
                xp.x = cur; xp.ln_g = params_.ptr(K + "ln2.weight"); xp.ln_b = params_.ptr(K + "ln2.bias");
                xp.w = params_.ptr(K + "cq#w"); xp.bias = params_.ptr(K + "cq#b");
                xp.kc = ckv + (size_t)l * B * S * 2 * D; xp.vc = xp.kc + D; xp.ldkv = 2 * D; xp.seq_stride = (long long)S * 2 * D;
                xp.st = st; xp.fixed_T = S; xp.out = a; xp.ldo = D;
                if (attn2) hipLaunchKernelGGL((dec_attn_fused2_kernel<false>), dim3(B, HEADS), dim3(256), fsh_cross, s, xp);
                else hipLaunchKernelGGL((dec_attn_fused_kernel<false>), dim3(B, HEADS), dim3(256), fsh_cross, s, xp);
                lin(a, D, K + "co", D, nxt, ACT_NONE, cur);
                std::swap(cur, nxt);
                lin(cur, D, K + "fc1", FFN, ff, ACT_GELU, nullptr, K + "ln3", h);
                lin(ff, FFN, K + "fc2", D, nxt, ACT_NONE, cur);
                std::swap(cur, nxt);
                continue;
            }
            // self-attention
            gemm(cur, B, D, K + "qkv", 3 * D, qkv, ACT_NONE, nullptr, s, K + "ln1", h);
            AttnDecParams ap{};
            ap.q = qkv; ap.ldq = 3 * D;
            ap.kc = kc + (size_t)l * B * Tmax * D; ap.vc = vc + (size_t)l * B * Tmax * D; ap.ldkv = D; ap.seq_stride = (long long)Tmax * D;
            ap.kcur = qkv + D; ap.vcur = qkv + 2 * D; ap.ldcur = 3 * D;
            ap.kw = kc + (size_t)l * B * Tmax * D; ap.vw = vc + (size_t)l * B * Tmax * D;
            ap.st = st; ap.fixed_T = 0; ap.out = a; ap.ldo = D;
            hipLaunchKernelGGL(dec_attention_kernel, dim3(B, HEADS), dim3(128), attn_sh_self, s, ap);
            gemm(a, B, D, K + "so", D, nxt, ACT_NONE, cur, s);
            std::swap(cur, nxt);
            // cross-attention over the (projected) encoder tokens
            gemm(cur, B, D, K + "cq", D, qkv, ACT_NONE, nullptr, s, K + "ln2", h);
            AttnDecParams cp{};
            cp.q = qkv; cp.ldq = D;
            cp.kc = ckv + (size_t)l * B * S * 2 * D; cp.vc = cp.kc + D; cp.ldkv = 2 * D; cp.seq_stride = (long long)S * 2 * D;
            cp.st = st; cp.fixed_T = S; cp.out = a; cp.ldo = D;
            hipLaunchKernelGGL(dec_attention_kernel, dim3(B, HEADS), dim3(128), attn_sh_cross, s, cp);
            gemm(a, B, D, K + "co", D, nxt, ACT_NONE, cur, s);
            std::swap(cur, nxt);
            // feed-forward
            gemm(cur, B, D, K + "fc1", FFN, ff, ACT_GELU, nullptr, s, K + "ln3", h);
            gemm(ff, B, FFN, K + "fc2", D, nxt, ACT_NONE, cur, s);
            std::swap(cur, nxt);
        }
        lin(cur, D, "lm", vocab_, lg, ACT_NONE, nullptr, "ln_out", h);
        RD_CHECK(cur == x, "formula decode: the layer stack must end in the buffer the next embedding is written to");
        NextEmbed ne{};
        if (embed_in_select) ne = NextEmbed{params_.ptr("emb"), params_.ptr("pos"), params_.ptr("ln_emb.weight"), params_.ptr("ln_emb.bias"), x, max_new};
        hipLaunchKernelGGL(dec_select_kernel, dim3(B), dim3(1024), 0, s, lg, vocab_, ids_out, ids_ld, unf, st, B, ne);
    };
    if (embed_in_select) enqueue_embed();
    hipGraph_t graph = nullptr;
    hipGraphExec_t gexec = nullptr;
    static const bool use_graph = [] { const char* e = getenv("RD_DECODE_GRAPH"); return !(e && e[0] == '0'); }();
    if (use_graph && max_new > 2) {
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) == hipSuccess) {
            enqueue_step();
            if (hipStreamEndCapture(s, &graph) != hipSuccess || hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0) != hipSuccess) {
                if (graph) (void)hipGraphDestroy(graph);
                graph = nullptr;
                gexec = nullptr;
                (void)hipGetLastError();
            }
        } else {
            (void)hipGetLastError();
        }
    }
    int host_unf = B, steps = 0;
    for (int t = 0; t < max_new; ++t) {
        if (gexec) RD_HIP(hipGraphLaunch(gexec, s));
        else enqueue_step();
        steps = t + 1;
        if ((t & 7) == 7 || t + 1 == max_new) {  // all sequences ended? (checked every 8 steps: one small D2H + sync)
            DecState hs;
            RD_HIP(hipMemcpyAsync(&hs, st, sizeof(DecState), hipMemcpyDeviceToHost, s));
            RD_HIP(hipStreamSynchronize(s));
            host_unf = hs.n_unfinished;
            if (host_unf == 0) break;
        }
    }
    if (gexec) {
        RD_HIP(hipStreamSynchronize(s));
        (void)hipGraphExecDestroy(gexec);
        (void)hipGraphDestroy(graph);
    }
    RD_HIP(hipGetLastError());
    // the reference stops right after the step in which the last sequence emitted EOS: trim the run-ahead columns
    if (host_unf == 0) {
        std::vector<long long> ids((size_t)B * ids_ld);
        RD_HIP(hipMemcpyAsync(ids.data(), ids_out, ids.size() * sizeof(long long), hipMemcpyDeviceToHost, s));
        RD_HIP(hipStreamSynchronize(s));
        int last = 0;
        for (int b = 0; b < B; ++b)
            for (int c = 1; c <= steps; ++c)
                if (ids[(size_t)b * ids_ld + c] == EOS_ID) { last = std::max(last, c); break; }
        return last + 1;
    }
    return steps + 1;
}

// ---- C++ side of the C-ABI handle (api.cpp) -----------------------------------------------------------------------
FormulaDecoder* formula_decoder_create(int device, const void* blob, size_t nbytes) {
    RD_HIP(hipSetDevice(device));
    WeightStore ws;
    ws.load_safetensors(blob, nbytes);
    auto* d = new FormulaDecoder(device);
    try {
        d->load(ws);
    } catch (...) {
        delete d;
        throw;
    }
    return d;
}
void formula_decoder_destroy(FormulaDecoder* d) { delete d; }
int formula_decoder_decode(FormulaDecoder* d, const float* enc, int B, int S, int max_new, long long* ids, hipStream_t s) {
    return d->decode(enc, B, S, max_new, ids, s);
}
int formula_decoder_max_new(FormulaDecoder* d) { return d->max_positions(); }

}  // namespace rd
