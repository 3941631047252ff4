// LightSVTR self-attention (necks/rnn.py:238-318: 8 heads of 15, softmax(q k^T / sqrt(hd)) v per text line) on the split-fp16 matrix
// cores.  The VALU kernel (kernels_misc.hip, one thread per query, K / V broadcast from LDS, two passes over the keys) was bound by
// its 45 FMAs + 12 LDS reads per (query, key) pair: 267 us per launch on the recogniser tail's ~500 lines x 187 tokens (1.6 ms/step).
// Here one wavefront owns 32 queries and walks the keys in tiles of 32, flash-attention style:
//   S^T[key][query] = K_tile . Q^T           one k-step (16 >= hd dims) of v_mfma_f32_32x32x16_f16, operands (hi, lo) split: 3 MFMAs.
//                                            Transposed on purpose: in the C/D layout a lane then holds 16 KEYS of ONE query, so the
//                                            softmax statistics are in-lane reductions plus one exchange with lane ^ 32.
//   online softmax per query                 running max m, running sum l, rescale factor exp(m_old - m_new) (lane-uniform per query)
//   O^T[dim][query] += V^T_tile . P^T        two k-steps (32 keys), 3 MFMAs each.  P^T's C/D registers ARE the B fragments: the k index
//                                            of an MFMA is free as long as A and B agree, so V^T is stored in LDS with its keys permuted
//                                            to the order in which a lane's registers hold them (slot 16 s + 8 h + t <-> key
//                                            16 s + 8 (t / 4) + 4 h + (t % 4) of the tile).
// K (row-major [key][16], 48-byte rows: conflict-free ds_read_b128) and V^T ([dim][keys], permuted) live in LDS as fp16 (hi, lo)
// planes, staged once per (line, head) by the whole workgroup; Q stays in registers.  Arithmetic as in kernels_conv_h3.hip
// (x = hi + lo 2^-11, hi.hi into one accumulator, hi.lo + lo.hi into a second one, fp32 accumulate); exp / max / sums in fp32.
// Lines longer than ATT_MAX_T tokens (a text line wider than ~6000 px at height 48) keep the VALU kernel.
#include <algorithm>
#include <cstdlib>

#include "rd_device.h"

namespace rd {

static constexpr int ATT_MAX_T = 768;       // K + V^T planes: 125 KB of LDS at 768 keys
static constexpr int ATT_KROW = 24;       // halfs per K row in LDS (16 + 8 padding: 48 bytes)

// bytes of one fp16 plane of V^T: 16 rows of (Tpad + pad) halfs, row stride = 48 bytes mod 256 (conflict-free 16-lane ds_read_b128)
__host__ __device__ static inline int att_vrow_halfs(int tpad) {
    int bytes = tpad * 2;
    const int r = bytes % 256;
    bytes += (r <= 48 ? 48 - r : 256 + 48 - r);
    return bytes / 2;
}

template <int HD>
__global__ void __launch_bounds__(256) attention_h3_kernel(const float* __restrict__ qkv, float* __restrict__ o, int T, int heads, float scale,
                                                           const int32_t* __restrict__ seg, int tpad_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
    const int b = blockIdx.x, head = blockIdx.y;
    const int C = heads * HD;
    size_t tok0 = (size_t)b * T;
    if (seg) {
        tok0 = (size_t)seg[2 * b];
        T = seg[2 * b + 1];
    }
    // (an empty line of a ragged batch: nothing to attend over, nothing to write; a line beyond ATT_MAX_T is the VALU kernel's - launch_attention
    //  runs it over the same line table: the kernel that serves a line follows from the LINE's length, never from its launch's longest)
    if (T <= 0 || T > ATT_MAX_T) return;
    const int tpad = (T + 31) & ~31, nkt = tpad >> 5;
    const int vrow = att_vrow_halfs(tpad_max);              // (the launcher sized the allocation with tpad_max)
    _Float16* Kh = reinterpret_cast<_Float16*>(sm);
    _Float16* Kl = Kh + (size_t)tpad_max * ATT_KROW;
    _Float16* Vh = Kl + (size_t)tpad_max * ATT_KROW;
    _Float16* Vl = Vh + (size_t)16 * vrow;
    const float* base = qkv + tok0 * 3 * C + head * HD;

    // ---- stage K and V^T (zero for dims >= HD and keys >= T: a zero key scores 0 and is masked below; a zero V row adds nothing)
    for (int i = threadIdx.x; i < tpad * 16; i += 256) {
        const int j = i >> 4, d = i & 15;
        const bool ok = j < T && d < HD;
        const float* src = base + (size_t)min(j, T - 1) * 3 * C + min(d, HD - 1);
        const float kv = src[C], vv = src[2 * C];
        _Float16 h, l;
        rd_split(ok ? kv : 0.f, h, l);
        Kh[j * ATT_KROW + d] = h;
        Kl[j * ATT_KROW + d] = l;
        rd_split(ok ? vv : 0.f, h, l);
        const int kk = j & 31, s = kk >> 4, r = kk & 15;
        const int pos = (j & ~31) + s * 16 + ((r & 7) >> 2) * 8 + (r >> 3) * 4 + (r & 3);     // slot of key kk inside its tile
        Vh[d * vrow + pos] = h;
        Vl[d * vrow + pos] = l;
    }
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    for (int qb = wave; qb < nkt; qb += 4) {               // this wavefront's blocks of 32 queries
        // Q^T as the B operand: lane (query l31, half lhi) holds dims 8 lhi .. + 7, scaled, split
        const int qi = qb * 32 + l31;
        const float* qrow = base + (size_t)min(qi, T - 1) * 3 * C;
        f16x8 qh, ql;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int d = 8 * lhi + t;
            const float v = qrow[min(d, HD - 1)];
            _Float16 h, l;
            rd_split(d < HD ? v * scale : 0.f, h, l);
            qh[t] = h;
            ql[t] = l;
        }
        f32x16 o1, o2;                                     // O^T accumulators (rows = dims: registers 0 .. 7 are dims < 16)
#pragma unroll
        for (int i = 0; i < 16; ++i) o1[i] = o2[i] = 0.f;
        float m_run = -INFINITY, l_run = 0.f;
        const unsigned ka = (unsigned)(l31 * ATT_KROW + 8 * lhi) * 2u;                     // byte offset of this lane's K fragment in a tile
        const unsigned va = (unsigned)((l31 & 15) * vrow + 8 * lhi) * 2u;                  // ... V^T fragment (rows 16 .. 31 repeat 0 .. 15)
        for (int kt = 0; kt < nkt; ++kt) {
            const unsigned char* kb = reinterpret_cast<const unsigned char*>(Kh) + (size_t)kt * 32 * ATT_KROW * 2 + ka;
            const f16x8 kh = *reinterpret_cast<const f16x8*>(kb);
            const f16x8 kl = *reinterpret_cast<const f16x8*>(kb + (size_t)tpad_max * ATT_KROW * 2);
            f32x16 s1, s2;
#pragma unroll
            for (int i = 0; i < 16; ++i) s1[i] = s2[i] = 0.f;
            s1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, qh, s1, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kh, ql, s2, 0, 0, 0);
            s2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kl, qh, s2, 0, 0, 0);
            // register i <-> key kt * 32 + 8 (i / 4) + 4 lhi + (i % 4) of query l31
            float sv[16], mx = -INFINITY;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int key = kt * 32 + 8 * (i >> 2) + 4 * lhi + (i & 3);
                sv[i] = key < T ? fmaf(s2[i], 1.f / 2048.f, s1[i]) : -INFINITY;
                mx = fmaxf(mx, sv[i]);
            }
            mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
            const float m_new = fmaxf(m_run, mx);          // (finite: every tile holds at least one key < T)
            const float alpha = __expf(m_run - m_new);
            m_run = m_new;
            float psum = 0.f;
            f16x8 ph[2], pl[2];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float pv = __expf(sv[i] - m_new);
                psum += pv;
                _Float16 h, l;
                rd_split(pv, h, l);
                ph[i >> 3][i & 7] = h;
                pl[i >> 3][i & 7] = l;
            }
            l_run = fmaf(l_run, alpha, psum);
#pragma unroll
            for (int i = 0; i < 8; ++i) { o1[i] *= alpha; o2[i] *= alpha; }
            const unsigned char* vb = reinterpret_cast<const unsigned char*>(Vh) + (size_t)kt * 64 + va;
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const f16x8 vh = *reinterpret_cast<const f16x8*>(vb + s * 32);
                const f16x8 vl = *reinterpret_cast<const f16x8*>(vb + s * 32 + (size_t)16 * vrow * 2);
                o1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, ph[s], o1, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vh, pl[s], o2, 0, 0, 0);
                o2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(vl, ph[s], o2, 0, 0, 0);
            }
        }
        l_run += __shfl_xor(l_run, 32, 64);                // the two halves of a query hold different keys
        const float inv = 1.f / l_run;
        if (qi < T) {
            float* op = o + (tok0 + qi) * C + head * HD;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int d = 8 * (i >> 2) + 4 * lhi + (i & 3);
                if (d < HD) op[d] = fmaf(o2[i], 1.f / 2048.f, o1[i]) * inv;
            }
        }
    }
}

bool attention_h3_applies(int T, int hd) {
    static const bool off = [] { const char* e = getenv("RD_ATTN_MFMA"); return e && e[0] == '0'; }();
    return !off && (hd == 15 || hd == 16) && T >= 1 && T <= ATT_MAX_T;
}
int attention_h3_max_t() { return ATT_MAX_T; }

void launch_attention_h3(const float* qkv, float* o, int B, int T, int heads, int hd, float scale, hipStream_t s, const int32_t* seg) {
    const int tpad = std::min((T + 31) & ~31, ATT_MAX_T);       // (ragged launches: longer lines are skipped by the kernel)
    const size_t sh = (size_t)2 * tpad * ATT_KROW * 2 + (size_t)2 * 16 * att_vrow_halfs(tpad) * 2;
    static unsigned long long ok15 = 0, ok16 = 0;
    if (hd == 15) {
        rd_allow_dynamic_lds((const void*)attention_h3_kernel<15>, sh, ok15);
        hipLaunchKernelGGL(attention_h3_kernel<15>, dim3(B, heads), dim3(256), sh, s, qkv, o, T, heads, scale, seg, tpad);
    } else {
        rd_allow_dynamic_lds((const void*)attention_h3_kernel<16>, sh, ok16);
        hipLaunchKernelGGL(attention_h3_kernel<16>, dim3(B, heads), dim3(256), sh, s, qkv, o, T, heads, scale, seg, tpad);
    }
}

}  // namespace rd
