// Device-side helpers shared by the HIP translation units: vector types, the GELU / activation epilogue, the (hi, lo)
// fp16 split of the split-MFMA kernels.  Everything is __forceinline__: no cross-TU device linking.
#pragma once
#include <hip/hip_runtime.h>

#include "rd_kernels.h"

namespace rd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// Raise a handle's range flag.  The word lives in pinned, device-mapped HOST memory (engine.cpp load_weights): the host reads it
// after synchronising on the forward's own events, with no device-to-host copy to queue behind other streams' kernels.  A plain
// system-scope store (every raiser writes the same 1), not an atomic: PCIe atomics are not needed.
__device__ __forceinline__ void rd_raise_flag(unsigned* flag) {
    __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// GELU(x) = x/2 * (1 + erf(x/sqrt2)) as nn.GELU() (approximate='none'), erf by Abramowitz-Stegun 7.1.26
// (|abs err| <= 1.5e-7, i.e. fp32 round-off class): 1 exp + 1 rcp + 7 FMAs instead of libm erff's ~60 instructions.
// v_rcp_f32 (1 ulp) on purpose: `__frcp_rn` compiles to the 10-instruction IEEE division sequence.
__device__ __forceinline__ float rd_gelu(float v) {
    const float z = fabsf(v) * 0.70710678118654752440f;
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float erfz = 1.f - poly * t * __expf(-z * z);
    return 0.5f * v * (1.f + copysignf(erfz, v));
}

// Two GELUs at once for the fused mixers, arranged for the packed fp32 VALU (v_pk_mul / v_pk_fma: one issue slot for two lanes'
// worth of work) - VALU time adds to MFMA time on gfx950, and GELU is most of a mixer's VALU time.  Same
// Abramowitz-Stegun 7.1.26 erfc as rd_gelu, rewritten without the sign handling:
//   GELU(v) = max(v, 0) - 0.5 |v| t p(t) exp(-z^2),  z = |v| / sqrt2,  t = 1 / (1 + 0.3275911 z)
// (v >= 0: 0.5 v (2 - p t e) ; v < 0: 0.5 v (p t e)), with z' = z sqrt(log2 e) so that exp(-z^2) = exp2(-z'^2) is one v_exp_f32 with a
// negated input, and the 0.5 folded into p's coefficients.  7.5 plain + 2 transcendental instructions per element (rd_gelu: ~12 + 2).
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef RD_GELU_SCALAR
__device__ __forceinline__ float rd_gelu1s(float v) {
    constexpr float SQ = 1.2011224087864498f;
    constexpr float S = 0.70710678118654752440f * SQ, D = 0.3275911f / SQ;
    const float a = fabsf(v);
    const float zp = a * S;
    const float t = __builtin_amdgcn_rcpf(fmaf(zp, D, 1.f));
    float pl = fmaf(t, 0.5f * 1.061405429f, 0.5f * -1.453152027f);
    pl = fmaf(pl, t, 0.5f * 1.421413741f);
    pl = fmaf(pl, t, 0.5f * -0.284496736f);
    pl = fmaf(pl, t, 0.5f * 0.254829592f);
    const float q = (a * t) * pl;
    const float e = __builtin_amdgcn_exp2f(-(zp * zp));
    return fmaf(-q, e, fmaxf(v, 0.f));
}
__device__ __forceinline__ f32x2 rd_gelu2(f32x2 v) { return f32x2{rd_gelu1s(v[0]), rd_gelu1s(v[1])}; }
#else
__device__ __forceinline__ f32x2 rd_gelu2(f32x2 v) {
    constexpr float SQ = 1.2011224087864498f;                       // sqrt(log2 e)
    constexpr float S = 0.70710678118654752440f * SQ, D = 0.3275911f / SQ;
    const f32x2 a = {fabsf(v[0]), fabsf(v[1])};
    const f32x2 zp = a * S;
    const f32x2 d = zp * D + 1.f;
    const f32x2 t = {__builtin_amdgcn_rcpf(d[0]), __builtin_amdgcn_rcpf(d[1])};
    f32x2 pl = t * (0.5f * 1.061405429f) + (0.5f * -1.453152027f);
    pl = pl * t + (0.5f * 1.421413741f);
    pl = pl * t + (0.5f * -0.284496736f);
    pl = pl * t + (0.5f * 0.254829592f);
    const f32x2 q = (a * t) * pl;
    const f32x2 m = zp * zp;
    const f32x2 e = {__builtin_amdgcn_exp2f(-m[0]), __builtin_amdgcn_exp2f(-m[1])};
    const f32x2 r = {fmaxf(v[0], 0.f), fmaxf(v[1], 0.f)};
    return r - q * e;
}
#endif

__device__ __forceinline__ float rd_act(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_GELU: return rd_gelu(v);
        case ACT_SILU: return v * __builtin_amdgcn_rcpf(1.f + __expf(-v));   // v_rcp_f32: 1 ulp, vs the IEEE division sequence
        case ACT_SIGMOID: {
            const float r = __builtin_amdgcn_rcpf(1.f + __expf(-v));
            return (r != r) ? 0.f : r;  // nan_to_num (det_db_head.py:143-144)
        }
        case ACT_HSIG: return fminf(fmaxf(v * (1.f / 6.f) + 0.5f, 0.f), 1.f);
        case ACT_HSIG_PADDLE: return fminf(fmaxf(0.2f * v + 0.5f, 0.f), 1.f);
        default: return v;
    }
}

// x = hi + lo * 2^-11 with hi = fp16(x), lo = fp16((x - hi) * 2^11).  The second line is ONE fused op on the f16 source
// (exact: the difference is representable and the scale is a power of two); it compiles to v_fma_mix{lo,hi}_f16, so the
// split costs 2 VALU ops per element (v_cvt_pk_f16_f32 + v_fma_mix) instead of 4.
__device__ __forceinline__ void rd_split(float v, _Float16& hi, _Float16& lo) {
    hi = (_Float16)v;
    lo = (_Float16)__builtin_fmaf((float)hi, -2048.f, v * 2048.f);
}

// Four elements at once, the same arithmetic (bit-identical to rd_split) in the instruction sequence it is meant to be: one packed
// conversion per pair, whose halves are the f16 sources of v_fma_mixlo / mixhi (2 - 2.5 VALU operations per element).  Written as
// instructions because the compiler's own selection for rd_split in a loop over a vector is 3.5 per element (separate conversions for
// the mix sources, or - SLP-vectorised - convert back + packed fp32 multiply-add + convert).
typedef _Float16 rd_f16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void rd_split_pair(float a, float b, float sa, float sb, rd_f16x2& hi, rd_f16x2& lo) {     // sa = 2048 a, sb = 2048 b
    hi = __builtin_convertvector(f32x2{a, b}, rd_f16x2);            // v_cvt_pk_f16_f32 (round to nearest even)
    const float k = -2048.f;
    const unsigned hp = __builtin_bit_cast(unsigned, hi);
    unsigned lp;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lp) : "v"(hp), "s"(k), "v"(sa), "v"(sb));
    lo = __builtin_bit_cast(rd_f16x2, lp);
}
__device__ __forceinline__ void rd_split4(const f32x4 v, f16x4& hi, f16x4& lo) {
    const f32x4 s = v * 2048.f;
    rd_f16x2 h0, l0, h1, l1;
    rd_split_pair(v[0], v[1], s[0], s[1], h0, l0);
    rd_split_pair(v[2], v[3], s[2], s[3], h1, l1);
    hi = f16x4{h0[0], h0[1], h1[0], h1[1]};
    lo = f16x4{l0[0], l0[1], l1[0], l1[1]};
}

}  // namespace rd
