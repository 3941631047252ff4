// Small-K convolutions (the stems' 2x2 and 1x1 layers: K = 48..256, 12..96 output channels) on the fp16 matrix cores with
// (hi, lo) split operands - arithmetic as in kernels_conv_h3.hip.
//
// These layers are HBM-bound, not MFMA-bound (rec stem2a: 80 MB in, 40 MB out, 3.9 GFLOP; at 4 TB/s that is 30 us against 2 us
// of MFMA issue), and the tiled kernels ran them at 1.5-1.9 TB/s: one or two wavefronts per SIMD, each K tile a
// load -> split -> ds_write -> barrier -> ds_read chain, cannot keep a memory system with ~2 us latency busy.  This kernel has
// no activation staging and no barrier in its loop:
//   * a wavefront owns 32 consecutive output pixels x all output channels; its MFMA A operand comes STRAIGHT from global memory
//     (lane = pixel, 8 consecutive input channels = two float4 loads per tap and k-step; the 2x2 taps of neighbouring pixels
//     and the other k-steps of the same 128-byte line hit in L1 / L2), split in registers;
//   * the whole split weight matrix (<= 36 KB) is copied to LDS once per workgroup (128 pixels) and read as B fragments;
//   * ~100 VGPRs and <= 36 KB of LDS: four to six workgroups = 16-24 wavefronts per CU hide the load latency.
// Any kernel size / stride / padding (the taps are address arithmetic), Cin % 4 == 0, padded K <= 256.
#include <algorithm>
#include <cstdlib>

#include "rd_device.h"

namespace rd {

struct StreamGeom {
    int C16;        // input channels per tap rounded up to 16 (k-steps per tap = C16 / 16)
    int SB;         // bytes per weight row in one fp16 plane: NT * C16 * 2 + 16 (16 * odd: conflict-free ds_read_b128)
    int N32;        // weight rows in LDS
};

template <int NB>
__global__ void __launch_bounds__(256) conv_stream_h3_kernel(ConvParams p, StreamGeom g) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int NT = p.KH * p.KW;
    const int SB = g.SB;
    unsigned char* Wh = smem;
    unsigned char* Wl = smem + (size_t)g.N32 * SB;
    const int Kp = (p.K + 31) & ~31;

    // ---- weights -> LDS: row n = [tap][C16] halfs (channels past Cin zero), 8-byte pieces (Cin % 4 == 0)
    {
        const uint16_t* whg = p.wh;
        const uint16_t* wlg = p.wl;
        const int pieces_row = NT * (g.C16 / 4);
        const int total = g.N32 * pieces_row;
        for (int q = tid; q < total; q += 256) {
            const int row = q / pieces_row, r = q - row * pieces_row;
            const int tap = r / (g.C16 / 4), c4 = (r - tap * (g.C16 / 4)) * 4;
            uint2 vh = {0u, 0u}, vl = {0u, 0u};
            if (row < p.Ng && c4 < p.Cin) {
                const size_t o = (size_t)row * Kp + (size_t)tap * p.Cin + c4;
                vh = *reinterpret_cast<const uint2*>(whg + o);
                vl = *reinterpret_cast<const uint2*>(wlg + o);
            }
            const size_t d = (size_t)row * SB + (size_t)(tap * g.C16 + c4) * 2;
            *reinterpret_cast<uint2*>(Wh + d) = vh;
            *reinterpret_cast<uint2*>(Wl + d) = vl;
        }
    }

    // ---- this lane's output pixel (one 128-pixel tile per workgroup: looping persistent workgroups over tiles measured SLOWER,
    // 56 vs 52 us on rec stem2a - short independent workgroups balance better and the weight copy is not what costs)
    const long m = (long)blockIdx.x * 128 + wave * 32 + l31;
    const long mm = m < p.M ? m : (long)p.M - 1;
    const int ow = (int)(mm % p.OW);
    const long t = mm / p.OW;
    const int oh = (int)(t % p.OH), img = (int)(t / p.OH);
    const float* ximg = p.x + (size_t)img * p.H * p.W * p.xld;
    const int ih0 = oh * p.SH - p.PT, iw0 = ow * p.SW - p.PL;

    f32x16 acc1[NB], acc2[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[nb][r] = acc2[nb][r] = 0.f;
    __syncthreads();

    float amax = 0.f;
    const int KS = g.C16 / 16;
    const unsigned b_lane = (unsigned)l31 * (unsigned)SB + (unsigned)lhi * 16u;
    // One step = (tap, k-step): two 16-byte loads per lane.  The loads are UNCONDITIONAL (padding taps read pixel 0, channels past
    // Cin re-read the last group) and masked afterwards - a load under a condition is waited for before the next one is issued,
    // which made every step two serial round trips - and the loads of step s + 1 are requested before the MFMAs of step s.
    const int nsteps = NT * KS;
    auto step_src = [&](int st, bool& ok0, bool& ok1, int& o1) -> const float* {
        const int tap = st / KS, ks = st - tap * KS;
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        const int ih = ih0 + kh, iw = iw0 + kw;
        const bool ok = (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
        const int c = ks * 16 + 8 * lhi;
        ok0 = ok && c < p.Cin;
        ok1 = ok && c + 4 < p.Cin;
        o1 = min(c + 4, p.Cin - 4) - min(c, p.Cin - 4);      // (4, or 0 when the second group lies past Cin: masked)
        return ximg + ((size_t)(ok ? ih : 0) * p.W + (ok ? iw : 0)) * p.xld + min(c, p.Cin - 4);
    };
    f32x4 x0, x1;
    bool ok0, ok1;
    int o1;
    {
        const float* xp = step_src(0, ok0, ok1, o1);
        x0 = *reinterpret_cast<const f32x4*>(xp);
        x1 = *reinterpret_cast<const f32x4*>(xp + o1);
    }
    for (int st = 0; st < nsteps; ++st) {
        const int tap = st / KS, ks = st - tap * KS;
        f32x4 c0 = ok0 ? x0 : f32x4{0.f, 0.f, 0.f, 0.f}, c1 = ok1 ? x1 : f32x4{0.f, 0.f, 0.f, 0.f};
        if (st + 1 < nsteps) {
            const float* xp = step_src(st + 1, ok0, ok1, o1);
            x0 = *reinterpret_cast<const f32x4*>(xp);
            x1 = *reinterpret_cast<const f32x4*>(xp + o1);
        }
        f16x8 ah, al;
        {
            f16x4 h0, l0, h1, l1;
            rd_split4(c0, h0, l0);
            rd_split4(c1, h1, l1);
            ah = f16x8{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]};
            al = f16x8{l0[0], l0[1], l0[2], l0[3], l1[0], l1[1], l1[2], l1[3]};
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fmaxf(fabsf(c0[e]), fabsf(c1[e])));
        const unsigned bo = b_lane + (unsigned)(tap * g.C16 + ks * 16) * 2u;
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const f16x8 bh = *reinterpret_cast<const f16x8*>(Wh + bo + (unsigned)nb * 32u * (unsigned)SB);
            const f16x8 bl = *reinterpret_cast<const f16x8*>(Wl + bo + (unsigned)nb * 32u * (unsigned)SB);
            acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc1[nb], 0, 0, 0);
            acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc2[nb], 0, 0, 0);
            acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc2[nb], 0, 0, 0);
        }
    }

    // ---- epilogue: lane = output channel, registers = 16 of this wavefront's 32 pixels
    unsigned emax = 0;
    const long mb = (long)blockIdx.x * 128 + wave * 32 + 4 * lhi;
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + l31;
        if (n >= p.Ng) continue;
        const float bv = p.bias ? p.bias[n] : 0.f;
        float o[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[r] = fmaf(acc2[nb][r], 1.f / 2048.f, acc1[nb][r]) + bv;
            emax = max(emax, __float_as_uint(o[r]) & 0x7fffffffu);
        }
        if (p.act == ACT_RELU) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = fmaxf(o[r], 0.f);
        } else if (p.act != ACT_NONE) {
#pragma unroll
            for (int r = 0; r < 16; ++r) o[r] = rd_act(o[r], p.act);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long mo = mb + (r & 3) + 8 * (r >> 2);
            if (mo < p.M) {
                float vv = o[r];
                if (p.res) vv += p.res[(size_t)mo * p.rld + n];
                __builtin_nontemporal_store(vv, &p.y[(size_t)mo * p.yld + n]);
            }
        }
    }
    if ((emax >= 0x7f800000u || !(amax < 65504.f)) && p.range_flag) rd_raise_flag(p.range_flag);
}

static bool stream_geom(const ConvParams& p, StreamGeom& g) {
    const int nt = p.KH * p.KW;
    g.C16 = (p.Cin + 15) / 16 * 16;
    g.N32 = (p.Ng + 31) / 32 * 32;
    const int row = nt * g.C16 * 2;
    g.SB = row + 16;
    if ((g.SB / 16) % 2 == 0) g.SB += 16;
    return nt * g.C16 <= 256 && g.C16 <= 64 && g.N32 <= 96 && (size_t)2 * g.N32 * g.SB <= 48 * 1024;
}

bool conv_stream_h3_supported(const ConvParams& p) {
    if (!p.wh || p.out_mode != OUT_NHWC || p.ascale || p.ln_g) return false;
    if (p.Cin % 4 != 0 || (p.xld % 4) != 0 || p.K % 4 != 0) return false;
    StreamGeom g;
    return stream_geom(p, g);
}
// routing policy (launch_conv_igemm_h3): every layer it supports, unless RD_CONV_STREAM=0
bool conv_stream_h3_applies(const ConvParams& p) {
    static const bool off = [] { const char* e = getenv("RD_CONV_STREAM"); return e && e[0] == '0'; }();
    return !off && conv_stream_h3_supported(p);
}

void launch_conv_stream_h3(const ConvParams& p, hipStream_t s) {
    StreamGeom g;
    if (!stream_geom(p, g)) return;
    const size_t lds = (size_t)2 * g.N32 * g.SB;
    const dim3 grid((unsigned)((p.M + 127) / 128)), block(256);
    switch (g.N32 / 32) {
        case 1: hipLaunchKernelGGL(conv_stream_h3_kernel<1>, grid, block, lds, s, p, g); break;
        case 2: hipLaunchKernelGGL(conv_stream_h3_kernel<2>, grid, block, lds, s, p, g); break;
        default: hipLaunchKernelGGL(conv_stream_h3_kernel<3>, grid, block, lds, s, p, g); break;
    }
}

}  // namespace rd
