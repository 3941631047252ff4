// Fused PPLCNetV4 channel mixer (point-wise expand -> GELU -> point-wise project -> + residual) on the fp32
// matrix cores.  Reference: PPLCNetV4DepthwiseSeparableConvLayer.forward, rec_lcnetv4.py:226-236, for the
// stride-1 / Cin == Cout blocks (15 of the 19 rec blocks, 9 of the 13 det blocks).
//
// Unfused, the 2C-wide hidden activation makes these layers HBM-bound in fp32 (write 2C + read 2C per pixel on
// top of read C / write C).  Here a block keeps 128 pixels x C of X in LDS, streams the weights in hidden
// chunks of 32 and chains the two GEMMs through registers:
//   GEMM1 (transposed)  Ht[hid 32][pix 32] = W1c . X^T      -> lane holds 16 hidden values of ONE pixel
//   GEMM2               Y[pix 32][C]      += H . W2c^T      with the accumulator registers of GEMM1 used
//                                                            directly as the A operand: the C/D register
//                                                            layout of v_mfma_f32_32x32x2_f32 (row = 8*(r>>2) +
//                                                            4*(lane>>5) + (r&3)) is exactly the k order the
//                                                            B fragment (ds_read_b128 at 8*g + 4*(lane>>5)) uses.
// No LDS round trip, shuffle or barrier between the two GEMMs.
#include "rd_device.h"

namespace rd {


static constexpr int MX_BM = 128;
static constexpr int MX_HC = 32;

template <int C, int DBG>
__global__ void __launch_bounds__(256) lc_mixer_kernel(MixerParams p) {
    constexpr int XS = C + 4;              // LDS row stride of X and W1 chunk (conflict-free b128 reads)
    constexpr int NT = C / 32;             // output n-tiles per wave (C = 48 -> 2 tiles, second half-masked)
    constexpr int NTT = (C + 31) / 32;
    constexpr int WQ = C * 8;              // float4 per weight chunk (both W1c and W2c)
    constexpr int WL = (WQ + 255) / 256;
    constexpr int KG = C / 8;              // k groups of GEMM1
    constexpr int W2ROWS = NTT * 32;
    (void)NT;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* Xs = smem;                        // [128][XS]
    float* W1s = Xs + MX_BM * XS;            // [32][XS]
    float* W2s = W1s + MX_HC * XS;           // [W2ROWS][36]
    float* B1s = W2s + W2ROWS * 36;          // [32]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int m0 = blockIdx.x * MX_BM;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

    // ---- X tile (SE gate applied on the way in)
    {
        constexpr int QPR = C / 4;
        for (int i = tid; i < MX_BM * QPR; i += 256) {
            const int r = i / QPR, q = i - r * QPR;
            const int m = m0 + r;
            f32x4 v = zero4;
            if (m < p.M) {
                v = *reinterpret_cast<const f32x4*>(p.x + (size_t)m * p.xld + 4 * q);
                if (p.gate) v *= *reinterpret_cast<const f32x4*>(p.gate + (size_t)(m / p.HW) * C + 4 * q);
            }
            *reinterpret_cast<f32x4*>(&Xs[r * XS + 4 * q]) = v;
        }
        if (C % 32 != 0) {  // zero the rows of W2s that pad the last n-tile (never loaded)
            for (int i = tid; i < (W2ROWS - C) * 36; i += 256) W2s[C * 36 + i] = 0.f;
        }
    }
    f32x4 w1reg[WL], w2reg[WL], b1reg = zero4;
    auto load_w = [&](int j) {
#pragma unroll
        for (int t = 0; t < WL; ++t) {
            const int i = tid + 256 * t;
            if (i < WQ) {
                const int r1 = i / (C / 4), q1 = i - r1 * (C / 4);          // W1 chunk: 32 rows x C
                w1reg[t] = *reinterpret_cast<const f32x4*>(p.w1 + (size_t)(j * MX_HC + r1) * C + 4 * q1);
                const int r2 = i >> 3, q2 = i & 7;                          // W2 chunk: C rows x 32
                w2reg[t] = *reinterpret_cast<const f32x4*>(p.w2 + (size_t)r2 * (2 * C) + j * MX_HC + 4 * q2);
            }
        }
        if (tid < 8) b1reg = *reinterpret_cast<const f32x4*>(p.b1 + j * MX_HC + 4 * tid);
    };
    auto store_w = [&]() {
#pragma unroll
        for (int t = 0; t < WL; ++t) {
            const int i = tid + 256 * t;
            if (i < WQ) {
                const int r1 = i / (C / 4), q1 = i - r1 * (C / 4);
                *reinterpret_cast<f32x4*>(&W1s[r1 * XS + 4 * q1]) = w1reg[t];
                const int r2 = i >> 3, q2 = i & 7;
                *reinterpret_cast<f32x4*>(&W2s[r2 * 36 + 4 * q2]) = w2reg[t];
            }
        }
        if (tid < 8) *reinterpret_cast<f32x4*>(&B1s[4 * tid]) = b1reg;
    };
    load_w(0);
    store_w();
    __syncthreads();

    f32x16 yacc[NTT];
#pragma unroll
    for (int n = 0; n < NTT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) yacc[n][r] = 0.f;

    const float* xp = &Xs[(wave * 32 + l31) * XS + 4 * lhi];   // this lane's pixel row (B operand of GEMM1)
    const float* w1p = &W1s[l31 * XS + 4 * lhi];               // hidden row (A operand of GEMM1)
    const float* w2p = &W2s[l31 * 36 + 4 * lhi];               // output-channel row (B operand of GEMM2)
    constexpr int NCHUNK = 2 * C / MX_HC;
    for (int j = 0; j < NCHUNK; ++j) {
        const bool more = (DBG & 2) ? false : (j + 1 < NCHUNK);   // DBG 2: no weight streaming / barriers
        if (more) load_w(j + 1);
        // GEMM1: Ht = W1c . X^T  (K = C).  One wavefront per SIMD here (LDS-limited), so nothing hides an exposed
        // ds_read latency: fragments are double-buffered by hand and the scheduler is told to interleave the reads
        // of group g+1 with the MFMAs of group g.
        f32x16 h;
#pragma unroll
        for (int r = 0; r < 16; ++r) h[r] = 0.f;
        if (!(DBG & 4)) {   // DBG 4: skip GEMM1
            f32x4 a[2], b[2];
            a[0] = *reinterpret_cast<const f32x4*>(w1p);
            b[0] = *reinterpret_cast<const f32x4*>(xp);
#pragma unroll
            for (int g = 0; g < KG; ++g) {
                if (g + 1 < KG) {
                    a[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(w1p + (g + 1) * 8);
                    b[(g + 1) & 1] = *reinterpret_cast<const f32x4*>(xp + (g + 1) * 8);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) h = __builtin_amdgcn_mfma_f32_32x32x2f32(a[g & 1][e], b[g & 1][e], h, 0, 0, 0);
                if (g + 1 < KG) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                }
            }
        }
        // (interleaving the GELU of k-group g+1 with GEMM2's MFMAs of group g measured 5 % SLOWER: VALU fillers beside
        //  dependent fp32 MFMAs cost more than the stall they remove)
        // bias + GELU in registers; register r holds hidden index 8*(r>>2) + 4*lhi + (r&3)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const f32x4 bv = *reinterpret_cast<const f32x4*>(&B1s[g * 8 + 4 * lhi]);
#pragma unroll
            for (int e = 0; e < 4; ++e) h[g * 4 + e] = (DBG & 1) ? h[g * 4 + e] + bv[e] : rd_gelu(h[g * 4 + e] + bv[e]);
        }
        // GEMM2: Y += H . W2c^T  (K = 32): A = h registers, B = W2s rows (double-buffered like GEMM1)
        if (!(DBG & 8)) {   // DBG 8: skip GEMM2
            f32x4 bf[2][NTT];
#pragma unroll
            for (int n = 0; n < NTT; ++n) bf[0][n] = *reinterpret_cast<const f32x4*>(w2p + n * 32 * 36);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (g + 1 < 4) {
#pragma unroll
                    for (int n = 0; n < NTT; ++n)
                        bf[(g + 1) & 1][n] = *reinterpret_cast<const f32x4*>(w2p + n * 32 * 36 + (g + 1) * 8);
                }
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int n = 0; n < NTT; ++n)
                        yacc[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(h[g * 4 + e], bf[g & 1][n][e], yacc[n], 0, 0, 0);
                if (g + 1 < 4) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, NTT, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 4 * NTT - 2, 0);
                }
            }
        }
        if (!(DBG & 2)) __syncthreads();
        if (more) {
            store_w();
            __syncthreads();
        }
        if (DBG & 8) {
#pragma unroll
            for (int r = 0; r < 16; ++r) yacc[0][r] += h[r];
        }
    }
    // ---- epilogue: + b2 + residual (the gated X tile is still in LDS), strided store
#pragma unroll
    for (int n = 0; n < NTT; ++n) {
        const int co = n * 32 + l31;
        if (co >= C) continue;
        const float bv = p.b2[co];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            const int m = m0 + row;
            if (m >= p.M) continue;
            p.y[(size_t)m * p.yld + co] = yacc[n][r] + bv + Xs[row * XS + co];
        }
    }
}

template <int C>
static size_t mixer_lds_bytes() {
    return (size_t)((MX_BM + MX_HC) * (C + 4) + ((C + 31) / 32 * 32) * 36 + 32) * sizeof(float);
}

bool mixer_fused_supported(int C) { return C == 48 || C == 96 || C == 192; }

template <int C, int DBG>
static void launch_mixer_c(const MixerParams& p, hipStream_t s) {
    const size_t sh = mixer_lds_bytes<C>();
    static unsigned long long lds_ok = 0;
    rd_allow_dynamic_lds((const void*)lc_mixer_kernel<C, DBG>, sh, lds_ok);
    hipLaunchKernelGGL((lc_mixer_kernel<C, DBG>), dim3((p.M + MX_BM - 1) / MX_BM), dim3(256), sh, s, p);
}

// ablation variants for tools/microbench.py (numerically meaningless; they only time what is left)
void launch_mixer_debug(const MixerParams& p, int variant, hipStream_t s) {
    if (p.C != 192) return;
    switch (variant) {
        case 0: launch_mixer_c<192, 0>(p, s); break;
        case 1: launch_mixer_c<192, 1>(p, s); break;
        case 2: launch_mixer_c<192, 2>(p, s); break;
        case 3: launch_mixer_c<192, 3>(p, s); break;
        case 4: launch_mixer_c<192, 4>(p, s); break;
        case 8: launch_mixer_c<192, 8>(p, s); break;
        default: break;
    }
}

void launch_mixer_fused(const MixerParams& p, hipStream_t s) {
    if (p.M <= 0) return;
    switch (p.C) {
        case 48: launch_mixer_c<48, 0>(p, s); break;
        case 96: launch_mixer_c<96, 0>(p, s); break;
        case 192: launch_mixer_c<192, 0>(p, s); break;
        default: break;
    }
}

}  // namespace rd
