// Developer probe, round 4: does VALU work of a wavefront issue under the MFMAs of ANOTHER wavefront of the
// same SIMD, and under its OWN MFMAs?  Round 2's probe (probe_mfma_valu.hip) answered "no - the times add" from wall clock alone,
// against MI355X_MICROARCH.md:59-61 ("separate pipes").  This one settles it with (i) >= 4 independent accumulators per MFMA
// wavefront, (ii) s_setprio variants, (iii) one kernel NAME per arm, so that `rocprofv3 --pmc` gives SQ_VALU_MFMA_BUSY_CYCLES /
// SQ_ACTIVE_INST_VALU / SQ_WAIT_INST_ANY / SQ_BUSY_CYCLES per arm (tools/probe_mfma_valu2.sh), and (iv) packed VALU.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_valu2.hip -o /tmp/pmv2 && /tmp/pmv2
// Workgroup = 8 wavefronts = 2 per SIMD (w and w + 4).  Per iteration: the MFMA stream is 16 v_mfma_f32_32x32x16_f16 on NACC
// accumulators round robin; the VALU stream is NV v_fma_f32 (or NV / 2 v_pk_fma_f32, or NV / 4 v_exp_f32 + mul) on 8 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Arm : int {
    MFMA_ONLY_1ACC = 0, MFMA_ONLY_4ACC, VALU_ONLY_FMA, VALU_ONLY_PK, VALU_ONLY_EXP,
    BOTH_FMA, BOTH_FMA_PRIO_MFMA, BOTH_FMA_PRIO_VALU, BOTH_PK, BOTH_EXP,
    BOTH_FMA_HALF,                // 56 fma per 16 MFMA in the partner (3.5 per MFMA)
    SAME_WAVE_K2, SAME_WAVE_K4, SAME_WAVE_K6, SAME_WAVE_K4_TWO_WAVES, SAME_WAVE_K4_PRIO, N_ARMS
};
static const char* kNames[N_ARMS] = {
    "MFMA only, 1 accumulator, 1 wave/SIMD", "MFMA only, 4 accumulators, 1 wave/SIMD", "VALU only: 112 v_fma, 1 wave/SIMD",
    "VALU only: 56 v_pk_fma (=112 fma), 1 wave/SIMD", "VALU only: 28 v_exp + 28 v_mul, 1 wave/SIMD",
    "MFMA wave + v_fma wave per SIMD", "  same, MFMA waves at s_setprio 1", "  same, VALU waves at s_setprio 1",
    "MFMA wave + v_pk_fma wave per SIMD", "MFMA wave + v_exp wave per SIMD", "MFMA wave + 56-v_fma wave per SIMD",
    "ONE wave/SIMD: 16 x (MFMA + 2 v_fma)", "ONE wave/SIMD: 16 x (MFMA + 4 v_fma)", "ONE wave/SIMD: 16 x (MFMA + 6 v_fma)",
    "TWO waves/SIMD, each 16 x (MFMA + 4 v_fma)", "ONE wave/SIMD: 16 x (setprio1 MFMA setprio0 + 4 v_fma)"};

#define MFMA(acc) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0)
// inline asm: hipcc -O3 SLP-packs adjacent scalar fma chains into v_pk_fma_f32 (round 2's "112 v_fma" were 56 v_pk_fma)
#define FMA(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(k1), "v"(k2))
#define PKFMA(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pk1), "v"(pk2))

template <int ARM>
__global__ void __launch_bounds__(512) probe(float* out, int iters) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (e + 1)); }
    f32x16 c0, c1, c2, c3;
    for (int r = 0; r < 16; ++r) c0[r] = c1[r] = c2[r] = c3[r] = 0.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.01f * j + 0.001f * threadIdx.x;
    f32x2 pv[4];
    for (int j = 0; j < 4; ++j) pv[j] = f32x2{v[2 * j], v[2 * j + 1]};
    const f32x2 pk1 = {1.0001f, 1.0001f}, pk2 = {0.0001f, 0.0001f};
    const float k1 = 1.0001f + 0.f * threadIdx.x, k2 = 0.0001f;
    constexpr bool one_wave = ARM == MFMA_ONLY_1ACC || ARM == MFMA_ONLY_4ACC || ARM == VALU_ONLY_FMA || ARM == VALU_ONLY_PK || ARM == VALU_ONLY_EXP ||
                              ARM == SAME_WAVE_K2 || ARM == SAME_WAVE_K4 || ARM == SAME_WAVE_K6 || ARM == SAME_WAVE_K4_PRIO;
    if (one_wave && wave >= 4) { out[blockIdx.x * 512 + threadIdx.x] = 0.f; return; }
    constexpr bool same = ARM >= SAME_WAVE_K2;
    if constexpr (same) {
        constexpr int K = ARM == SAME_WAVE_K2 ? 2 : ARM == SAME_WAVE_K6 ? 6 : 4;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#define STEP(acc)                                                                                       \
    if constexpr (ARM == SAME_WAVE_K4_PRIO) __builtin_amdgcn_s_setprio(1);                                \
    MFMA(acc);                                                                                          \
    if constexpr (ARM == SAME_WAVE_K4_PRIO) __builtin_amdgcn_s_setprio(0);                                \
    _Pragma("unroll") for (int j = 0; j < K; ++j) FMA(v[(j + 2 * u) & 7]);                               \
    __builtin_amdgcn_sched_barrier(0);
                STEP(c0) STEP(c1) STEP(c2) STEP(c3)
#undef STEP
            }
        }
    } else {
        constexpr bool mfma_only = ARM == MFMA_ONLY_1ACC || ARM == MFMA_ONLY_4ACC;
        constexpr bool both = ARM >= BOTH_FMA;
        const bool do_mfma = mfma_only || (both && wave < 4);      // (one-wave arms: wavefronts 4-7 have left)
        if (ARM == BOTH_FMA_PRIO_MFMA && do_mfma) __builtin_amdgcn_s_setprio(1);
        if (ARM == BOTH_FMA_PRIO_VALU && !do_mfma) __builtin_amdgcn_s_setprio(1);
        if (do_mfma) {
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if constexpr (ARM == MFMA_ONLY_1ACC) { MFMA(c0); MFMA(c0); MFMA(c0); MFMA(c0); }
                    else { MFMA(c0); MFMA(c1); MFMA(c2); MFMA(c3); }
                }
            }
        } else {
            if constexpr (ARM == VALU_ONLY_PK || ARM == BOTH_PK) {
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 14; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) PKFMA(pv[j]);
                }
            } else if constexpr (ARM == VALU_ONLY_EXP || ARM == BOTH_EXP) {
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < 7; ++u)
#pragma unroll
                        for (int j = 0; j < 4; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.5f;
                }
            } else {
                constexpr int REP = ARM == BOTH_FMA_HALF ? 7 : 14;
                for (int i = 0; i < iters; ++i) {
#pragma unroll
                    for (int u = 0; u < REP; ++u)
#pragma unroll
                        for (int j = 0; j < 8; ++j) FMA(v[j]);
                }
            }
        }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r] + c2[r] + c3[r];
    for (int j = 0; j < 8; ++j) s += v[j];
    for (int j = 0; j < 4; ++j) s += pv[j][0] + pv[j][1];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int ARM>
static float run_arm(float* out, int iters, int n_cu) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<ARM>, dim3(n_cu), dim3(512), 0, 0, out, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(probe<ARM>, dim3(n_cu), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int ARM>
static void run_all(float* out, int iters, int n_cu, float* ns) {
    if constexpr (ARM < N_ARMS) {
        ns[ARM] = run_arm<ARM>(out, iters, n_cu) * 1e6f / iters;
        run_all<ARM + 1>(out, iters, n_cu, ns);
    }
}
int main() {
    int n_cu = 256;
    (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0);
    float* out;
    (void)hipMalloc(&out, (size_t)n_cu * 512 * 4);
    const int iters = 4000;
    float ns[N_ARMS];
    run_all<0>(out, iters, n_cu, ns);
    printf("per iteration and SIMD: 16 v_mfma_f32_32x32x16_f16; partner / interleaved VALU as named.  %d CUs, %d iterations\n", n_cu, iters);
    for (int a = 0; a < N_ARMS; ++a) printf("arm %2d  %-58s %8.1f ns/iter\n", a, kNames[a], ns[a]);
    auto overlap = [&](int both, int m, int v) { return (ns[m] + ns[v] - ns[both]) / (ns[m] < ns[v] ? ns[m] : ns[v]); };
    printf("hidden share of the shorter stream (1 = fully under the other, 0 = times add):\n");
    printf("  two waves, v_fma      : %.2f   (MFMA prio %.2f, VALU prio %.2f)\n", overlap(BOTH_FMA, MFMA_ONLY_4ACC, VALU_ONLY_FMA),
           overlap(BOTH_FMA_PRIO_MFMA, MFMA_ONLY_4ACC, VALU_ONLY_FMA), overlap(BOTH_FMA_PRIO_VALU, MFMA_ONLY_4ACC, VALU_ONLY_FMA));
    printf("  two waves, v_pk_fma   : %.2f\n", overlap(BOTH_PK, MFMA_ONLY_4ACC, VALU_ONLY_PK));
    printf("  two waves, v_exp      : %.2f\n", overlap(BOTH_EXP, MFMA_ONLY_4ACC, VALU_ONLY_EXP));
    printf("  two waves, 56 v_fma   : both %.1f vs MFMA alone %.1f + VALU alone %.1f\n", ns[BOTH_FMA_HALF], ns[MFMA_ONLY_4ACC], ns[VALU_ONLY_FMA] / 2);
    printf("  one wave, MFMA + k fma: k=2 %.1f  k=4 %.1f  k=6 %.1f ns  (MFMA alone %.1f; a v_fma alone %.2f ns)\n", ns[SAME_WAVE_K2], ns[SAME_WAVE_K4],
           ns[SAME_WAVE_K6], ns[MFMA_ONLY_4ACC], ns[VALU_ONLY_FMA] / 112);
    (void)hipFree(out);
    return 0;
}
