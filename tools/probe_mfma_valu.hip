// Developer probe (GPU box): do VALU instructions of one wavefront run under the MFMAs of another wavefront on the same SIMD
// (and under the MFMAs of the same wavefront when interleaved in program order)?
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_valu.hip -o /tmp/pmv && /tmp/pmv
// Workgroup = 8 wavefronts (2 per SIMD: w and w + 4).  mode bits: 1 = wavefronts 0-3 run MFMAs, 2 = wavefronts 4-7 run VALU fma
// chains, 4 = wavefronts 4-7 run v_exp instead, 8 = every wavefront runs 1 MFMA + 7 VALU interleaved in program order.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void __launch_bounds__(512) k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (e + 1)); }
    f32x16 acc0, acc1;
    for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
    float v[8];
    for (int j = 0; j < 8; ++j) v[j] = 0.5f + 0.01f * j + 0.001f * threadIdx.x;
    if (mode & 16) { if (wave >= 4) __builtin_amdgcn_s_setprio(2); }      // VALU wavefronts above the MFMA wavefronts
    if (mode & 32) { if (wave < 4) __builtin_amdgcn_s_setprio(2); }       // MFMA wavefronts above
    if (mode & 64) {
        // same wavefront: 1 MFMA + NV VALU, NV = mode >> 8
        const int nv = mode >> 8;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                if (nv == 2) { v[0] = fmaf(v[0], 1.0001f, 0.0001f); v[1] = fmaf(v[1], 1.0001f, 0.0001f); }
                if (nv == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = fmaf(v[j], 1.0001f, 0.0001f);
                }
                if (nv == 6) {
#pragma unroll
                    for (int j = 0; j < 6; ++j) v[j] = fmaf(v[j], 1.0001f, 0.0001f);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (mode & 8) {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = fmaf(v[j], 1.0001f, 0.0001f);
                __builtin_amdgcn_sched_barrier(0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
#pragma unroll
                for (int j = 0; j < 7; ++j) v[j] = fmaf(v[j], 1.0001f, 0.0001f);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (wave < 4) {
        if (mode & 1)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc1, 0, 0, 0);
                }
            }
    } else {
        if (mode & 2)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 14; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = fmaf(v[j], 1.0001f, 0.0001f);
            }
        if (mode & 4)
            for (int i = 0; i < iters; ++i) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = __builtin_amdgcn_exp2f(v[j]) * 0.5f;
            }
    }
    float s = 0.f;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    for (int j = 0; j < 8; ++j) s += v[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float run(int mode, int iters) {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, 10, mode);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, iters, mode);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}
int main() {
    const int iters = 4000;
    printf("per iteration: 16 MFMA 32x32x16 (wavefronts 0-3), 112 v_fma or 32 (v_exp + v_mul) (wavefronts 4-7)\n");
    const char* names[] = {"", "MFMA only", "fma only", "MFMA + fma", "exp only", "MFMA + exp"};
    for (int mode : {1, 2, 3, 4, 5}) {
        const float ms = run(mode, iters);
        printf("mode %d %-12s: %8.3f ms  = %6.1f ns per iteration\n", mode, names[mode], ms, ms * 1e6 / iters);
    }
    for (int mode : {3 | 16, 3 | 32, 5 | 16}) {
        const float ms = run(mode, iters);
        printf("mode %d (MFMA + %s, %s wavefronts at s_setprio 2): %8.3f ms = %6.1f ns per iteration\n", mode, (mode & 4) ? "exp" : "fma",
               (mode & 16) ? "VALU" : "MFMA", ms, ms * 1e6 / iters);
    }
    for (int nv : {0, 2, 4, 6}) {
        const float ms = run(64 | (nv << 8), iters);
        printf("same wavefront, 16 x (MFMA + %d fma), 8 wavefronts: %8.3f ms = %6.1f ns per iteration (per SIMD: 32 MFMA = 527 ns alone)\n", nv, ms, ms * 1e6 / iters);
    }
    const float ms = run(8, iters);
    printf("mode 8 same wavefront, 16 x (MFMA + 7 fma), 8 wavefronts: %8.3f ms = %6.1f ns per iteration (2 wavefronts per SIMD -> 32 MFMA + 224 fma per SIMD)\n",
           ms, ms * 1e6 / iters);
    return 0;
}
