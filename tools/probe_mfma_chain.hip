// Developer probe (GPU box): issue rate of v_mfma_f32_16x16x32_f16 / v_mfma_f32_32x32x16_f16 as a function of the distance
// between two MFMAs that accumulate into the same registers (1 = back to back on one accumulator, 2 = two alternating
// accumulators, ...), one and two wavefronts per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_mfma_chain.hip -o /tmp/probe_chain && /tmp/probe_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int D, int BIG>
__global__ void __launch_bounds__(512) chain(float* out, int iters) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (threadIdx.x + e)); b[e] = (_Float16)(0.002f * (e + 1)); }
    if constexpr (BIG) {
        f32x16 acc[D];
        for (int d = 0; d < D; ++d) for (int r = 0; r < 16; ++r) acc[d][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 24 / D; ++u)
#pragma unroll
                for (int d = 0; d < D; ++d) acc[d] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[d], 0, 0, 0);
        }
        float s = 0.f;
        for (int d = 0; d < D; ++d) for (int r = 0; r < 16; ++r) s += acc[d][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x4 acc[D];
        for (int d = 0; d < D; ++d) acc[d] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 24 / D; ++u)
#pragma unroll
                for (int d = 0; d < D; ++d) acc[d] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[d], 0, 0, 0);
        }
        float s = 0.f;
        for (int d = 0; d < D; ++d) for (int r = 0; r < 4; ++r) s += acc[d][r];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

template <int D, int BIG>
static void run(int threads) {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((chain<D, BIG>), dim3(256), dim3(threads), 0, 0, out, 10);
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((chain<D, BIG>), dim3(256), dim3(threads), 0, 0, out, iters);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double n_per_simd = (double)iters * 24 * (threads / 256);      // MFMAs per SIMD (waves per SIMD = threads / 256)
    const double ns_per_mfma = ms * 1e6 / n_per_simd;
    const double flops = 256.0 * 4 * n_per_simd * (BIG ? 32768.0 : 16384.0);
    printf("%s distance %d, %d wave(s)/SIMD: %6.2f ns per MFMA per SIMD (%.1f cycles at 2.4 GHz)  %.0f TFLOP/s\n", BIG ? "32x32x16" : "16x16x32", D,
           threads / 256, ns_per_mfma, ns_per_mfma * 2.4, flops / (ms * 1e-3) / 1e12);
    (void)hipFree(out);
}
int main() {
    run<1, 0>(256); run<2, 0>(256); run<3, 0>(256); run<4, 0>(256); run<6, 0>(256); run<8, 0>(256);
    run<1, 0>(512); run<2, 0>(512); run<4, 0>(512);
    run<1, 1>(256); run<2, 1>(256); run<3, 1>(256); run<4, 1>(256); run<6, 1>(256);
    run<1, 1>(512); run<2, 1>(512);
    return 0;
}
