// Developer probe (GPU box): (1) do the fp16 matrix cores keep SUBNORMAL inputs (needed by an unscaled-lo split)?
// (2) operand / result register layouts of v_mfma_f32_16x16x32_f16 and v_mfma_f32_32x32x16_f16 as this repo assumes them.
//   hipcc --offload-arch=gfx950 -O2 tools/probe_mfma.hip -o /tmp/probe_mfma && /tmp/probe_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// A [16][32], B [32][16] row-major fp16 in global; D [16][16]
__global__ void k16(const _Float16* A, const _Float16* B, float* D) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[r * 32 + 8 * g + e]; b[e] = B[(8 * g + e) * 16 + r]; }
    f32x4 c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 4; ++i) D[(4 * g + i) * 16 + r] = c[i];
}
// A [32][16], B [16][32]; D [32][32]
__global__ void k32(const _Float16* A, const _Float16* B, float* D) {
    const int l = threadIdx.x, r = l & 31, g = l >> 5;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = A[r * 16 + 8 * g + e]; b[e] = B[(8 * g + e) * 32 + r]; }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * g) * 32 + r] = c[i];
}
template <typename F> static void run(const char* name, int M, int N, int K, F launch, bool sub) {
    std::vector<_Float16> A(M * K), B(K * N);
    for (int i = 0; i < M * K; ++i) A[i] = sub ? (_Float16)(ldexpf((float)((i * 7) % 13 + 1), -24)) : (_Float16)(float)((i * 7) % 13 - 6);
    for (int i = 0; i < K * N; ++i) B[i] = (_Float16)(float)((i * 5) % 11 - 5);
    _Float16 *dA, *dB; float* dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dD, M * N * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    launch(dA, dB, dD);
    std::vector<float> D(M * N);
    hipMemcpy(D.data(), dD, M * N * 4, hipMemcpyDeviceToHost);
    double maxerr = 0, maxref = 0;
    for (int m = 0; m < M; ++m) for (int n = 0; n < N; ++n) {
        double s = 0;
        for (int k = 0; k < K; ++k) s += (double)(float)A[m * K + k] * (double)(float)B[k * N + n];
        maxerr = fmax(maxerr, fabs(s - D[m * N + n])); maxref = fmax(maxref, fabs(s));
    }
    printf("%s %s: max |ref| %.6g  max err %.3g  -> %s\n", name, sub ? "SUBNORMAL-A" : "layout", maxref, maxerr, maxerr <= 1e-6 * maxref ? "OK" : "MISMATCH");
    hipFree(dA); hipFree(dB); hipFree(dD);
}
int main() {
    for (int sub = 0; sub < 2; ++sub) {
        run("mfma_f32_16x16x32_f16", 16, 16, 32, [](const _Float16* a, const _Float16* b, float* d) { hipLaunchKernelGGL(k16, dim3(1), dim3(64), 0, 0, a, b, d); hipDeviceSynchronize(); }, sub);
        run("mfma_f32_32x32x16_f16", 32, 32, 16, [](const _Float16* a, const _Float16* b, float* d) { hipLaunchKernelGGL(k32, dim3(1), dim3(64), 0, 0, a, b, d); hipDeviceSynchronize(); }, sub);
    }
    return 0;
}
