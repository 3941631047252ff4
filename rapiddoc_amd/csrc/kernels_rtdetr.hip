// Architecture-independent operators of an RT-DETR-family layout head - PREPARATION ONLY.
//
// PP-DocLayout's neck / decoder exist in /root/reference as ONNX files that are not part of the offline tree: the graph cannot be
// read, so nothing here is wired into a model and nothing here counts towards the pages/s line.  Three operators any RT-DETR-style head
// needs are nevertheless fully defined by their published definitions, and are built here as C-ABI entries with fp64 tests
// (tests/test_gpu_rtdetr_ops.py), PARITY UNPINNED (no reference output exists to pin them to):
//   * multi-scale deformable-attention sampling (Deformable DETR, Zhu et al. 2020, eq. 3; the I/O contract of the decoder's cross-attention:
//     bilinear samples of P points on L feature levels per head, weighted by softmaxed attention weights) - msdeform_attn_kernel;
//   * top-k over a flattened (queries x classes) score row with a defined tie rule (lower flat index first) - topk_row_kernel: the
//     reference's post-processing receives exactly such a selection (pp_doclayout/main.py:88-139 reads [boxes, box_nums] of 300 rows);
//   * the AIFI transformer encoder layer (multi-head self-attention over the 625 tokens of the coarsest level + FFN, post-norm) is composed on
//     the host side of the C-ABI (api.cpp: rd_encoder_layer) from the engine's existing GEMM / attention / LayerNorm kernels.
#include <cmath>
#include <cstdint>

#include "rd_device.h"

namespace rd {

// ------------------------------------------------------------------------------------------------------------------
// Multi-scale deformable attention.  value [B][S][H][D] (S = sum over levels of h_l * w_l, level l starts at row start[l]),
// shapes int32 [L][2] = (h_l, w_l), loc [B][Q][H][L][P][2] = (x, y) in [0, 1] (normalised to each level's own extent),
// attn [B][Q][H][L][P] (already softmaxed over L * P), out [B][Q][H * D].
// A sample is grid_sample(align_corners = False, padding_mode = "zeros"): pixel centre i sits at (i + 0.5) / size, corners outside the map
// contribute zero.  One workgroup per (query, image), one thread per (head, channel): the four corner reads of a sample are D consecutive
// floats per head (128 bytes at D = 32).
__global__ void __launch_bounds__(1024) msdeform_attn_kernel(const float* __restrict__ value, const int32_t* __restrict__ shapes,
                                                             const int32_t* __restrict__ start, const float* __restrict__ loc,
                                                             const float* __restrict__ attn, float* __restrict__ out, int S, int H, int D, int Q,
                                                             int L, int P) {
    const int q = blockIdx.x, b = blockIdx.y, tid = threadIdx.x;
    if (tid >= H * D) return;
    const int h = tid / D, d = tid - h * D;
    const float* lp = loc + ((((size_t)b * Q + q) * H + h) * L) * P * 2;
    const float* ap = attn + ((((size_t)b * Q + q) * H + h) * L) * P;
    const float* vb = value + (size_t)b * S * H * D + (size_t)h * D + d;
    float acc = 0.f;
    for (int l = 0; l < L; ++l) {
        const int hl = shapes[2 * l], wl = shapes[2 * l + 1];
        const float* vl = vb + (size_t)start[l] * H * D;
        for (int pt = 0; pt < P; ++pt) {
            const float x = lp[(l * P + pt) * 2] * (float)wl - 0.5f, y = lp[(l * P + pt) * 2 + 1] * (float)hl - 0.5f;
            const float w = ap[l * P + pt];
            const float xf = floorf(x), yf = floorf(y);
            const int x0 = (int)xf, y0 = (int)yf;
            const float fx = x - xf, fy = y - yf;
            float s = 0.f;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const int xi = x0 + (c & 1), yi = y0 + (c >> 1);
                const float cw = ((c & 1) ? fx : 1.f - fx) * ((c >> 1) ? fy : 1.f - fy);
                if ((unsigned)xi < (unsigned)wl && (unsigned)yi < (unsigned)hl) s += cw * vl[((size_t)yi * wl + xi) * H * D];
            }
            acc += w * s;
        }
    }
    out[((size_t)b * Q + q) * H * D + tid] = acc;
}

void launch_msdeform_attn(const float* value, const int32_t* shapes, const int32_t* start, const float* loc, const float* attn, float* out, int B,
                          int S, int H, int D, int Q, int L, int P, hipStream_t s) {
    if (B <= 0 || Q <= 0) return;
    const int threads = (H * D + 63) / 64 * 64;
    hipLaunchKernelGGL(msdeform_attn_kernel, dim3(Q, B), dim3(threads), 0, s, value, shapes, start, loc, attn, out, S, H, D, Q, L, P);
}

// ------------------------------------------------------------------------------------------------------------------
// Top-k of a row of n floats, descending; equal values in ascending index order (so the result is a function of the row alone); NaN
// sorts above +inf (torch.topk's convention).  One 1024-thread workgroup per row:
//   key = (order-preserving 32-bit image of the value) << 32 | ~index  - all keys of a row are distinct;
//   eight radix passes of 8 bits (LDS histogram of the elements that still match the prefix) find the k-th largest key exactly;
//   the k elements with key >= it are collected and bitonic-sorted in LDS.
__device__ __forceinline__ unsigned long long topk_key(float v, unsigned idx) {
    unsigned u = __float_as_uint(v);
    u ^= (u >> 31) ? 0xffffffffu : 0x80000000u;
    return ((unsigned long long)u << 32) | (unsigned long long)(0xffffffffu - idx);
}

__global__ void __launch_bounds__(1024) topk_row_kernel(const float* __restrict__ scores, int n, int k, float* __restrict__ out_vals,
                                                        int32_t* __restrict__ out_idx) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sel[1024];
    __shared__ unsigned long long s_prefix;
    __shared__ unsigned s_remaining, s_count;
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* z = scores + (size_t)row * n;
    if (tid == 0) { s_prefix = 0ull; s_remaining = (unsigned)k; s_count = 0u; }
    __syncthreads();
    for (int pass = 0; pass < 8; ++pass) {
        const int shift = 56 - 8 * pass;
        if (tid < 256) hist[tid] = 0u;
        __syncthreads();
        const unsigned long long prefix = s_prefix;
        for (int i = tid; i < n; i += 1024) {
            const unsigned long long key = topk_key(z[i], (unsigned)i);
            if (pass == 0 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&hist[(unsigned)(key >> shift) & 255u], 1u);
        }
        __syncthreads();
        if (tid == 0) {        // the bin that holds the remaining-th largest of the matching elements
            unsigned rem = s_remaining, bin = 255;
            for (;; --bin) {
                if (hist[bin] >= rem) break;
                rem -= hist[bin];
                if (bin == 0) break;
            }
            s_remaining = rem;
            s_prefix = prefix | ((unsigned long long)bin << shift);
        }
        __syncthreads();
    }
    const unsigned long long kth = s_prefix;            // exactly k keys are >= kth
    for (int i = tid; i < 1024; i += 1024) sel[i] = 0ull;
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const unsigned long long key = topk_key(z[i], (unsigned)i);
        if (key >= kth) sel[atomicAdd(&s_count, 1u)] = key;
    }
    __syncthreads();
    // bitonic sort of the 1024 slots, descending (empty slots are key 0: below every real key)
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int j = tid ^ stride;
            if (j > tid) {
                const bool desc = (tid & size) == 0;
                const unsigned long long a = sel[tid], c = sel[j];
                if (desc ? a < c : a > c) { sel[tid] = c; sel[j] = a; }
            }
            __syncthreads();
        }
    }
    if (tid < k) {
        const unsigned long long key = sel[tid];
        const unsigned idx = 0xffffffffu - (unsigned)(key & 0xffffffffull);
        out_idx[(size_t)row * k + tid] = (int32_t)idx;
        out_vals[(size_t)row * k + tid] = z[idx];
    }
}

void launch_topk_rows(const float* scores, int rows, int n, int k, float* out_vals, int32_t* out_idx, hipStream_t s) {
    if (rows <= 0) return;
    hipLaunchKernelGGL(topk_row_kernel, dim3(rows), dim3(1024), 0, s, scores, n, k, out_vals, out_idx);
}

}  // namespace rd
