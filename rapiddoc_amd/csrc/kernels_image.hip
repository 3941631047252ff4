// 8-bit image operations of the page hot path with OpenCV's fixed-point arithmetic (the reference does them with cv2 on
// the host; pixels decide what the networks see, and the strings must be bit-identical):
//   resize_u8_norm_kernel ..... cv2.resize INTER_CUBIC / INTER_LINEAR on uint8 + normalise -> NCHW fp32
//                               (PPPreProcess pp_doclayout/pre_process.py:22-42; rapidocr DetPreProcess, rapid_ocr.py:517-518)
//   line_warp_kernel .......... cv2.warpPerspective(INTER_CUBIC, BORDER_REPLICATE) of every text line of a rec batch into a
//                               packed uint8 scratch (utils/ocr_utils.py:494-536 get_rotate_crop_image)
//   line_resize_norm_kernel ... np.rot90 for tall crops, cv2.resize (linear) to height 48, /255, (x - 0.5) / 0.5, zero right
//                               padding (rapidocr resize_norm_img, called from rapid_ocr.py:436-440)
// The arithmetic restated here (11-bit resize coefficients rounded half-to-even from float32, int32 rows, the two vertical
// rounding rules, 1/32-pixel remap positions with a 15-bit 4x4 weight table whose sum is forced to 2^15) is the public OpenCV
// 4.x algorithm; oracle/cv2_ops.py is its numpy twin and tests/test_gpu_image_ops.py demands bit-equality with it.
// cv2 itself is absent from the build container, so against the real library this is PARITY UNPINNED.
// OpenCV's scalar code rounds every float operation separately: no FMA contraction in this file.
#pragma clang fp contract(off)
#include <cmath>
#include <mutex>
#include <vector>

#include "rd_device.h"

namespace rd {

static inline int img_grid(long total, int cap = 16384) {
    long g = (total + 255) / 256;
    return (int)(g < 1 ? 1 : g > cap ? cap : g);
}

__device__ __forceinline__ void cv_cubic_coeffs(float x, float* c) {   // interpolateCubic, A = -0.75, float32
    const float A = -0.75f;
    c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
    c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
    c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
    c[3] = 1.f - c[0] - c[1] - c[2];
}
// saturate_cast<short>(v): round half to even, clamp
__device__ __forceinline__ int cv_round_short(float v) {
    const int r = __float2int_rn(v);
    return min(max(r, -32768), 32767);
}
// source position of destination index d: fx = (float)((d + 0.5) * scale - 0.5) with scale = 1 / (dst / src) in double
__device__ __forceinline__ void cv_src_pos(int d, double scale, int& s, float& f) {
    const float fx = (float)(((double)d + 0.5) * scale - 0.5);
    s = (int)floorf(fx);
    f = fx - (float)s;
}

// ---- whole-image resize + normalise.  interp 2: INTER_CUBIC, 1: INTER_LINEAR (both on uint8, result uint8, then normalised)
__global__ void __launch_bounds__(256) resize_u8_norm_kernel(PreprocParams p, double scale_x, double scale_y) {
    const long total = (long)p.OH * p.OW;
    p.src += (size_t)blockIdx.y * p.src_stride;      // image blockIdx.y of a batch (strides 0 / grid.y 1 for a single image)
    p.dst += (size_t)blockIdx.y * p.dst_stride;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int ox = (int)(idx % p.OW), oy = (int)(idx / p.OW);
        int sx, sy;
        float fx, fy;
        cv_src_pos(ox, scale_x, sx, fx);
        cv_src_pos(oy, scale_y, sy, fy);
        int out[3];
        if (p.interp == 2) {
            float cx[4], cy[4];
            cv_cubic_coeffs(fx, cx);
            cv_cubic_coeffs(fy, cy);
            int ax[4], ay[4], xi[4], yi[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                ax[k] = cv_round_short(cx[k] * 2048.f);
                ay[k] = cv_round_short(cy[k] * 2048.f);
                xi[k] = min(max(sx - 1 + k, 0), p.W - 1);
                yi[k] = min(max(sy - 1 + k, 0), p.H - 1);
            }
            long acc[3] = {0, 0, 0};
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const uint8_t* row = p.src + (size_t)yi[a] * p.W * 3;
                int h[3] = {0, 0, 0};
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const uint8_t* px = row + (size_t)xi[b] * 3;
                    h[0] += (int)px[0] * ax[b];
                    h[1] += (int)px[1] * ax[b];
                    h[2] += (int)px[2] * ax[b];
                }
#pragma unroll
                for (int c = 0; c < 3; ++c) acc[c] += (long)h[c] * ay[a];
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) out[c] = (int)min(max((acc[c] + (1L << 21)) >> 22, 0L), 255L);
        } else {
            if (sx < 0) { fx = 0.f; sx = 0; }
            if (sx >= p.W - 1) { fx = 0.f; sx = p.W - 1; }
            if (sy < 0) { fy = 0.f; sy = 0; }
            if (sy >= p.H - 1) { fy = 0.f; sy = p.H - 1; }
            const int x1 = min(sx + 1, p.W - 1), y1 = min(sy + 1, p.H - 1);
            const int a0 = cv_round_short((1.f - fx) * 2048.f), a1 = cv_round_short(fx * 2048.f);
            const int b0 = cv_round_short((1.f - fy) * 2048.f), b1 = cv_round_short(fy * 2048.f);
            const uint8_t* r0 = p.src + (size_t)sy * p.W * 3;
            const uint8_t* r1 = p.src + (size_t)y1 * p.W * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int s0 = (int)r0[sx * 3 + c] * a0 + (int)r0[x1 * 3 + c] * a1;
                const int s1 = (int)r1[sx * 3 + c] * a0 + (int)r1[x1 * 3 + c] * a1;
                out[c] = min(max((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2, 0), 255);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int sc = p.swap_rb ? 2 - c : c;
            p.dst[(size_t)c * total + idx] = ((float)out[sc] * p.scale - p.mean[c]) * p.inv_std[c];
        }
    }
}

void launch_preproc_resize_norm(const PreprocParams& p, hipStream_t s) {
    const double sx = 1.0 / ((double)p.OW / (double)p.W), sy = 1.0 / ((double)p.OH / (double)p.H);
    hipLaunchKernelGGL(resize_u8_norm_kernel, dim3(img_grid((long)p.OH * p.OW), p.batch > 1 ? p.batch : 1), dim3(256), 0, s, p, sx, sy);
}

// ---- remap weight table of INTER_CUBIC: [32 * 32][16] int16, block (fy * 32 + fx), taps row-major (y, x)
static std::vector<int16_t> build_cubic_remap_table() {
    auto coeffs = [](float x, float* c) {
        const float A = -0.75f;
        c[0] = ((A * (x + 1.f) - 5.f * A) * (x + 1.f) + 8.f * A) * (x + 1.f) - 4.f * A;
        c[1] = ((A + 2.f) * x - (A + 3.f)) * x * x + 1.f;
        c[2] = ((A + 2.f) * (1.f - x) - (A + 3.f)) * (1.f - x) * (1.f - x) + 1.f;
        c[3] = 1.f - c[0] - c[1] - c[2];
    };
    float t1[32][4];
    for (int i = 0; i < 32; ++i) coeffs((float)i * (1.f / 32.f), t1[i]);
    std::vector<int16_t> tab((size_t)1024 * 16);
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            int it[4][4], sum = 0;
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    const float v = t1[i][a] * t1[j][b];
                    long r = std::lrintf(v * 32768.f);   // round half to even (default rounding mode)
                    r = r < -32768 ? -32768 : r > 32767 ? 32767 : r;
                    it[a][b] = (int)r;
                    sum += (int)r;
                }
            const int diff = sum - 32768;
            if (diff != 0) {   // force the sum: adjust the smallest / largest weight of rows / columns ksize/2 .. ksize/2 + 1
                int mk1 = 2, mk2 = 2, Mk1 = 2, Mk2 = 2;
                for (int k1 = 2; k1 < 4; ++k1)
                    for (int k2 = 2; k2 < 4; ++k2) {
                        if (it[k1][k2] < it[mk1][mk2]) { mk1 = k1; mk2 = k2; }
                        else if (it[k1][k2] > it[Mk1][Mk2]) { Mk1 = k1; Mk2 = k2; }
                    }
                if (diff < 0) it[Mk1][Mk2] -= diff;
                else it[mk1][mk2] -= diff;
            }
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) tab[((size_t)i * 32 + j) * 16 + a * 4 + b] = (int16_t)it[a][b];
        }
    return tab;
}
static const int16_t* cubic_remap_table_dev() {   // one copy per device, built on first use
    static std::mutex mu;
    static const int16_t* dev_tab[64] = {nullptr};
    int dev = 0;
    (void)hipGetDevice(&dev);
    std::lock_guard<std::mutex> lk(mu);
    if (!dev_tab[dev & 63]) {
        const std::vector<int16_t> t = build_cubic_remap_table();
        void* d = nullptr;
        if (hipMalloc(&d, t.size() * 2) != hipSuccess) return nullptr;
        (void)hipMemcpy(d, t.data(), t.size() * 2, hipMemcpyHostToDevice);
        dev_tab[dev & 63] = (const int16_t*)d;
    }
    return dev_tab[dev & 63];
}

// ---- text lines of one rec batch: page -> rectified uint8 crop (packed scratch) -> [n][3][48][out_w_padded] fp32
__global__ void __launch_bounds__(256) line_warp_kernel(LineCropParams p, const int16_t* __restrict__ tab) {
    const LineCropDesc d = p.descs[blockIdx.y];
    const long total = (long)d.crop_w * d.crop_h;
    const uint8_t* src = p.pages + (size_t)d.page * p.page_stride;
    uint8_t* dst = p.scratch + d.scratch_off;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int x = (int)(idx % d.crop_w), y = (int)(idx / d.crop_w);
        double w = d.m[6] * x + d.m[7] * y + d.m[8];
        w = w != 0.0 ? 32.0 / w : 0.0;
        const double fx = fmax(-2147483648.0, fmin(2147483647.0, (d.m[0] * x + d.m[1] * y + d.m[2]) * w));
        const double fy = fmax(-2147483648.0, fmin(2147483647.0, (d.m[3] * x + d.m[4] * y + d.m[5]) * w));
        const int X = __double2int_rn(fx), Y = __double2int_rn(fy);
        const int sx = X >> 5, sy = Y >> 5;
        const int16_t* wt = tab + (size_t)(((Y & 31) << 5) + (X & 31)) * 16;
        int acc[3] = {0, 0, 0};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const int yy = min(max(sy - 1 + a, 0), p.H - 1);
            const uint8_t* row = src + (size_t)yy * p.W * 3;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int xx = min(max(sx - 1 + b, 0), p.W - 1);
                const int wv = wt[a * 4 + b];
                acc[0] += (int)row[xx * 3 + 0] * wv;
                acc[1] += (int)row[xx * 3 + 1] * wv;
                acc[2] += (int)row[xx * 3 + 2] * wv;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[idx * 3 + c] = (uint8_t)min(max((acc[c] + (1 << 14)) >> 15, 0), 255);
    }
}

__global__ void __launch_bounds__(256) line_resize_norm_kernel(LineCropParams p) {
    const int i = blockIdx.y;
    const LineCropDesc d = p.descs[i];
    const long plane = (long)p.OH * p.OWp;
    float* dst = p.dst + (size_t)i * 3 * plane;
    const uint8_t* crop = p.scratch + d.scratch_off;
    // the image the recogniser resizes: the crop, or np.rot90(crop) for tall boxes: rot[r][c] = crop[c][cw - 1 - r]
    const int rw = d.rot90 ? d.crop_h : d.crop_w, rh = d.rot90 ? d.crop_w : d.crop_h;
    const double scale_x = 1.0 / ((double)d.out_w / (double)rw), scale_y = 1.0 / ((double)p.OH / (double)rh);
    auto at = [&](int r, int c, int ch) -> int {
        return d.rot90 ? crop[((size_t)c * d.crop_w + (d.crop_w - 1 - r)) * 3 + ch] : crop[((size_t)r * d.crop_w + c) * 3 + ch];
    };
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < plane; idx += (long)gridDim.x * 256) {
        const int ox = (int)(idx % p.OWp), oy = (int)(idx / p.OWp);
        float out[3] = {0.f, 0.f, 0.f};
        if (ox < d.out_w) {
            int sx, sy;
            float fx, fy;
            cv_src_pos(ox, scale_x, sx, fx);
            cv_src_pos(oy, scale_y, sy, fy);
            if (sx < 0) { fx = 0.f; sx = 0; }
            if (sx >= rw - 1) { fx = 0.f; sx = rw - 1; }
            if (sy < 0) { fy = 0.f; sy = 0; }
            if (sy >= rh - 1) { fy = 0.f; sy = rh - 1; }
            const int x1 = min(sx + 1, rw - 1), y1 = min(sy + 1, rh - 1);
            const int a0 = cv_round_short((1.f - fx) * 2048.f), a1 = cv_round_short(fx * 2048.f);
            const int b0 = cv_round_short((1.f - fy) * 2048.f), b1 = cv_round_short(fy * 2048.f);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int s0 = at(sy, sx, c) * a0 + at(sy, x1, c) * a1;
                const int s1 = at(y1, sx, c) * a0 + at(y1, x1, c) * a1;
                const int v = min(max((((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2, 0), 255);
                out[c] = ((float)v / 255.f - 0.5f) / 0.5f;      // resize_norm_img: /255, -= 0.5, /= 0.5 in float32
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) dst[(size_t)c * plane + idx] = out[p.swap_rb ? 2 - c : c];
    }
}

// stage 1 alone: the rectified uint8 crops of p.n lines into the scratch buffer (p.dst unused)
int launch_line_warp(const LineCropParams& p, hipStream_t s) {
    if (p.n <= 0) return 0;
    const int16_t* tab = cubic_remap_table_dev();
    if (!tab) return 1;
    hipLaunchKernelGGL(line_warp_kernel, dim3(img_grid(p.max_crop_pixels, 256), p.n), dim3(256), 0, s, p, tab);
    return 0;
}
// stage 2 alone: scratch crops -> [n][3][OH][OWp] (p.pages unused)
int launch_line_resize_norm(const LineCropParams& p, hipStream_t s) {
    if (p.n <= 0) return 0;
    hipLaunchKernelGGL(line_resize_norm_kernel, dim3(img_grid((long)p.OH * p.OWp, 64), p.n), dim3(256), 0, s, p);
    return 0;
}
int launch_line_crops(const LineCropParams& p, hipStream_t s) {
    if (launch_line_warp(p, s) != 0) return 1;
    return launch_line_resize_norm(p, s);
}

}  // namespace rd

// ---------------------------------------------------------------------------------------------------------------------
// CTC greedy decode on the device (rapidocr CTCLabelDecode as called from rapid_doc/model/ocr/rapid_ocr.py:444-449):
// per text line collapse repeated indices, drop the blank (0), map the kept indices to the UTF-8 bytes of their dictionary
// entries and average the kept max-probabilities - so that only the finished strings cross PCIe and the host loop over
// [B][T] indices (14 ms per 32-page step in round 1) disappears.
// The confidence reproduces numpy's float32 `np.mean` bit for bit: add.reduce starts from the identity 0 and sums with the
// pairwise scheme of numpy/core/src/umath/loops_utils.h (n < 8: a plain loop; n <= 128: eight strided partial sums
// combined as ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a tail loop; larger n: split at (n/2 rounded down to a multiple of 8)),
// then one float32 division by the count (tests/test_ocr_host.py pins the scheme against np.mean on the CPU).
// Row layout of `out` (stride row_bytes): int32 n_text_bytes, float32 confidence, int32 n_kept, int32 pad, then the text.
// ---------------------------------------------------------------------------------------------------------------------
#pragma clang fp contract(off)
__device__ float np_pairwise_sum_f32(const float* a, int n) {
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res += a[i];
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum_f32(a, n2) + np_pairwise_sum_f32(a + n2, n - n2);
}

// seg == nullptr: line b is idx[b * T .. b * T + T).  seg != nullptr (ragged lines, the batched recogniser tail): line b is the
// seg[2 b + 1] <= T tokens starting at token seg[2 b].  kept_cols (optional, uint16 [lines][T]): the time step of every kept
// character, in order - what rapidocr's CTCLabelDecode hands to get_word_info as `selection` (word boxes, ocr_patch.py:333-389).
__global__ void __launch_bounds__(256) ctc_collapse_kernel(const int32_t* __restrict__ idx, const float* __restrict__ prob, int Tmax,
                                                           const int32_t* __restrict__ seg, const uint8_t* __restrict__ ctab, int max_len,
                                                           int n_classes, uint8_t* __restrict__ out, int row_bytes,
                                                           uint16_t* __restrict__ kept_cols, float* __restrict__ kept_conf) {
    extern __shared__ unsigned char smem[];
    float* kept = reinterpret_cast<float*>(smem);                  // [T] kept probabilities, compacted
    int* scan = reinterpret_cast<int*>(smem + (size_t)Tmax * 4);   // [2][256] scan scratch
    __shared__ int carry_k, carry_b;
    const int line = blockIdx.x, tid = threadIdx.x;
    const size_t first = seg ? (size_t)seg[2 * line] : (size_t)line * Tmax;
    const int T = seg ? min(seg[2 * line + 1], Tmax) : Tmax;
    const int32_t* row = idx + first;
    const float* prow = prob + first;
    uint8_t* orow = out + (size_t)line * row_bytes;
    if (tid == 0) carry_k = carry_b = 0;
    __syncthreads();
    for (int t0 = 0; t0 < T; t0 += 256) {
        const int t = t0 + tid;
        int id = 0, keep = 0, len = 0;
        if (t < T) {
            id = row[t];
            keep = id != 0 && (t == 0 || id != row[t - 1]);
            if (keep && id > 0 && id < n_classes) len = ctab[(size_t)id * (max_len + 1)];
        }
        // block-wide inclusive scans of `keep` and `len` (Hillis-Steele over 256 entries)
        int* sk = scan;
        int* sb = scan + 256;
        sk[tid] = keep;
        sb[tid] = len;
        __syncthreads();
        for (int off = 1; off < 256; off <<= 1) {
            const int vk = tid >= off ? sk[tid - off] : 0, vb = tid >= off ? sb[tid - off] : 0;
            __syncthreads();
            sk[tid] += vk;
            sb[tid] += vb;
            __syncthreads();
        }
        const int pos_k = carry_k + sk[tid] - keep, pos_b = carry_b + sb[tid] - len;
        if (keep) {
            kept[pos_k] = prow[t];
            if (kept_cols) kept_cols[(size_t)line * Tmax + pos_k] = (uint16_t)t;
            if (kept_conf) kept_conf[(size_t)line * Tmax + pos_k] = prow[t];
            const uint8_t* src = ctab + (size_t)id * (max_len + 1) + 1;
            for (int b = 0; b < len; ++b) orow[16 + pos_b + b] = src[b];
        }
        __syncthreads();
        if (tid == 255) {
            carry_k += sk[255];
            carry_b += sb[255];
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int n = carry_k;
        float conf = 0.f;
        if (n > 0) conf = (0.f + np_pairwise_sum_f32(kept, n)) / (float)n;
        reinterpret_cast<int32_t*>(orow)[0] = carry_b;
        reinterpret_cast<float*>(orow)[1] = conf;
        reinterpret_cast<int32_t*>(orow)[2] = n;
        reinterpret_cast<int32_t*>(orow)[3] = 0;
    }
}

namespace rd {
int launch_ctc_collapse(const int32_t* idx, const float* prob, int B, int T, const int32_t* seg, const uint8_t* ctab, int max_len, int n_classes,
                        uint8_t* out, int row_bytes, uint16_t* kept_cols, float* kept_conf, hipStream_t s) {
    if (B <= 0 || T <= 0) return 0;
    if (row_bytes < 16 + T * max_len || T > 65535) return 1;
    const size_t sh = (size_t)T * 4 + 2 * 256 * sizeof(int);
    if (sh > 60000) return 1;
    hipLaunchKernelGGL(ctc_collapse_kernel, dim3(B), dim3(256), sh, s, idx, prob, T, seg, ctab, max_len, n_classes, out, row_bytes, kept_cols, kept_conf);
    return 0;
}
}  // namespace rd

// ---------------------------------------------------------------------------------------------------------------------
// box_score_fast of the DB post-process's candidates (ocr_patch.py:223-241); the rest of the chain is kernels_dbpost.hip.
// ---------------------------------------------------------------------------------------------------------------------
struct DbCand { double box[8], rect[8], w, h; };

__global__ void __launch_bounds__(256) db_scores_kernel(const float* __restrict__ prob, int H, int W, const DbCand* cand, const int32_t* n_cand,
                                                        int max_cand, double* scores) {
    const int i = blockIdx.x, b = blockIdx.y;
    if (i >= n_cand[b]) return;
    const DbCand& c = cand[(size_t)b * max_cand + i];
    const float* pred = prob + (size_t)b * H * W;
    double xmn = 1e300, xmx = -1e300, ymn = 1e300, ymx = -1e300;
    for (int k = 0; k < 4; ++k) {
        xmn = fmin(xmn, c.box[2 * k]); xmx = fmax(xmx, c.box[2 * k]);
        ymn = fmin(ymn, c.box[2 * k + 1]); ymx = fmax(ymx, c.box[2 * k + 1]);
    }
    const int x0 = min(max((int)floor(xmn), 0), W - 1), x1 = min(max((int)ceil(xmx), 0), W - 1);
    const int y0 = min(max((int)floor(ymn), 0), H - 1), y1 = min(max((int)ceil(ymx), 0), H - 1);
    long qx[4], qy[4];
    for (int k = 0; k < 4; ++k) { qx[k] = (long)(c.box[2 * k] - x0); qy[k] = (long)(c.box[2 * k + 1] - y0); }
    long area2 = 0;
    for (int k = 0; k < 4; ++k) area2 += qx[k] * qy[(k + 1) & 3] - qx[(k + 1) & 3] * qy[k];
    const int sgn = area2 >= 0 ? 1 : -1;
    const int bw = x1 - x0 + 1;
    const long total = (long)bw * (y1 - y0 + 1);
    double sum = 0.0;
    long cnt = 0;
    for (long t = threadIdx.x; t < total; t += 256) {
        const long px = t % bw, py = t / bw;
        bool in = true;
        for (int k = 0; k < 4 && in; ++k) {
            const long cr = (qx[(k + 1) & 3] - qx[k]) * (py - qy[k]) - (qy[(k + 1) & 3] - qy[k]) * (px - qx[k]);
            in = (cr * sgn) >= 0;
        }
        if (in) { sum += (double)pred[(size_t)(y0 + py) * W + x0 + px]; ++cnt; }
    }
    __shared__ double ssum[256];
    __shared__ long scnt[256];
    ssum[threadIdx.x] = sum;
    scnt[threadIdx.x] = cnt;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) { ssum[threadIdx.x] += ssum[threadIdx.x + off]; scnt[threadIdx.x] += scnt[threadIdx.x + off]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) scores[(size_t)b * max_cand + i] = scnt[0] ? ssum[0] / (double)scnt[0] : 0.0;
}

namespace rd {
int launch_db_scores(const float* prob, int B, int H, int W, const void* cand, const int32_t* n_cand, int max_cand, double* scores, hipStream_t s) {
    if (B <= 0 || max_cand <= 0) return 0;
    hipLaunchKernelGGL(db_scores_kernel, dim3(max_cand, B), dim3(256), 0, s, prob, H, W, reinterpret_cast<const DbCand*>(cand), n_cand, max_cand, scores);
    return 0;
}
}  // namespace rd
