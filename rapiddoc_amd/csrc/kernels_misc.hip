// HBM-bound kernels of the page-inference hot path: depthwise stencils, pooling, squeeze-excite,
// nearest upsample (FPN), LayerNorm, the small LightSVTR attention, CTC row statistics, layout
// conversion at the C-ABI boundary and the image resize+normalise pre-processing.
// Everything is NHWC fp32 with the channel dimension innermost => 16-byte coalesced accesses.
#include <cstdlib>
#include <string>

#include "rd_device.h"

namespace rd {


__device__ __forceinline__ f32x4 act4(f32x4 v, int act) {
    f32x4 r = {rd_act(v[0], act), rd_act(v[1], act), rd_act(v[2], act), rd_act(v[3], act)};
    return r;
}

static inline int grid_for(long total, int block = 256, int cap = 16384) {
    long g = (total + block - 1) / block;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// --------------------------------------------------------------------------------------------------
// Depthwise KHxKW conv (reference: rec_lcnetv4.py:187-206 token conv, db_fpn.py:315-323 7x7,
// rec_pphgnetv2.py:945-953 light-block k5, necks/rnn.py:343 local 1x7).  One thread = one output pixel
// x 4 channels; neighbouring threads walk the channel dimension so every tap is a coalesced 16-B load
// and the taps of neighbouring pixels hit in L1/L2.
// --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) dwconv_kernel(DwParams p) {
    const int c4n = p.C >> 2;
    const long total = (long)p.N * p.OH * p.OW * c4n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) << 2;
        long pix = idx / c4n;
        const int ow = pix % p.OW;
        const long t = pix / p.OW;
        const int oh = t % p.OH;
        const int b = t / p.OH;
        f32x4 acc = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : f32x4{0.f, 0.f, 0.f, 0.f};
        const float* xb = p.x + (size_t)b * p.H * p.W * p.xld + c;
        const int ih0 = oh * p.SH - p.PT, iw0 = ow * p.SW - p.PL;
        int w_lo = 0, w_hi = p.W;                 // valid input columns: the row, or this token's text line (ragged rows)
        if (p.line_w) w_hi = min(p.W, p.line_w[b * p.line_w_stride]);
        if (p.tokinfo) {
            const int ti = p.tokinfo[pix];
            w_lo = ow - (ti & 0xffff);
            w_hi = w_lo + (ti >> 16);
        }
        for (int kh = 0; kh < p.KH; ++kh) {
            const int ih = ih0 + kh;
            if ((unsigned)ih >= (unsigned)p.H) continue;
            for (int kw = 0; kw < p.KW; ++kw) {
                const int iw = iw0 + kw;
                if (iw < w_lo || iw >= w_hi) continue;
                const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + ((size_t)ih * p.W + iw) * p.xld);
                const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + (size_t)(kh * p.KW + kw) * p.C + c);
                acc += xv * wv;
            }
        }
        acc = act4(acc, p.act);
        if (p.res) acc += *reinterpret_cast<const f32x4*>(p.res + (size_t)pix * p.rld + c);
        *reinterpret_cast<f32x4*>(p.y + (size_t)pix * p.yld + c) = acc;
    }
}
// Register-tiled variant: one thread = TW consecutive output pixels of one row x 4 channels.  The input row segment
// ((TW-1)*SW + KW pixels) is loaded once per kernel row and reused by every tap: 3.75 loads per output for 3x3/s1
// at TW = 8 instead of 9, which moves the kernel from L1-request-bound towards the HBM roofline.  Optionally emits
// deterministic per-block partial sums of its OUTPUT for the squeeze-excite global average pool that follows
// (rec_lcnetv4.py:228-229): partial[n][chunk][c].
template <int KH, int KW, int SW, int TW, int DBG = 0>
__global__ void __launch_bounds__(256) dwconv_tiled_kernel(DwParams p, int c4n, int groups_w, int groups, int gpb) {
    constexpr int NCOL = (TW - 1) * SW + KW;
    constexpr int dbg = DBG;           // developer (RD_DW_DBG, 3x3 only): 1 no stores, 2 no loads, 4 no boundary masks - timing only
    __shared__ f32x4 red[256];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lanes_p = blockDim.x / c4n;           // pixel-group lanes per block
    const int c4 = threadIdx.x % c4n, pl = threadIdx.x / c4n;
    const int c = c4 << 2;
    const float* xb = p.x + (size_t)n * p.H * p.W * p.xld + c;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : zero4;
    f32x4 gsum = zero4;
    // the image's own width under a line table (rd_kernels.h LineTab): columns beyond it are padding to the loads and to the SE sums
    const int wlim = p.line_w ? min(p.W, p.line_w[n * p.line_w_stride]) : p.W;
    const int g_end = min(groups, (chunk + 1) * gpb);
    for (int g = chunk * gpb + pl; g < g_end; g += lanes_p) {
        // column-major over (row, w-group): a block walks DOWN a strip of columns, so the KH-1 input rows shared by
        // vertically adjacent outputs are re-read from this CU's L1/L2 instead of by another XCD from HBM
        const int owg = g / p.OH, oh = g - owg * p.OH;
        const int ow0 = owg * TW;
        f32x4 acc[TW];
#pragma unroll
        for (int t = 0; t < TW; ++t) acc[t] = bias;
        const int ih0 = oh * p.SH - p.PT, iw0 = ow0 * SW - p.PL;
        // Ablation at 101 376 x 192 (RD_DW_DBG): arithmetic alone 19 us, + loads 15, + stores 9 = the kernel's 43 - the phases add at three
        // wavefronts per SIMD.  Requesting all three rows before the first is used (120 VGPRs of rows: two wavefronts per SIMD)
        // measured 54-61 us, four outputs per thread with all rows up front 54, forcing 128 VGPRs spills: the row-by-row form stays.
#pragma unroll
        for (int kh = 0; kh < KH; ++kh) {
            const int ih = ih0 + kh;
            if ((unsigned)ih >= (unsigned)p.H) continue;
            const float* xr = xb + (size_t)ih * p.W * p.xld;
            f32x4 row[NCOL];
            // column loads are unconditional from a clamped address and masked afterwards (a predicated load costs one
            // s_waitcnt per load); whole out-of-range ROWS are still skipped - with the short feature maps of the
            // recogniser a third of the rows are padding, and loading them measured slower
#pragma unroll
            for (int j = 0; j < NCOL; ++j) {
                if constexpr ((dbg & 2) != 0) row[j] = bias;
                else row[j] = *reinterpret_cast<const f32x4*>(xr + (size_t)min(max(iw0 + j, 0), p.W - 1) * p.xld);
            }
            if constexpr (!(dbg & 4)) {
#pragma unroll
                for (int j = 0; j < NCOL; ++j) row[j] = ((unsigned)(iw0 + j) < (unsigned)wlim) ? row[j] : zero4;
            }
#pragma unroll
            for (int kw = 0; kw < KW; ++kw) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(p.w + (size_t)(kh * KW + kw) * p.C + c);
#pragma unroll
                for (int t = 0; t < TW; ++t) acc[t] += row[t * SW + kw] * wv;
            }
        }
        const size_t pix0 = ((size_t)n * p.OH + oh) * p.OW + ow0;
#pragma unroll
        for (int t = 0; t < TW; ++t) {
            if (ow0 + t < p.OW) {
                f32x4 v = act4(acc[t], p.act);
                if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (pix0 + t) * p.rld + c);
                if (!(dbg & 1)) *reinterpret_cast<f32x4*>(p.y + (pix0 + t) * p.yld + c) = v;
                if (ow0 + t < wlim) gsum += v;
            }
        }
    }
    if (p.gap_partial) {
        red[threadIdx.x] = gsum;
        __syncthreads();
        if (pl == 0) {
            for (int r = 1; r < lanes_p; ++r) gsum += red[r * c4n + c4];
            *reinterpret_cast<f32x4*>(p.gap_partial + ((size_t)n * gridDim.x + chunk) * p.C + c) = gsum;
        }
    }
}

// Column-strip variant for 3x3 / stride 1 (the depthwise conv of every PPLCNetV4 block, rec_lcnetv4.py:187-206): one thread =
// TW consecutive output columns x 4 channels x ALL output rows.  It walks down its strip with a three-row register window, so
// every input row is loaded ONCE (1.5 loads per output at TW = 4 instead of 3.75 for the row-tiled kernel above, whose three
// kernel rows re-read their input from L1 / L2): round 1 measured the row-tiled kernel at 3.2 TB/s = 0.40 of HBM with the
// L1 request rate as the suspected limiter.  MEASURED (round 2, bench.py): 17.6 ms per step against 16.5 for the row-tiled kernel
// (TW = 8 without prefetch: 20.3) - fewer, longer threads hide latency worse than the L1 re-reads cost; kept behind RD_DW_COL=1.
// Same optional squeeze-excite partial sums.
template <int TW>
__global__ void __launch_bounds__(256) dwconv3x3_col_kernel(DwParams p, int c4n, int groups_w, int gpb) {
    constexpr int NCOL = TW + 2;
    __shared__ f32x4 red[256];
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int lanes_p = blockDim.x / c4n;
    const int c4 = threadIdx.x % c4n, pl = threadIdx.x / c4n;
    const int c = c4 << 2;
    const float* xb = p.x + (size_t)n * p.H * p.W * p.xld + c;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    const f32x4 bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : zero4;
    f32x4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(p.w + (size_t)k * p.C + c);
    f32x4 gsum = zero4;
    const int wlim = p.line_w ? min(p.W, p.line_w[n * p.line_w_stride]) : p.W;
    const int g_end = min(groups_w, (chunk + 1) * gpb);
    for (int g = chunk * gpb + pl; g < g_end; g += lanes_p) {
        const int ow0 = g * TW, iw0 = ow0 - 1;
        size_t coff[NCOL];
        bool cok[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            coff[j] = (size_t)min(max(iw0 + j, 0), p.W - 1) * p.xld;      // unconditional loads from clamped columns, masked after
            cok[j] = (unsigned)(iw0 + j) < (unsigned)wlim;
        }
        auto load_row = [&](int ih, f32x4* row) {
            const float* xr = xb + (size_t)ih * p.W * p.xld;
#pragma unroll
            for (int j = 0; j < NCOL; ++j) row[j] = *reinterpret_cast<const f32x4*>(xr + coff[j]);
#pragma unroll
            for (int j = 0; j < NCOL; ++j) row[j] = cok[j] ? row[j] : zero4;
        };
        // four-row window: r0..r2 feed output row oh while r3 = input row oh + 2 is already in flight
        f32x4 r0[NCOL], r1[NCOL], r2[NCOL], r3[NCOL];
#pragma unroll
        for (int j = 0; j < NCOL; ++j) r0[j] = r2[j] = r3[j] = zero4;           // row -1: padding
        load_row(0, r1);
        if (p.H > 1) load_row(1, r2);
        for (int oh = 0; oh < p.OH; ++oh) {
            if (oh + 2 < p.H) load_row(oh + 2, r3);
            else {
#pragma unroll
                for (int j = 0; j < NCOL; ++j) r3[j] = zero4;
            }
            const size_t pix0 = ((size_t)n * p.OH + oh) * p.OW + ow0;
#pragma unroll
            for (int t = 0; t < TW; ++t) {
                f32x4 a = bias;
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) a += r0[t + kw] * wv[kw];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) a += r1[t + kw] * wv[3 + kw];
#pragma unroll
                for (int kw = 0; kw < 3; ++kw) a += r2[t + kw] * wv[6 + kw];
                if (ow0 + t < p.OW) {
                    f32x4 v = act4(a, p.act);
                    if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + (pix0 + t) * p.rld + c);
                    *reinterpret_cast<f32x4*>(p.y + (pix0 + t) * p.yld + c) = v;
                    if (ow0 + t < wlim) gsum += v;
                }
            }
#pragma unroll
            for (int j = 0; j < NCOL; ++j) { r0[j] = r1[j]; r1[j] = r2[j]; r2[j] = r3[j]; }
        }
    }
    if (p.gap_partial) {
        red[threadIdx.x] = gsum;
        __syncthreads();
        if (pl == 0) {
            for (int r = 1; r < lanes_p; ++r) gsum += red[r * c4n + c4];
            *reinterpret_cast<f32x4*>(p.gap_partial + ((size_t)n * gridDim.x + chunk) * p.C + c) = gsum;
        }
    }
}
// column-strip geometry: groups are the W / TW column strips of one image
static inline bool dw_col_applies(const DwParams& p) {
    static const bool off = std::getenv("RD_DW_COL") && std::string(std::getenv("RD_DW_COL")) == "0";
    static const bool on = std::getenv("RD_DW_COL") && std::string(std::getenv("RD_DW_COL")) == "1";   // measured 17.6 vs 16.5 ms per step: OFF by default
    return on && !off && p.KH == 3 && p.KW == 3 && p.SH == 1 && p.SW == 1 && p.PT == 1 && p.PL == 1 && p.OH == p.H && p.OW == p.W &&
           p.C % 4 == 0 && (p.C >> 2) <= 256;
}
static inline void dw_col_geom(const DwParams& p, int& c4n, int& threads, int& groups_w, int& gpb, int& chunks) {
    constexpr int TW = 4;
    c4n = p.C >> 2;
    threads = (256 / c4n) * c4n;
    groups_w = (p.OW + TW - 1) / TW;
    gpb = threads / c4n;             // one strip per pixel lane: as many blocks as the geometry allows
    chunks = (groups_w + gpb - 1) / gpb;
}

// chunk geometry shared by the launcher and by the planner (which sizes the partial-sum buffer)
static inline void dw_tiled_geom(const DwParams& p, int TW, int& c4n, int& threads, int& groups_w, int& groups, int& gpb,
                                 int& chunks) {
    c4n = p.C >> 2;
    threads = (256 / c4n) * c4n;
    groups_w = (p.OW + TW - 1) / TW;
    groups = p.OH * groups_w;
    const int lanes_p = threads / c4n;
    // groups per pixel lane: 4 on the large maps (fewer, longer blocks), down to 1 when that would leave the chip under-filled:
    // the recogniser's 6 x 68 x 192 maps gave 192 blocks of 4 wavefronts (under one wavefront per SIMD) with a fixed 4, and the
    // kernel ran latency-bound at 2.3 TB/s
    static const int k_env = std::getenv("RD_DW_GPL") ? atoi(std::getenv("RD_DW_GPL")) : 0;
    int k = 4;
    while (k > 1 && (long)((groups + lanes_p * k - 1) / (lanes_p * k)) * p.N < 4096) --k;
    if (k_env > 0) k = k_env;
    gpb = lanes_p * k;
    // ... but not so many blocks that the SE partial buffer explodes
    while ((long)((groups + gpb - 1) / gpb) * p.N > 8192) gpb += lanes_p;
    chunks = (groups + gpb - 1) / gpb;
}
static inline int dw_env_tw(const char* name, int dflt) {
    const char* e = std::getenv(name);
    const int v = e ? atoi(e) : 0;
    return (v == 4 || v == 8) ? v : dflt;
}
static inline int dw3_tw() { static const int v = dw_env_tw("RD_DW3_TW", 8); return v; }
static inline int dw5_tw() { static const int v = dw_env_tw("RD_DW5_TW", 8); return v; }
static inline int dw7_tw() {
    static const int v = [] { const char* e = std::getenv("RD_DW7_TW"); return e && atoi(e) == 8 ? 8 : 4; }();
    return v;
}
static inline int dw_tiled_tw(const DwParams& p) {
    if (p.C % 4 != 0 || (p.C >> 2) > 256) return 0;
    if (p.KH == 3 && p.KW == 3 && (p.SW == 1 || p.SW == 2)) return p.SW == 1 ? dw3_tw() : 4;
    if (p.KH == 5 && p.KW == 5 && p.SW == 1) return dw5_tw();
    // 7x7: TW = 8 needs 260 VGPRs (14-column row window + 8 accumulators of float4): one wavefront per SIMD, 271 us on the det
    // neck's 96-channel map; TW = 4 fits three
    if (p.KH == 7 && p.KW == 7 && p.SW == 1) return dw7_tw();
    return 0;
}
int dwconv_gap_chunks(const DwParams& p) {
    if (dwconv_lds_applies(p)) return dwconv_lds_gap_chunks(p);
    if (dw_col_applies(p)) {
        int c4n, threads, gw, gpb, chunks;
        dw_col_geom(p, c4n, threads, gw, gpb, chunks);
        return chunks;
    }
    const int tw = dw_tiled_tw(p);
    if (!tw) return 0;
    int c4n, threads, gw, g, gpb, chunks;
    dw_tiled_geom(p, tw, c4n, threads, gw, g, gpb, chunks);
    return chunks;
}

void launch_dwconv(const DwParams& p_in, hipStream_t s) {
    static const int dbg_env = [] { const char* e = std::getenv("RD_DW_DBG"); return e ? atoi(e) : 0; }();
    DwParams p = p_in;
    p.dbg = dbg_env;
    if (p.tokinfo) {      // ragged rows: only the per-pixel kernel knows about line ends
        const long total = (long)p.N * p.OH * p.OW * (p.C >> 2);
        hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for(total)), dim3(256), 0, s, p);
        return;
    }
    if (dwconv_lds_applies(p)) {
        launch_dwconv_lds(p, s);
        return;
    }
    if (dwconv_kxk_lds_applies(p)) {
        launch_dwconv_kxk_lds(p, s);
        return;
    }
    if (dw_col_applies(p)) {
        int c4n, threads, gw, gpb, chunks;
        dw_col_geom(p, c4n, threads, gw, gpb, chunks);
        hipLaunchKernelGGL((dwconv3x3_col_kernel<4>), dim3(chunks, p.N), dim3(threads), 0, s, p, c4n, gw, gpb);
        return;
    }
    const int tw = dw_tiled_tw(p);
    if (tw) {
        int c4n, threads, gw, g, gpb, chunks;
        dw_tiled_geom(p, tw, c4n, threads, gw, g, gpb, chunks);
        dim3 grid(chunks, p.N), block(threads);
        if (p.KH == 3 && p.SW == 1 && tw == 8 && p.dbg) {
            switch (p.dbg) {
                case 1: hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 1, 8, 1>), grid, block, 0, s, p, c4n, gw, g, gpb); break;
                case 2: hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 1, 8, 2>), grid, block, 0, s, p, c4n, gw, g, gpb); break;
                case 3: hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 1, 8, 3>), grid, block, 0, s, p, c4n, gw, g, gpb); break;
                case 4: hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 1, 8, 4>), grid, block, 0, s, p, c4n, gw, g, gpb); break;
                default: hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 1, 8, 7>), grid, block, 0, s, p, c4n, gw, g, gpb); break;
            }
        } else if (p.KH == 3 && p.SW == 1 && tw == 8)
            hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 1, 8>), grid, block, 0, s, p, c4n, gw, g, gpb);
        else if (p.KH == 3 && p.SW == 1)
            hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 1, 4>), grid, block, 0, s, p, c4n, gw, g, gpb);
        else if (p.KH == 3)
            hipLaunchKernelGGL((dwconv_tiled_kernel<3, 3, 2, 4>), grid, block, 0, s, p, c4n, gw, g, gpb);
        else if (p.KH == 5 && tw == 8)
            hipLaunchKernelGGL((dwconv_tiled_kernel<5, 5, 1, 8>), grid, block, 0, s, p, c4n, gw, g, gpb);
        else if (p.KH == 5)
            hipLaunchKernelGGL((dwconv_tiled_kernel<5, 5, 1, 4>), grid, block, 0, s, p, c4n, gw, g, gpb);
        else if (tw == 8)
            hipLaunchKernelGGL((dwconv_tiled_kernel<7, 7, 1, 8>), grid, block, 0, s, p, c4n, gw, g, gpb);
        else
            hipLaunchKernelGGL((dwconv_tiled_kernel<7, 7, 1, 4>), grid, block, 0, s, p, c4n, gw, g, gpb);
        return;
    }
    const long total = (long)p.N * p.OH * p.OW * (p.C >> 2);
    hipLaunchKernelGGL(dwconv_kernel, dim3(grid_for(total)), dim3(256), 0, s, p);
}

// --------------------------------------------------------------------------------------------------
// Pools
// --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) maxpool2x2s1_kernel(const float* x, int xld, float* y, int yld, int N, int H,
                                                           int W, int C) {
    // F.pad(x,(0,1,0,1)) then MaxPool2d(2, stride 1, ceil) (rec_lcnetv4.py:161-166; rec_pphgnetv2.py:962-976):
    // the padding is ZERO, so the border takes max(.., 0).
    const int c4n = C >> 2;
    const long total = (long)N * H * W * c4n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) << 2;
        const long pix = idx / c4n;
        const int w = pix % W;
        const int h = (pix / W) % H;
        const float* xp = x + (size_t)pix * xld + c;
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        f32x4 v = *reinterpret_cast<const f32x4*>(xp);
        const f32x4 r = (w + 1 < W) ? *reinterpret_cast<const f32x4*>(xp + xld) : z;
        const f32x4 d = (h + 1 < H) ? *reinterpret_cast<const f32x4*>(xp + (size_t)W * xld) : z;
        const f32x4 rd_ = (w + 1 < W && h + 1 < H) ? *reinterpret_cast<const f32x4*>(xp + (size_t)(W + 1) * xld) : z;
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = fmaxf(fmaxf(v[i], r[i]), fmaxf(d[i], rd_[i]));
        *reinterpret_cast<f32x4*>(y + (size_t)pix * yld + c) = v;
    }
}
void launch_maxpool2x2s1(const float* x, int xld, float* y, int yld, int N, int H, int W, int C, hipStream_t s) {
    const long total = (long)N * H * W * (C >> 2);
    hipLaunchKernelGGL(maxpool2x2s1_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, xld, y, yld, N, H, W, C);
}

__global__ void __launch_bounds__(256) avgpool3x2_kernel(const float* x, int xld, float* y, int yld, int N, int H,
                                                         int W, int C, int OH, int OW, const int32_t* line_tab) {
    const int c4n = C >> 2;
    const long total = (long)N * OH * OW * c4n;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) << 2;
        const long pix = idx / c4n;
        const int ow = pix % OW;
        const int oh = (pix / OW) % OH;
        const int b = pix / ((long)OW * OH);
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 2; ++kw)
                acc += *reinterpret_cast<const f32x4*>(
                    x + (((size_t)b * H + oh * 3 + kh) * W + ow * 2 + kw) * xld + c);
        acc *= (1.f / 6.f);
        if (line_tab) {     // compact token buffer: line b owns rows [tok_off, tok_off + w4 / 2) (OH == 1)
            const int t_b = line_tab[b * kLineTabStride + 2] >> 1;
            if (ow < t_b) *reinterpret_cast<f32x4*>(y + (size_t)(line_tab[b * kLineTabStride + 3] + ow) * yld + c) = acc;
        } else {
            *reinterpret_cast<f32x4*>(y + (size_t)pix * yld + c) = acc;
        }
    }
}
void launch_avgpool3x2(const float* x, int xld, float* y, int yld, int N, int H, int W, int C, hipStream_t s, const int32_t* line_tab) {
    const int OH = (H - 3) / 3 + 1, OW = (W - 2) / 2 + 1;
    const long total = (long)N * OH * OW * (C >> 2);
    hipLaunchKernelGGL(avgpool3x2_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, xld, y, yld, N, H, W, C, OH, OW, line_tab);
}

__global__ void __launch_bounds__(256) mask_cols_kernel(float* y, int yld, int H, int W, int C, const int32_t* line_w, int stride, long total) {
    const int c4n = C >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) << 2;
        const long pix = idx / c4n;
        const int w = pix % W;
        const int b = pix / ((long)W * H);
        if (w >= line_w[b * stride]) *reinterpret_cast<f32x4*>(y + (size_t)pix * yld + c) = f32x4{0.f, 0.f, 0.f, 0.f};
    }
}
void launch_mask_cols(float* y, int yld, int N, int H, int W, int C, const int32_t* line_w, int stride, hipStream_t s) {
    const long total = (long)N * H * W * (C >> 2);
    hipLaunchKernelGGL(mask_cols_kernel, dim3(grid_for(total)), dim3(256), 0, s, y, yld, H, W, C, line_w, stride, total);
}

// --------------------------------------------------------------------------------------------------
// Squeeze-excite (rec_lcnetv4.py:120-142; db_fpn.py:288-308)
// --------------------------------------------------------------------------------------------------
// partial[n][chunk][c] = sum over the chunk's pixels; fixed summation order => run-to-run deterministic
__global__ void __launch_bounds__(256) gap_partial_kernel(const float* x, int xld, int HW, int C, float* partial,
                                                          int chunks) {
    const int n = blockIdx.y, chunk = blockIdx.x;
    const int per = (HW + chunks - 1) / chunks;
    const int p0 = chunk * per, p1 = min(HW, p0 + per);
    const int c4n = C >> 2;
    __shared__ f32x4 red[256];
    // threads: tc = channel quad, tr = pixel lane
    const int lanes_c = c4n < 256 ? c4n : 256;
    const int rows = 256 / lanes_c;
    const int tc = threadIdx.x % lanes_c, tr = threadIdx.x / lanes_c;
    for (int cbase = 0; cbase < c4n; cbase += lanes_c) {  // uniform trip count (barriers inside)
        const int cq = cbase + tc;
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        if (tr < rows && cq < c4n)
            for (int pidx = p0 + tr; pidx < p1; pidx += rows)
                acc += *reinterpret_cast<const f32x4*>(x + ((size_t)n * HW + pidx) * xld + (cq << 2));
        red[threadIdx.x] = acc;
        __syncthreads();
        if (tr == 0 && cq < c4n) {
            for (int r = 1; r < rows; ++r) acc += red[r * lanes_c + tc];
            *reinterpret_cast<f32x4*>(partial + ((size_t)n * chunks + chunk) * C + (cq << 2)) = acc;
        }
        __syncthreads();
    }
}
void launch_gap_partial(const float* x, int xld, int N, int HW, int C, float* partial, int chunks, hipStream_t s) {
    hipLaunchKernelGGL(gap_partial_kernel, dim3(chunks, N), dim3(256), 0, s, x, xld, HW, C, partial, chunks);
}

// One workgroup per image; three short phases, each spread over all 256 threads (round 2's version walked the pooling partials
// and the rows of W1 serially per thread: 128 launches of 16 - 44 us of pure load latency).  Every sum has a fixed order.
__global__ void __launch_bounds__(256) se_fc_kernel(SeFcParams p) {
    extern __shared__ __attribute__((aligned(16))) float sm[];  // mean[C] + hid[Cr] + part[KS][C]
    float* mean = sm;
    float* hid = sm + p.C;
    float* part = hid + ((p.Cr + 3) & ~3);
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // 1. pooled mean: thread (slice, channel quad) adds every KS-th partial row, then the KS slices are added in order
    const float inv_hw = p.line_w ? 1.f / (float)(p.H * p.line_w[n * p.line_w_stride]) : p.inv_hw;   // a line's own extent under a line table
    const int c4n = p.C >> 2;
    const int KS = max(1, min(8, 256 / c4n));
    if (tid < KS * c4n) {
        const int sl = tid / c4n, cg = tid - sl * c4n;
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
        for (int k = sl; k < p.chunks; k += KS) s += *reinterpret_cast<const f32x4*>(p.partial + ((size_t)n * p.chunks + k) * p.C + 4 * cg);
        *reinterpret_cast<f32x4*>(part + (size_t)sl * p.C + 4 * cg) = s;
    }
    __syncthreads();
    for (int c = tid; c < p.C; c += 256) {
        float s = 0.f;
        for (int sl = 0; sl < KS; ++sl) s += part[sl * p.C + c];
        mean[c] = s * inv_hw;
    }
    __syncthreads();
    // 2. hidden units: a wavefront takes Cr / 4 consecutive units, eight at a time: lanes across the channels (16-byte coalesced
    //    loads of the eight rows of W1, all issued before the first is used), halving butterfly over the eight sums
    {
        const int per = (p.Cr + 3) >> 2, r_lo = wave * per, r_hi = min(p.Cr, r_lo + per);
        const int steps = (p.C + 255) >> 8;                 // channel quads per lane (C <= 512)
        for (int g0 = r_lo; g0 < r_hi; g0 += 8) {
            f32x4 wv[8][2];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int r = min(g0 + j, p.Cr - 1);
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    const int c = 4 * lane + 256 * i;
                    wv[j][i] = (i < steps && c < p.C) ? *reinterpret_cast<const f32x4*>(p.w1 + (size_t)r * p.C + c) : f32x4{0.f, 0.f, 0.f, 0.f};
                }
            }
            float acc[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[j] = 0.f;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int c = 4 * lane + 256 * i;
                if (i < steps && c < p.C) {
                    const f32x4 mv = *reinterpret_cast<const f32x4*>(mean + c);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        acc[j] += wv[j][i][0] * mv[0] + wv[j][i][1] * mv[1] + wv[j][i][2] * mv[2] + wv[j][i][3] * mv[3];
                }
            }
            int cnt = 8, idx = 0;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                if (cnt > 1) {
                    cnt >>= 1;
                    const bool up = (lane & o) != 0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (j < cnt) {
                            const float lo = acc[j], hi = acc[j + cnt];
                            acc[j] = (up ? hi : lo) + __shfl_xor(up ? lo : hi, o, 64);
                        }
                    }
                    if (up) idx += cnt;
                } else {
                    acc[0] += __shfl_xor(acc[0], o, 64);
                }
            }
            const int r = g0 + idx;
            if ((lane & 7) == 0 && r < r_hi) hid[r] = fmaxf(acc[0] + p.b1[r], 0.f);
        }
    }
    __syncthreads();
    // 3. gates: one thread per channel, its row of W2 (Cr floats, 16-byte loads)
    for (int c = tid; c < p.C; c += 256) {
        float s = p.b2[c];
        const float* w = p.w2 + (size_t)c * p.Cr;
        if ((p.Cr & 3) == 0) {
            for (int r = 0; r < p.Cr; r += 4) {
                const f32x4 wv = *reinterpret_cast<const f32x4*>(w + r);
                s = fmaf(wv[0], hid[r], s); s = fmaf(wv[1], hid[r + 1], s); s = fmaf(wv[2], hid[r + 2], s); s = fmaf(wv[3], hid[r + 3], s);
            }
        } else {
            for (int r = 0; r < p.Cr; ++r) s = fmaf(w[r], hid[r], s);
        }
        p.scale[(size_t)n * p.C + c] = rd_act(s, p.gate);
    }
}
void launch_se_fc(const SeFcParams& p, hipStream_t s) {
    const int ks = std::max(1, std::min(8, 256 / (p.C >> 2)));
    const size_t sh = ((size_t)p.C + ((p.Cr + 3) & ~3) + (size_t)ks * p.C) * sizeof(float);
    hipLaunchKernelGGL(se_fc_kernel, dim3(p.N), dim3(256), sh, s, p);
}

__global__ void __launch_bounds__(256) scale_channels_kernel(const float* x, int xld, float* y, int yld,
                                                             const float* scale, float alpha, int HW, int C,
                                                             long total) {
    const int c4n = C >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) << 2;
        const long pix = idx / c4n;
        const int n = pix / HW;
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)pix * xld + c);
        const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + (size_t)n * C + c);
        v *= (sc + alpha);
        *reinterpret_cast<f32x4*>(y + (size_t)pix * yld + c) = v;
    }
}
void launch_scale_channels(const float* x, int xld, float* y, int yld, const float* scale, float alpha, int N, int HW,
                           int C, hipStream_t s) {
    const long total = (long)N * HW * (C >> 2);
    hipLaunchKernelGGL(scale_channels_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, xld, y, yld, scale, alpha, HW,
                       C, total);
}

// --------------------------------------------------------------------------------------------------
// DB head tail: ConvTranspose2d(C, C, 2, 2) + BN + ReLU -> ConvTranspose2d(C, 1, 2, 2) -> sigmoid (det_db_head.py:66-72,124-144) as ONE
// kernel.  As two transposed-conv GEMMs (N = 96 and N = 4 on 128-wide fp32 MFMA tiles, scatter epilogues) they took 232 us per
// 8 pages at 1.2 - 1.5 TB/s and passed a 130-MB 24-channel map of the doubled resolution through HBM.  A stride-2 2x2 transposed
// conv has no overlap between taps: one input pixel owns a 2 x 2 block of the middle map and through it a 4 x 4 block of the
// probability map.  One thread = one input pixel: 24 inputs in registers, per tap 24 x 24 FMAs (weights are wave-uniform: scalar
// loads), ReLU, 4 x 24 FMAs of the final layer, sigmoid; four 16-byte stores per thread, consecutive lanes = consecutive columns.
// Plain fp32 FMAs in every precision mode.
// --------------------------------------------------------------------------------------------------
// (Two pixels per thread, so that a scalar weight load feeds two FMAs, measured 126 us against 69: the wider loop body made the
//  compiler hoist the scalar loads until it ran out of SGPRs - 2400 v_readlane / v_writelane spills - and half the wavefronts hide
//  the scalar-load latency worse.)
template <int CIN, int CMID>
__global__ void __launch_bounds__(256) det_head_tail_kernel(const float* __restrict__ x, int xld, int N, int H, int W,
                                                            const float* __restrict__ w1, const float* __restrict__ b1,
                                                            const float* __restrict__ w2, const float* __restrict__ b2p, float* __restrict__ y) {
    const long total = (long)N * H * W;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int iw = (int)(idx % W);
    const long t = idx / W;
    const int ih = (int)(t % H), n = (int)(t / H);
    float xin[CIN];
#pragma unroll
    for (int c = 0; c < CIN; c += 4) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + (size_t)idx * xld + c);
        xin[c] = v[0]; xin[c + 1] = v[1]; xin[c + 2] = v[2]; xin[c + 3] = v[3];
    }
    float o[4][4];          // [output row 2 dy + dy2][output column 2 dx + dx2]
#pragma unroll
    for (int tap = 0; tap < 4; ++tap) {
        const int dy = tap >> 1, dx = tap & 1;
        float u[CMID];
#pragma unroll
        for (int co = 0; co < CMID; ++co) {
            float a = b1[co];
            const float* wr = w1 + ((size_t)tap * CMID + co) * CIN;
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci) a = fmaf(wr[ci], xin[ci], a);
            u[co] = fmaxf(a, 0.f);
        }
#pragma unroll
        for (int tap2 = 0; tap2 < 4; ++tap2) {
            float a = b2p[0];
            const float* wr = w2 + (size_t)tap2 * CMID;
#pragma unroll
            for (int co = 0; co < CMID; ++co) a = fmaf(wr[co], u[co], a);
            o[2 * dy + (tap2 >> 1)][2 * dx + (tap2 & 1)] = rd_act(a, ACT_SIGMOID);
        }
    }
    float* yb = y + ((size_t)n * 4 * H + 4 * ih) * (size_t)(4 * W) + 4 * iw;
#pragma unroll
    for (int r = 0; r < 4; ++r)
        __builtin_nontemporal_store(f32x4{o[r][0], o[r][1], o[r][2], o[r][3]}, reinterpret_cast<f32x4*>(yb + (size_t)r * 4 * W));
}
bool det_head_tail_supported(int cin, int cmid, int cout) { return cin == 24 && cmid == 24 && cout == 1; }
void launch_det_head_tail(const float* x, int xld, int N, int H, int W, const float* w1, const float* b1, const float* w2, const float* b2,
                          float* y, hipStream_t s) {
    const long total = (long)N * H * W;
    hipLaunchKernelGGL((det_head_tail_kernel<24, 24>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, xld, N, H, W, w1, b1, w2,
                       b2, y);
}

// --------------------------------------------------------------------------------------------------
// nearest upsample (+ accumulate) - RepLKFPN top-down path and concat (db_fpn.py:395-415)
// --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) upsample_kernel(const float* x, int xld, float* y, int yld, int H, int W, int C,
                                                       int f, int accumulate, long total) {
    // H, W are the OUTPUT dims; input is H/f x W/f
    const int c4n = C >> 2;
    const int IH = H / f, IW = W / f;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) << 2;
        const long pix = idx / c4n;
        const int w = pix % W;
        const int h = (pix / W) % H;
        const int n = pix / ((long)W * H);
        f32x4 v = *reinterpret_cast<const f32x4*>(x + (((size_t)n * IH + h / f) * IW + w / f) * xld + c);
        float* yp = y + (size_t)pix * yld + c;
        if (accumulate) v += *reinterpret_cast<const f32x4*>(yp);
        *reinterpret_cast<f32x4*>(yp) = v;
    }
}
void launch_upsample(const float* x, int xld, float* y, int yld, int N, int H, int W, int C, int f, int accumulate,
                     hipStream_t s) {
    const long total = (long)N * H * W * (C >> 2);
    hipLaunchKernelGGL(upsample_kernel, dim3(grid_for(total)), dim3(256), 0, s, x, xld, y, yld, H, W, C, f, accumulate,
                       total);
}

__global__ void __launch_bounds__(256) add_kernel(const float* a, int ald, const float* b, int bld, float* y, int yld,
                                                  int C, long total) {
    const int c4n = C >> 2;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
        const int c = (int)(idx % c4n) << 2;
        const long m = idx / c4n;
        const f32x4 v = *reinterpret_cast<const f32x4*>(a + (size_t)m * ald + c) +
                        *reinterpret_cast<const f32x4*>(b + (size_t)m * bld + c);
        *reinterpret_cast<f32x4*>(y + (size_t)m * yld + c) = v;
    }
}
void launch_add(const float* a, int ald, const float* b, int bld, float* y, int yld, int M, int C, hipStream_t s) {
    const long total = (long)M * (C >> 2);
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(total)), dim3(256), 0, s, a, ald, b, bld, y, yld, C, total);
}

// --------------------------------------------------------------------------------------------------
// LayerNorm over the channel dimension: one wavefront per token (necks/rnn.py:306-318,363)
// --------------------------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

__global__ void __launch_bounds__(256) layernorm_kernel(const float* x, int xld, float* y, int yld, const float* g,
                                                        const float* b, int M, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + (size_t)row * xld;
    float v[8];  // C <= 512
    float s = 0.f;
    int cnt = 0;
    for (int c = lane; c < C; c += 64) {
        v[cnt] = xr[c];
        s += v[cnt++];
    }
    const float mean = wave_sum(s) / C;
    float q = 0.f;
    for (int i = 0; i < cnt; ++i) {
        const float d = v[i] - mean;
        q += d * d;
    }
    const float rstd = rsqrtf(wave_sum(q) / C + eps);
    float* yr = y + (size_t)row * yld;
    cnt = 0;
    for (int c = lane; c < C; c += 64) yr[c] = (v[cnt++] - mean) * rstd * g[c] + b[c];
}
void launch_layernorm(const float* x, int xld, float* y, int yld, const float* g, const float* b, int M, int C,
                      float eps, hipStream_t s) {
    hipLaunchKernelGGL(layernorm_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, xld, y, yld, g, b, M, C, eps);
}

// --------------------------------------------------------------------------------------------------
// LightSVTR global self-attention (necks/rnn.py:238-271): heads=8, head_dim=15, T = W/8 (<= 1024).
// One block per (sequence, head): K and V live in LDS, each thread owns query rows, two passes
// (row max, then exp/sum/PV) exactly like a max-subtracted softmax.
// --------------------------------------------------------------------------------------------------
template <int HD>
__global__ void __launch_bounds__(256) attention_kernel(const float* qkv, float* o, int T, int heads, float scale, const int32_t* seg,
                                                        int tk_cap, int skip_upto) {
    // K / V rows are padded to HDP = 16 / 32 floats in LDS so that a key costs HDP / 4 ds_read_b128 (every lane reads the same
    // address: a broadcast) instead of HD ds_read_b32 - round 1's 15 scalar reads per dot product made this kernel LDS-issue
    // bound (51 us per launch for 0.4 GFLOP); same two-pass max-subtracted softmax, same operation order per element.
    // Sequences longer than the LDS holds (tk_cap keys; > 1200 tokens = a text line wider than ~9600 px at height 48) run the
    // same two passes over KEY TILES that are re-staged per pass: identical arithmetic in identical order, more LDS fills.
    constexpr int HDP = (HD + 3) / 4 * 4;
    constexpr int NV = HDP / 4;
    extern __shared__ float sm[];
    const int TL = min(T, tk_cap);                          // LDS rows (the launcher sized the allocation with the same rule)
    f32x4* Ks = reinterpret_cast<f32x4*>(sm);               // [TL][NV]
    f32x4* Vs = Ks + (size_t)TL * NV;
    const int b = blockIdx.x, h = blockIdx.y;
    const int C = heads * HD;
    size_t tok0 = (size_t)b * T;
    if (seg) {                 // ragged batch: this sequence's own offset and length (T was the longest: it sized the LDS)
        tok0 = (size_t)seg[2 * b];
        T = seg[2 * b + 1];
        if (T <= skip_upto) return;      // (lines the matrix-core kernel serves: launch_attention)
    }
    const float* base = qkv + tok0 * 3 * C;
    auto stage = [&](int k0, int kn, bool with_v) {         // keys [k0, k0 + kn) -> LDS rows [0, kn)
        for (int i = threadIdx.x; i < kn * HDP; i += 256) {
            const int t = i / HDP, d = i - t * HDP;
            reinterpret_cast<float*>(Ks)[i] = d < HD ? base[(size_t)(k0 + t) * 3 * C + C + h * HD + d] : 0.f;
            if (with_v) reinterpret_cast<float*>(Vs)[i] = d < HD ? base[(size_t)(k0 + t) * 3 * C + 2 * C + h * HD + d] : 0.f;
        }
    };
    const bool tiled = T > TL;
    if (!tiled) {
        stage(0, T, true);
        __syncthreads();
    }
    const int rounds = (T + 255) / 256;
    for (int r = 0; r < rounds; ++r) {
        const int t = r * 256 + threadIdx.x;
        const bool live = t < T;
        f32x4 q[NV];
        const float* qrow = base + (size_t)min(t, T - 1) * 3 * C + h * HD;      // (clamped: unconditional loads are issued back to back)
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) q[v][e] = qrow[min(4 * v + e, HD - 1)];
#pragma unroll
        for (int v = 0; v < NV; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) q[v][e] = (live && 4 * v + e < HD) ? q[v][e] * scale : 0.f;
        auto dot = [&](int j) {
            float sdot = 0.f;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const f32x4 kv = Ks[(size_t)j * NV + v];
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    if (4 * v + e < HD) sdot = fmaf(q[v][e], kv[e], sdot);
            }
            return sdot;
        };
        float mx = -INFINITY;
        if (!tiled) {
            if (live)
                for (int j = 0; j < T; ++j) mx = fmaxf(mx, dot(j));
        } else {
            for (int k0 = 0; k0 < T; k0 += TL) {
                const int kn = min(TL, T - k0);
                __syncthreads();
                stage(k0, kn, false);
                __syncthreads();
                if (live)
                    for (int j = 0; j < kn; ++j) mx = fmaxf(mx, dot(j));
            }
        }
        float l = 0.f;
        f32x4 acc[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) acc[v] = f32x4{0.f, 0.f, 0.f, 0.f};
        auto accumulate = [&](int kn) {
            for (int j = 0; j < kn; ++j) {
                const float e = __expf(dot(j) - mx);
                l += e;
#pragma unroll
                for (int v = 0; v < NV; ++v) {
                    const f32x4 vv = Vs[(size_t)j * NV + v];
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[v][c] = fmaf(e, vv[c], acc[v][c]);
                }
            }
        };
        if (!tiled) {
            if (live) accumulate(T);
        } else {
            for (int k0 = 0; k0 < T; k0 += TL) {
                const int kn = min(TL, T - k0);
                __syncthreads();
                stage(k0, kn, true);
                __syncthreads();
                if (live) accumulate(kn);
            }
        }
        if (live) {
            const float inv = 1.f / l;
            float* op = o + (tok0 + t) * C + h * HD;
#pragma unroll
            for (int v = 0; v < NV; ++v)
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    if (4 * v + c < HD) op[4 * v + c] = acc[v][c] * inv;
        }
    }
}
// keys whose padded K and V rows fit the dynamic LDS of one workgroup (144 KB of the CU's 160: the rest is left to the
// kernels of other streams that share the CU)
int attention_lds_keys(int hd) { return (144 * 1024) / (2 * ((hd + 3) / 4 * 4) * (int)sizeof(float)); }
template <int HD>
static void launch_attention_t(const float* qkv, float* o, int B, int T, int heads, float scale, hipStream_t s, const int32_t* seg, int skip_upto) {
    const int cap = attention_lds_keys(HD);
    const size_t sh = (size_t)2 * std::min(T, cap) * ((HD + 3) / 4 * 4) * sizeof(float);
    static unsigned long long lds_ok = 0;
    if (sh > 64 * 1024) rd_allow_dynamic_lds((const void*)attention_kernel<HD>, (size_t)144 * 1024, lds_ok);
    hipLaunchKernelGGL(attention_kernel<HD>, dim3(B, heads), dim3(256), sh, s, qkv, o, T, heads, scale, seg, cap, skip_upto);
}
void launch_attention(const float* qkv, float* o, int B, int T, int heads, int hd, float scale, hipStream_t s, const int32_t* seg) {
    // The kernel that serves a line follows from the LINE's length (matrix cores up to attention_h3_max_t() tokens, the VALU kernel beyond),
    // not from the longest line of its launch: a ragged launch (seg) that holds lines of both kinds runs both kernels over the same line
    // table, each skipping the other's lines, so a line's bits do not depend on its neighbours in the launch.
    int skip_upto = 0;
    if (attention_h3_applies(1, hd)) {
        if (attention_h3_applies(T, hd) || seg) launch_attention_h3(qkv, o, B, T, heads, hd, scale, s, seg);
        if (attention_h3_applies(T, hd)) return;
        if (seg) skip_upto = attention_h3_max_t();
    }
    if (hd == 15) launch_attention_t<15>(qkv, o, B, T, heads, scale, s, seg, skip_upto);
    else if (hd == 16) launch_attention_t<16>(qkv, o, B, T, heads, scale, s, seg, skip_upto);
    else if (hd == 32) launch_attention_t<32>(qkv, o, B, T, heads, scale, s, seg, skip_upto);
}

// --------------------------------------------------------------------------------------------------
// CTC row statistics over materialised logits (used for validation and for the reference-shaped
// softmax output; the production path is the fused kernel in kernels_ctc.hip)
// --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) rowmax_softmax_kernel(const float* z, int ld, int M, int C, int32_t* idx,
                                                             float* prob) {
    const int row = blockIdx.x;
    const float* zr = z + (size_t)row * ld;
    float mx = -INFINITY;
    int mi = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float v = zr[c];
        if (v > mx) { mx = v; mi = c; }
    }
    __shared__ float smx[256];
    __shared__ int smi[256];
    __shared__ float ssum[256];
    smx[threadIdx.x] = mx;
    smi[threadIdx.x] = mi;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) {
            const float a = smx[threadIdx.x], b = smx[threadIdx.x + o];
            const int ia = smi[threadIdx.x], ib = smi[threadIdx.x + o];
            if (b > a || (b == a && ib < ia)) { smx[threadIdx.x] = b; smi[threadIdx.x] = ib; }
        }
        __syncthreads();
    }
    const float gmx = smx[0];
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += __expf(zr[c] - gmx);
    ssum[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) ssum[threadIdx.x] += ssum[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        idx[row] = smi[0];
        prob[row] = 1.f / ssum[0];
    }
}
void launch_rowmax_softmax(const float* logits, int ld, int M, int C, int32_t* idx, float* prob, hipStream_t s) {
    if (M > 0) hipLaunchKernelGGL(rowmax_softmax_kernel, dim3(M), dim3(256), 0, s, logits, ld, M, C, idx, prob);
}

// With `idx` / `prob`: the two reductions rapidocr's CTCLabelDecode performs on this very tensor (`preds.argmax(axis=2)`, `preds.max(axis=2)`,
// rapid_ocr.py:443-449), computed from the VALUES WRITTEN - prob = the row's largest softmax value (exp(0) * inv = inv), idx = the lowest class
// whose written value equals it, which is what numpy's argmax returns when two logits round to the same probability - so a host that asks the
// lazy S2 result (session.LazySoftmax) for argmax / max gets bit for bit what it would get from the materialised array.
__global__ void __launch_bounds__(256) row_softmax_kernel(const float* z, int ld, float* out, int C, int32_t* idx, float* prob) {
    const int row = blockIdx.x;
    const float* zr = z + (size_t)row * ld;
    float mx = -INFINITY;
    for (int c = threadIdx.x; c < C; c += 256) mx = fmaxf(mx, zr[c]);
    __shared__ float red[4];
    mx = wave_max(mx);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    __syncthreads();
    float s = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) s += __expf(zr[c] - mx);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const float inv = 1.f / (red[0] + red[1] + red[2] + red[3]);
    float* orow = out + (size_t)row * C;
    int first = 0x7fffffff;
    for (int c = threadIdx.x; c < C; c += 256) {
        const float pv = __expf(zr[c] - mx) * inv;
        orow[c] = pv;
        if (pv == inv && c < first) first = c;          // (the row's maximum is written as exactly `inv`: __expf(0) = 1)
    }
    if (idx == nullptr) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) first = min(first, __shfl_xor(first, o, 64));
    __shared__ int redi[4];
    if ((threadIdx.x & 63) == 0) redi[threadIdx.x >> 6] = first;
    __syncthreads();
    if (threadIdx.x == 0) {
        idx[row] = min(min(redi[0], redi[1]), min(redi[2], redi[3]));
        prob[row] = inv;
    }
}
void launch_row_softmax(const float* logits, int ld, float* out, int M, int C, hipStream_t s, int32_t* idx, float* prob) {
    if (M > 0) hipLaunchKernelGGL(row_softmax_kernel, dim3(M), dim3(256), 0, s, logits, ld, out, C, idx, prob);
}

// --------------------------------------------------------------------------------------------------
// Layout conversion at the boundary (the reference seam is NCHW numpy: inference_engine/base.py)
// --------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const float* x, int xld, float* y, int HW, int C) {
    // tile transpose through LDS: 32 pixels x 32 channels
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int pp = p0 + r, c = c0 + tx;
        tile[r][tx] = (pp < HW && c < C) ? x[((size_t)n * HW + pp) * xld + c] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, pp = p0 + tx;
        if (c < C && pp < HW) y[((size_t)n * C + c) * HW + pp] = tile[tx][r];
    }
}
void launch_nhwc_to_nchw(const float* x, int xld, float* y, int N, int H, int W, int C, hipStream_t s) {
    const int HW = H * W;
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(256), 0, s, x, xld, y, HW, C);
}
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* x, float* y, int yld, int HW, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, pp = p0 + tx;
        tile[r][tx] = (pp < HW && c < C) ? x[((size_t)n * C + c) * HW + pp] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int pp = p0 + r, c = c0 + tx;
        if (c < C && pp < HW) y[((size_t)n * HW + pp) * yld + c] = tile[tx][r];
    }
}
void launch_nchw_to_nhwc(const float* x, float* y, int yld, int N, int C, int H, int W, hipStream_t s) {
    const int HW = H * W;
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((HW + 31) / 32, (C + 31) / 32, N), dim3(256), 0, s, x, y, yld, HW, C);
}

// (image pre-processing - resize / normalise / text-line crops with OpenCV's 8-bit arithmetic - lives in kernels_image.hip;
//  crop_batch_kernel below is the round-1 single-tap bilinear line crop, kept behind rd_crop_resize_norm_batch)
__global__ void __launch_bounds__(256) crop_batch_kernel(CropBatchParams p) {
    const int i = blockIdx.y;
    const CropDesc d = p.descs[i];
    const long plane = (long)p.OH * p.OWp;
    float* dst = p.dst + (size_t)i * 3 * plane;
    const uint8_t* src = p.pages + (size_t)d.page * p.page_stride;
    const float sx = d.rot90 ? d.crop_h / d.out_w : d.crop_w / d.out_w;
    const float sy = d.rot90 ? d.crop_w / p.OH : d.crop_h / p.OH;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < plane; idx += (long)gridDim.x * 256) {
        const int ox = idx % p.OWp, oy = idx / p.OWp;
        float out[3] = {0.f, 0.f, 0.f};
        const bool inside = ox < d.out_w;
        if (inside) {
            // resized-line pixel -> (possibly rotated) crop pixel centre
            float u = (ox + 0.5f) * sx - 0.5f, v = (oy + 0.5f) * sy - 0.5f;
            float cx = u, cy = v;
            if (d.rot90) {  // np.rot90(crop): rotated[r][c] = crop[c][Wc-1-r]
                cx = d.crop_w - 1.f - v;
                cy = u;
            }
            cx = fminf(fmaxf(cx, 0.f), d.crop_w - 1.f);
            cy = fminf(fmaxf(cy, 0.f), d.crop_h - 1.f);
            const float wq = d.m[6] * cx + d.m[7] * cy + d.m[8];
            float px = (d.m[0] * cx + d.m[1] * cy + d.m[2]) / wq;
            float py = (d.m[3] * cx + d.m[4] * cy + d.m[5]) / wq;
            px = fminf(fmaxf(px, 0.f), (float)(p.W - 1));  // BORDER_REPLICATE
            py = fminf(fmaxf(py, 0.f), (float)(p.H - 1));
            const int x0 = (int)px, y0 = (int)py;
            const int x1 = min(x0 + 1, p.W - 1), y1 = min(y0 + 1, p.H - 1);
            const float tx = px - x0, ty = py - y0;
            const uint8_t* r0 = src + ((size_t)y0 * p.W) * 3;
            const uint8_t* r1 = src + ((size_t)y1 * p.W) * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const float a = r0[x0 * 3 + c], b = r0[x1 * 3 + c], e = r1[x0 * 3 + c], f = r1[x1 * 3 + c];
                out[c] = (a * (1.f - tx) + b * tx) * (1.f - ty) + (e * (1.f - tx) + f * tx) * ty;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int sc = p.swap_rb ? 2 - c : c;
            dst[(size_t)c * plane + idx] = inside ? (out[sc] * p.scale - p.mean[c]) * p.inv_std[c] : 0.f;
        }
    }
}
void launch_crop_resize_norm_batch(const CropBatchParams& p, hipStream_t s) {
    if (p.n <= 0) return;
    const long plane = (long)p.OH * p.OWp;
    hipLaunchKernelGGL(crop_batch_kernel, dim3(grid_for(plane, 256, 64), p.n), dim3(256), 0, s, p);
}

}  // namespace rd
