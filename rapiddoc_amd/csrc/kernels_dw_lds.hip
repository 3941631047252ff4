// Depthwise 3x3 / stride 1 of the recogniser's PPLCNetV4 blocks (rec_lcnetv4.py:187-206) through LDS-DMA-staged tiles.
//
// The row-tiled register kernel (kernels_misc.hip) reads every input element 3.75 times through the vector L1 and its three
// phases - loads, arithmetic + column masks, stores - add up at three wavefronts per SIMD (43 us on a 101 376 x 192 map whose
// HBM floor is 25).  Here a workgroup owns a FULL-HEIGHT strip of 16 output columns x one slice of 4 * CS channels:
//   * the strip plus its two halo columns travels global -> LDS by `global_load_lds_dwordx4` - every input element is requested
//     once, no staging VGPRs, no per-load s_waitcnt, and the recogniser's maps are 3 / 6 / 12 rows high so there is no row halo;
//   * one thread = one output column x RT rows x 4 channels, fed by ds_read_b128 (a wavefront reads 1 KB contiguous: no bank
//     conflicts), 3 reads per output;
//   * columns outside the image / beyond the line's own width (LineTab) are handled by zeroing the three per-column WEIGHT vectors
//     once per thread - the DMA address is clamped, so what sits in those LDS columns is finite data of the same map;
//   * nothing is persistent: 27.6 KB of LDS per workgroup = five workgroups per CU, and the hardware overlaps one group's DMA with
//     another's arithmetic and stores - the overlap the register kernel could not get;
//   * workgroup -> tile order is XCD-aware: neighbouring strips (which share a halo column) run on the same XCD's L2.
// Same optional fused squeeze-excite partial sums of the OUTPUT as the register kernel: partial[n][column strip][c].
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "rd_device.h"

namespace rd {

static constexpr int kDwTileCols = 16;

__device__ __forceinline__ f32x4 act4v(f32x4 v, int act) {
    f32x4 r = {rd_act(v[0], act), rd_act(v[1], act), rd_act(v[2], act), rd_act(v[3], act)};
    return r;
}

__device__ __forceinline__ void dw_dma16(const void* src, unsigned lds_byte_offset, unsigned char* smem) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(smem + lds_byte_offset), 16, 0, 0);
}

// CS = float4 channel lanes per workgroup (16: 64 channels, 8: 32 channels); RT = output rows per thread.  256 threads =
// CS channel lanes x 16 columns x RH row groups; the OUTPUT map is RT * RH rows high, the input SH times that (SH = 2: the
// stride-(2, 1) depthwise conv in front of a stage's first block, pad 1: output row r reads input rows 2 r - 1 .. 2 r + 1).
template <int CS, int RT, int SH = 1>
__global__ void __launch_bounds__(256) dwconv3x3_lds_kernel(DwParams p, int ct_n, int ns, int ntiles, int per_xcd) {
    constexpr int PX = 256 / CS, RH = PX / kDwTileCols, HO = RT * RH, HH = SH * HO;
    constexpr int TCOL = kDwTileCols + 2;
    constexpr int NPIX = HH * TCOL;
    constexpr int PPC = 64 / CS;                          // pixels per DMA instruction of one wavefront
    constexpr int NCHUNK = (NPIX + PPC - 1) / PPC;
    __shared__ __attribute__((aligned(16))) unsigned char smem[NCHUNK * 1024];
    __shared__ f32x4 red[256];

    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);   // consecutive tiles on one XCD
    if (tile >= ntiles) return;
    const int sl = tile % ns;
    const int ct = (tile / ns) % ct_n;
    const int n = tile / (ns * ct_n);
    const int x0 = ct * kDwTileCols;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cs0 = sl * CS * 4;

    // ---- the strip -> LDS: pixel pi of the tile (row-major over HH x 18) lands at byte pi * CS * 16
    {
        const float* xb = p.x + (size_t)n * HH * p.W * p.xld + cs0 + (lane % CS) * 4;
#pragma unroll
        for (int k0 = 0; k0 < NCHUNK; k0 += 4) {
            const int k = k0 + wave;
            if (k < NCHUNK) {
                const int pi = min(k * PPC + lane / CS, NPIX - 1);
                const int row = pi / TCOL, col = pi - row * TCOL;
                const int gx = min(max(x0 - 1 + col, 0), p.W - 1);
                dw_dma16(xb + ((size_t)row * p.W + gx) * p.xld, (unsigned)k * 1024u, smem);
            }
        }
    }

    const int c4 = tid % CS, pxl = tid / CS;
    const int col = pxl % kDwTileCols, rh = pxl / kDwTileCols;
    const int c = cs0 + c4 * 4;
    const int gx = x0 + col;
    const int wlim = p.line_w ? min(p.W, p.line_w[n * p.line_w_stride]) : p.W;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 wv[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wv[k] = *reinterpret_cast<const f32x4*>(p.w + (size_t)k * p.C + c);
    const f32x4 bias = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + c) : zero4;
#pragma unroll
    for (int kw = 0; kw < 3; ++kw) {
        const bool ok = (unsigned)(gx - 1 + kw) < (unsigned)wlim;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) wv[kh * 3 + kw] = ok ? wv[kh * 3 + kw] : zero4;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const int r0 = RH == 1 ? 0 : rh * RT;          // (one row group: a compile-time zero, so that the row tests below fold)
    f32x4 acc[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) acc[t] = bias;
    const f32x4* tilep = reinterpret_cast<const f32x4*>(smem) + c4;
    constexpr int NROW = SH * (RT - 1) + 3;               // input rows under this thread's RT outputs
#pragma unroll
    for (int j = 0; j < NROW; ++j) {
        const int ir = SH * r0 - 1 + j;
        if (RH == 1 ? (j == 0 || ir >= HH) : ((unsigned)ir >= (unsigned)HH)) continue;     // the conv's zero rows
        const f32x4* rp = tilep + (size_t)(ir * TCOL + col) * CS;
        const f32x4 v0 = rp[0], v1 = rp[CS], v2 = rp[2 * CS];
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            if ((j - kh) % SH != 0) continue;
            const int t = (j - kh) / SH;
            if (j - kh < 0 || t >= RT) continue;
            acc[t] += v0 * wv[kh * 3 + 0];
            acc[t] += v1 * wv[kh * 3 + 1];
            acc[t] += v2 * wv[kh * 3 + 2];
        }
        // pin the row's contribution here: left alone the compiler sinks every FMA into the `gx < OW` block below and keeps all
        // 18 LDS values live (138 VGPRs = three wavefronts per SIMD)
#pragma unroll
        for (int t = 0; t < RT; ++t) asm volatile("" : "+v"(acc[t]));
    }

    f32x4 gsum = zero4;
    if (gx < p.OW) {
        const size_t pix0 = ((size_t)n * HO + r0) * p.OW + gx;
        if (p.act != ACT_NONE) {            // the switch once, not once per value
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] = act4v(acc[t], p.act);
        }
        if (p.res) {
            f32x4 rv[RT];
#pragma unroll
            for (int t = 0; t < RT; ++t) rv[t] = *reinterpret_cast<const f32x4*>(p.res + (pix0 + (size_t)t * p.OW) * p.rld + c);
#pragma unroll
            for (int t = 0; t < RT; ++t) acc[t] += rv[t];
        }
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            *reinterpret_cast<f32x4*>(p.y + (pix0 + (size_t)t * p.OW) * p.yld + c) = acc[t];
            gsum += acc[t];
        }
        if (gx >= wlim) gsum = zero4;
    }
    if (p.gap_partial) {
        red[tid] = gsum;
        __syncthreads();
        if (pxl == 0) {
            for (int r = 1; r < PX; ++r) gsum += red[r * CS + c4];
            *reinterpret_cast<f32x4*>(p.gap_partial + ((size_t)n * ct_n + ct) * p.C + c) = gsum;
        }
    }
}

// --------------------------------------------------------------------------------------------------------------------------------
// Depthwise 5x5 / 7x7, stride 1 (PPHGNetV2's light blocks, rec_pphgnetv2.py:945-953; the RepLK neck of the detector, db_fpn.py:315-323)
// on the same staging scheme.  The register-tiled kernel keeps float4 channel vectors per thread, so its 25 / 49 weight vectors (100 /
// 196 registers) cannot stay in registers: they are re-read per kernel row, four outputs per thread at 7x7 (260 VGPRs for eight), and the
// kernel runs at 2.0 / 1.3 TB/s.  Here a lane owns ONE channel: the K x K weights are 25 / 49 scalar registers, a thread computes a 4 x 4
// patch of outputs of its channel from a (4 + K - 1)^2 window of ds_read_b32 (a wavefront = 32 channels x 2 patches: two 128-byte rows of
// LDS per read), and a workgroup = 32 channels x (4 x 2 patches) = a 16 x 8 output tile whose (16 + K - 1) x (8 + K - 1) x 32-channel input
// tile arrives by LDS-DMA (addresses clamped at the borders, the zero padding applied by a select per value read).  Nothing persistent:
// 31 / 39 KB of LDS, four to five workgroups per CU.
template <int K>
__global__ void __launch_bounds__(256) dwconv_kxk_lds_kernel(DwParams p, int tx_n, int ty_n, int ns, int ntiles, int per_xcd) {
    constexpr int R = K / 2, TW = 4, TH = 4, OWT = 16, OHT = 8;
    constexpr int IW = OWT + K - 1, IH = OHT + K - 1, NPIX = IW * IH;
    constexpr int NCHUNK = (NPIX + 7) / 8;                 // one DMA instruction of a wavefront = 8 pixels x 32 channels
    __shared__ __attribute__((aligned(16))) unsigned char smem[NCHUNK * 1024];
    const int tile = (blockIdx.x & 7) * per_xcd + (blockIdx.x >> 3);
    if (tile >= ntiles) return;
    const int sl = tile % ns;
    const int tx = (tile / ns) % tx_n;
    const int ty = (tile / (ns * tx_n)) % ty_n;
    const int n = tile / (ns * tx_n * ty_n);
    const int x0 = tx * OWT, y0 = ty * OHT, c0 = sl * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    {
        const float* xb = p.x + (size_t)n * p.H * p.W * p.xld + c0 + (lane & 7) * 4;
#pragma unroll
        for (int k0 = 0; k0 < NCHUNK; k0 += 4) {
            const int k = k0 + wave;
            if (k < NCHUNK) {
                const int pi = min(k * 8 + (lane >> 3), NPIX - 1);
                const int row = pi / IW, col = pi - row * IW;
                const int gy = min(max(y0 - R + row, 0), p.H - 1), gx = min(max(x0 - R + col, 0), p.W - 1);
                dw_dma16(xb + ((size_t)gy * p.W + gx) * p.xld, (unsigned)k * 1024u, smem);
            }
        }
    }
    const int ch = tid & 31, pl = tid >> 5, plx = pl & 3, ply = pl >> 2;
    const int c = c0 + ch;
    float wr[K * K];
#pragma unroll
    for (int k = 0; k < K * K; ++k) wr[k] = p.w[(size_t)k * p.C + c];
    const float bias = p.bias ? p.bias[c] : 0.f;
    unsigned colok = 0;                                    // bit i: input column i of this thread's window lies inside the image
#pragma unroll
    for (int i = 0; i < TW + K - 1; ++i) colok |= ((unsigned)(x0 + plx * TW - R + i) < (unsigned)p.W) ? (1u << i) : 0u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    const bool interior = x0 >= R && x0 + OWT + R <= p.W && y0 >= R && y0 + OHT + R <= p.H;
    float acc[TH][TW];
#pragma unroll
    for (int t = 0; t < TH; ++t)
#pragma unroll
        for (int q = 0; q < TW; ++q) acc[t][q] = bias;
    const float* win = reinterpret_cast<const float*>(smem) + (size_t)((ply * TH) * IW + plx * TW) * 32 + ch;
#pragma unroll
    for (int j = 0; j < TH + K - 1; ++j) {
        const bool rowok = (unsigned)(y0 + ply * TH - R + j) < (unsigned)p.H;
        float v[TW + K - 1];
#pragma unroll
        for (int i = 0; i < TW + K - 1; ++i) v[i] = win[(j * IW + i) * 32];
        if (!interior) {                 // (workgroup-uniform: most tiles of a large map lie inside it and skip the selects)
#pragma unroll
            for (int i = 0; i < TW + K - 1; ++i) v[i] = (rowok && ((colok >> i) & 1u)) ? v[i] : 0.f;
        }
#pragma unroll
        for (int kh = 0; kh < K; ++kh) {
            const int t = j - kh;
            if (t < 0 || t >= TH) continue;
#pragma unroll
            for (int q = 0; q < TW; ++q)
#pragma unroll
                for (int kw = 0; kw < K; ++kw) acc[t][q] = fmaf(v[q + kw], wr[kh * K + kw], acc[t][q]);
        }
#pragma unroll
        for (int t = 0; t < TH; ++t)
#pragma unroll
            for (int q = 0; q < TW; ++q) asm volatile("" : "+v"(acc[t][q]));     // (keeps the FMAs here: see the 3x3 kernel)
    }
#pragma unroll
    for (int t = 0; t < TH; ++t) {
        const int oy = y0 + ply * TH + t;
#pragma unroll
        for (int q = 0; q < TW; ++q) {
            const int ox = x0 + plx * TW + q;
            if (oy < p.OH && ox < p.OW) {
                const size_t pix = ((size_t)n * p.OH + oy) * p.OW + ox;
                float o = rd_act(acc[t][q], p.act);
                if (p.res) o += p.res[pix * p.rld + c];
                p.y[pix * p.yld + c] = o;
            }
        }
    }
}

static inline int dw_kxk_lds_k(const DwParams& p) {
    static const bool off = std::getenv("RD_DW_LDS") && std::string(std::getenv("RD_DW_LDS")) == "0";
    if (off || p.tokinfo || p.line_w || p.gap_partial) return 0;       // (gap_partial: the planner sized its buffer for the register kernel)
    if (p.KH != p.KW || (p.KH != 5 && p.KH != 7) || p.SH != 1 || p.SW != 1 || p.PT != p.KH / 2 || p.PL != p.KH / 2) return 0;
    if (p.OH != p.H || p.OW != p.W || p.C % 32 != 0 || p.xld % 4 != 0) return 0;
    return p.KH;
}
bool dwconv_kxk_lds_applies(const DwParams& p) { return dw_kxk_lds_k(p) != 0; }
void launch_dwconv_kxk_lds(const DwParams& p, hipStream_t s) {
    const int k = dw_kxk_lds_k(p);
    const int tx_n = (p.W + 15) / 16, ty_n = (p.H + 7) / 8, ns = p.C / 32;
    const int ntiles = p.N * ty_n * tx_n * ns, per_xcd = (ntiles + 7) / 8;
    dim3 grid(per_xcd * 8), block(256);
    if (k == 5) hipLaunchKernelGGL((dwconv_kxk_lds_kernel<5>), grid, block, 0, s, p, tx_n, ty_n, ns, ntiles, per_xcd);
    else if (k == 7) hipLaunchKernelGGL((dwconv_kxk_lds_kernel<7>), grid, block, 0, s, p, tx_n, ty_n, ns, ntiles, per_xcd);
    else throw std::runtime_error("launch_dwconv_kxk_lds: geometry not supported");
}

// which instantiation serves this geometry: 0 = none
static inline int dw_lds_variant(const DwParams& p) {
    static const bool off = std::getenv("RD_DW_LDS") && std::string(std::getenv("RD_DW_LDS")) == "0";
    if (off || p.tokinfo) return 0;
    if (!(p.KH == 3 && p.KW == 3 && p.SW == 1 && p.PT == 1 && p.PL == 1 && p.OW == p.W)) return 0;
    if (p.SH == 2) {                                      // the stride-(2, 1) layer in front of a stage (H even: OH = H / 2)
        if (p.OH * 2 != p.H || p.C % 4 != 0) return 0;
        if (p.H == 12 && p.C % 32 == 0) return 4;         // <8, 3, 2>
        if (p.H == 6 && p.C % 64 == 0) return 5;          // <16, 3, 2>
        return 0;
    }
    if (p.SH != 1 || p.OH != p.H) return 0;
    // geometry only - no pointer or leading dimension may enter: the planner asks with an unbound DwParams (dwconv_gap_chunks)
    // and must get the answer the launch gets.  The SE partial buffer is one value per 16 x H outputs, ~1 % of the map.
    if (p.H == 6 && p.C % 64 == 0) return 1;              // <16, 6>
    if (p.H == 12 && p.C % 32 == 0) return 2;             // <8, 6>: two row groups
    if (p.H == 3 && p.C % 64 == 0) return 3;              // <16, 3>
    return 0;
}
bool dwconv_lds_applies(const DwParams& p) { return dw_lds_variant(p) != 0; }
int dwconv_lds_gap_chunks(const DwParams& p) { return (p.W + kDwTileCols - 1) / kDwTileCols; }

void launch_dwconv_lds(const DwParams& p, hipStream_t s) {
    const int v = dw_lds_variant(p);
    const int ct_n = (p.W + kDwTileCols - 1) / kDwTileCols;
    const int ns = p.C / ((v == 2 || v == 4) ? 32 : 64);
    const int ntiles = p.N * ct_n * ns;
    const int per_xcd = (ntiles + 7) / 8;
    dim3 grid(per_xcd * 8), block(256);
    switch (v) {
        case 1: hipLaunchKernelGGL((dwconv3x3_lds_kernel<16, 6>), grid, block, 0, s, p, ct_n, ns, ntiles, per_xcd); break;
        case 2: hipLaunchKernelGGL((dwconv3x3_lds_kernel<8, 6>), grid, block, 0, s, p, ct_n, ns, ntiles, per_xcd); break;
        case 3: hipLaunchKernelGGL((dwconv3x3_lds_kernel<16, 3>), grid, block, 0, s, p, ct_n, ns, ntiles, per_xcd); break;
        case 4: hipLaunchKernelGGL((dwconv3x3_lds_kernel<8, 3, 2>), grid, block, 0, s, p, ct_n, ns, ntiles, per_xcd); break;
        case 5: hipLaunchKernelGGL((dwconv3x3_lds_kernel<16, 3, 2>), grid, block, 0, s, p, ct_n, ns, ntiles, per_xcd); break;
        default: throw std::runtime_error("launch_dwconv_lds: geometry not supported");
    }
}

}  // namespace rd
