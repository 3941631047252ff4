// Direct k x k convolution (2x2 / 3x3, stride 1) on the fp16 matrix cores with (hi, lo) split operands - the narrow layers
// of the stems, PPHGNetV2's 3x3 stacks (stages.0 / stages.1 of the PP-DocLayout backbone, rec_pphgnetv2.py:1001-1071) and the DB
// head's conv_down (det_db_head.py).  Arithmetic as in kernels_conv_h3.hip: x = hi + lo * 2^-11, three
// v_mfma_f32_32x32x16_f16 per product, fp32 accumulate, weights split at load time ([Cout][Kp], k = (kh * KW + kw) * Cin + ci).
//
// Why not the implicit GEMM: im2col hands every input element to KH * KW output rows, and the implicit-GEMM kernels split
// (2 VALU per element) and address (integer divisions) each of those copies.  On gfx950 VALU time does not hide under MFMA time
// (tools/probe_mfma_valu.hip), so a 3x3 layer paid its operand split nine times: the register-staged kernel ran B4's 3x3 layers at
// 105 TFLOP/s with the matrix pipe 19 % busy.  Here a workgroup owns an 8 x 32 (or 4 x 64) tile of OUTPUT PIXELS of one image:
//   * the input patch (tile + halo, one chunk of <= 64 input channels) is loaded ONCE, split ONCE and kept in LDS as two fp16
//     planes; the KH * KW taps are LDS address offsets into it (pixel stride = 16 * odd bytes: conflict-free ds_read_b128);
//   * a wavefront owns 32 consecutive output pixels of one row (MFMA A operand straight from the patch) x all output channels
//     (<= 96: one to three 32-wide blocks, accumulators in registers across taps and channel chunks);
//   * the weights of one (tap, channel chunk) are a [Cout][chunk] slab: register-prefetched under the previous tap's MFMAs,
//     triple-buffered in LDS, one barrier per tap; fragment reads run one k-step ahead of their MFMAs (also across taps);
//   * epilogue (bias, activation, residual, range guard) as in the other split kernels.
#include <cstdlib>
#include <type_traits>

#include "rd_device.h"

namespace rd {

struct DirectGeom {
    int PW, PP;          // patch width / pixels
    int S;               // bytes per patch pixel (and per weight row) in one fp16 plane: CCpad * 2 + 16
    int CC, npass;       // input channels per pass, passes
    int TR, TC, WPR;     // tile rows / cols, wavefronts per tile row
    int tiles_r, tiles_c;
    int N32;             // weight rows in LDS (Cout rounded up to 32)
    int fast_epi;        // interior wavefront tiles store through buffer accesses (round 5; RD_DIRECT_FAST_EPI=0: A/B switch)
};

template <int NB, int KS>
__global__ void __launch_bounds__(512) conv_direct_h3_kernel(ConvParams p, DirectGeom g, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lhi = lane >> 5;
    const int wr = wave / g.WPR, wc0 = (wave - wr * g.WPR) * 32;      // this wavefront's row / first column inside the tile
    const int S = g.S, PW = g.PW, PP = g.PP;
    unsigned char* Ph = smem;
    unsigned char* Pl = smem + (size_t)PP * S;
    unsigned char* Wb = smem + (size_t)2 * PP * S;
    const unsigned wplane = (unsigned)g.N32 * (unsigned)S, wbuf = 2 * wplane;
    const int NT = p.KH * p.KW;
    const _Float16* wh = reinterpret_cast<const _Float16*>(p.wh);
    const _Float16* wl = reinterpret_cast<const _Float16*>(p.wl);
    const int Kp = (p.K + 31) & ~31;

    // ---- weight slab staging: 16-byte chunks (8 halfs) of [plane][row][chunk]; this thread's up-to-three chunks
    constexpr int CH8 = KS * 2;                       // chunks per row (CCpad / 8)
    const int wtotal = 2 * g.N32 * CH8;
    int w_src[3];
    unsigned w_dst[3];
    bool w_ok[3];
    bool w_lo[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int q = tid + 512 * i;
        const int plane = q / (g.N32 * CH8), rem = q - plane * (g.N32 * CH8);
        const int row = rem / CH8, ch = rem - row * CH8;
        w_ok[i] = q < wtotal && row < p.Ng && 8 * ch < g.CC;
        w_lo[i] = plane != 0;
        w_src[i] = w_ok[i] ? row * Kp + 8 * ch : 0;
        w_dst[i] = q < wtotal ? (unsigned)plane * wplane + (unsigned)row * S + (unsigned)ch * 16 : 0xffffffffu;
    }
    u32x4 wreg[3];
    const int nsteps = g.npass * NT;                  // steps of one tile = (channel pass, tap); the slab sequence is cyclic
    auto load_w = [&](int gstep) {                    // slab of global step gstep -> registers
        const int st = gstep % nsteps;
        const int koff = (st % NT) * p.Cin + (st / NT) * g.CC;
        // unconditional loads (w_src is 0 for the slots this thread does not own: a valid address) and a mask afterwards: a load
        // under a condition is waited for before the next one is issued (ISA of the first version: load ; s_waitcnt vmcnt(0) pairs)
#pragma unroll
        for (int i = 0; i < 3; ++i) wreg[i] = *reinterpret_cast<const u32x4*>((w_lo[i] ? wl : wh) + w_src[i] + (w_ok[i] ? koff : 0));
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (!w_ok[i]) wreg[i] = u32x4{0u, 0u, 0u, 0u};
    };
    auto write_w = [&](int gstep) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (w_dst[i] != 0xffffffffu) *reinterpret_cast<u32x4*>(Wb + (unsigned)(gstep % 3) * wbuf + w_dst[i]) = wreg[i];
    };

    // ---- patch staging, row by row: a thread owns up to two (column, 4-channel group) slots of a patch row, fixed for the whole
    // kernel, so a row costs it one uniform row test, one add and one 16-byte load per slot.  NB >= 2 (one workgroup per CU
    // anyway): the loads of ALL rows are in flight before the first split; NB == 1: four rows at a time, which keeps the kernel
    // under 128 VGPRs and two workgroups per CU.
    const int G = KS * 4;                             // 4-channel groups per pixel (CCpad / 4)
    const int row_slots = PW * G;                     // <= 1024 (launcher)
    int s_col[2], s_g[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int q = tid + 512 * j;
        s_col[j] = q / G;
        s_g[j] = q - s_col[j] * G;
    }
    const int PH = PP / PW;                           // patch rows (<= 10)
    constexpr int RGRP = NB == 1 ? 4 : 10;            // rows whose loads are in flight together
    float amax = 0.f;
    int img = 0, oh0 = 0, ow0 = 0;
    auto stage_patch = [&](int c0) {
        bool s_ok[2];
#pragma unroll
        for (int j = 0; j < 2; ++j)
            s_ok[j] = tid + 512 * j < row_slots && 4 * s_g[j] < g.CC && c0 + 4 * s_g[j] < p.Cin &&
                      (unsigned)(ow0 - p.PL + s_col[j]) < (unsigned)p.W;
        const float* xb = p.x + (size_t)img * p.H * p.W * p.xld;
        size_t s_off[2];           // element offset of the slot inside an image row (a slot that is masked reads the row's first pixel)
#pragma unroll
        for (int j = 0; j < 2; ++j) s_off[j] = s_ok[j] ? (size_t)(ow0 - p.PL + s_col[j]) * p.xld + c0 + 4 * s_g[j] : 0;
        for (int r0 = 0; r0 < PH; r0 += RGRP) {
            f32x4 v[RGRP][2];
            // every load of the row group is issued before the first is used: unconditional, from clamped rows / the slot's own (or a
            // valid substitute) column, masked afterwards
#pragma unroll
            for (int rr = 0; rr < RGRP; ++rr) {
                const int ih = oh0 - p.PT + r0 + rr;
                const float* xr = xb + (size_t)min(max(ih, 0), p.H - 1) * p.W * p.xld;
#pragma unroll
                for (int j = 0; j < 2; ++j) v[rr][j] = *reinterpret_cast<const f32x4*>(xr + s_off[j]);
            }
#pragma unroll
            for (int rr = 0; rr < RGRP; ++rr) {
                const int ih = oh0 - p.PT + r0 + rr;
                const bool row_ok = r0 + rr < PH && (unsigned)ih < (unsigned)p.H;
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if (!(row_ok && s_ok[j])) v[rr][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int rr = 0; rr < RGRP; ++rr) {
                if (r0 + rr >= PH) break;
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    if (tid + 512 * j >= row_slots) continue;
                    f16x4 hi, lo;
                    rd_split4(v[rr][j], hi, lo);
#pragma unroll
                    for (int e = 0; e < 4; ++e) amax = fmaxf(amax, fabsf(v[rr][j][e]));
                    const size_t o = (size_t)((r0 + rr) * PW + s_col[j]) * S + s_g[j] * 8;
                    *reinterpret_cast<f16x4*>(Ph + o) = hi;
                    *reinterpret_cast<f16x4*>(Pl + o) = lo;
                }
            }
        }
    };

    // ---- fragments.  The weight slab of global step s lives in buffer s % 3 and is written during step s - 2, so that during
    // step s - 1 (after its opening barrier) the FIRST fragments of step s can already be read: fragment reads run one k-step
    // ahead of their MFMAs, across tap boundaries too.
    struct Frag { f16x8 ah, al, bh[NB], bl[NB]; };
    auto read_frag = [&](Frag& f, unsigned a_addr, unsigned b_addr) {
        f.ah = *reinterpret_cast<const f16x8*>(Ph + a_addr);
        f.al = *reinterpret_cast<const f16x8*>(Pl + a_addr);
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            f.bh[nb] = *reinterpret_cast<const f16x8*>(Wb + b_addr + (unsigned)nb * 32u * (unsigned)S);
            f.bl[nb] = *reinterpret_cast<const f16x8*>(Wb + b_addr + wplane + (unsigned)nb * 32u * (unsigned)S);
        }
    };
    auto a_addr_of = [&](int tap) -> unsigned {
        const int kh = tap / p.KW, kw = tap - kh * p.KW;
        return (unsigned)((wr + kh) * PW + wc0 + l31 + kw) * (unsigned)S + (unsigned)lhi * 16u;
    };
    auto b_addr_of = [&](int gstep) -> unsigned { return (unsigned)(gstep % 3) * wbuf + (unsigned)l31 * (unsigned)S + (unsigned)lhi * 16u; };

    f32x16 acc1[NB], acc2[NB];
    Frag f[2];
    // NTAPS consecutive taps of one pass (1 or 2: with an odd KS the fragment set holding a tap's first k-step alternates,
    // so taps are taken in pairs and every pair starts on set 0); `last` = the pass ends with them
    auto taps_body = [&](auto ntaps_c, int gstep0, int tap0, bool last) {
        constexpr int NTAPS = decltype(ntaps_c)::value;
#pragma unroll
        for (int j = 0; j < NTAPS; ++j) {
            const int gstep = gstep0 + j, tap = tap0 + j;
            const bool cross = !(last && j == NTAPS - 1);          // prefetch the next tap's first fragments
            load_w(gstep + 2);                                      // global loads: land under this tap's MFMAs
            const unsigned a_cur = a_addr_of(tap), b_cur = b_addr_of(gstep);
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                Frag& cur = f[(j * KS + ks) & 1];
                Frag& nxt = f[(j * KS + ks + 1) & 1];
                if (ks + 1 < KS) read_frag(nxt, a_cur + (ks + 1) * 32, b_cur + (ks + 1) * 32);
                else if (cross) read_frag(nxt, a_addr_of(tap + 1), b_addr_of(gstep + 1));
#pragma unroll
                for (int nb = 0; nb < NB; ++nb) {
                    acc1[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.ah, cur.bh[nb], acc1[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.ah, cur.bl[nb], acc2[nb], 0, 0, 0);
                    acc2[nb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(cur.al, cur.bh[nb], acc2[nb], 0, 0, 0);
                }
            }
            write_w(gstep + 2);     // buffer of step - 1: every wavefront left it at the previous barrier
            __syncthreads();
        }
    };

    // ---- persistent loop over output tiles (block launch, the weight prologue and the tile decode were ~1/4 of a tile's time
    // in the one-tile-per-block version); the weight slabs keep cycling across tiles
    load_w(0);
    write_w(0);
    load_w(1);
    write_w(1);
    unsigned emax = 0;
    int gstep = 0;
    for (int v = blockIdx.x; v < ntiles; v += gridDim.x) {
        {
            int t = v;
            const int tc = t % g.tiles_c;
            t /= g.tiles_c;
            const int tr = t % g.tiles_r;
            img = t / g.tiles_r;
            oh0 = tr * g.TR;
            ow0 = tc * g.TC;
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[nb][r] = acc2[nb][r] = 0.f;
        stage_patch(0);
        __syncthreads();                               // (also publishes the first two weight slabs on the first tile)
        read_frag(f[0], a_addr_of(0), b_addr_of(gstep));
        for (int pass = 0; pass < g.npass; ++pass) {
            int tap = 0;
            for (; tap + 2 <= NT - (NT & 1); tap += 2, gstep += 2)
                taps_body(std::integral_constant<int, 2>{}, gstep, tap, (NT & 1) == 0 && tap + 2 == NT);
            if (NT & 1) {
                taps_body(std::integral_constant<int, 1>{}, gstep, tap, true);
                ++gstep;
            }
            if (pass + 1 < g.npass) {     // next chunk of input channels: the patch is free (barrier above)
                stage_patch((pass + 1) * g.CC);
                __syncthreads();
                read_frag(f[0], a_addr_of(0), b_addr_of(gstep));
            }
        }

        // ---- epilogue: lane = output channel, registers = 16 of this wavefront's 32 pixels
        const int oh = oh0 + wr;
        // Interior wavefront tile (its 32 pixels all inside the row - every tile but a row's last): no per-pixel bounds test, and every
        // load / store is a buffer access - descriptor at the row's first pixel, the lane's (pixel group, channel) offset in the vector
        // operand, the pixel in the scalar one: one instruction per 4-byte store.  The general path below spends a 64-bit multiply-add,
        // a compare and an EXEC save / branch on each (round 5; same change as the interior epilogue of kernels_gemm_h1.hip).
        const bool interior = g.fast_epi && oh < p.OH && ow0 + wc0 + 32 <= p.OW;
        if (interior) {
            typedef __amdgpu_buffer_rsrc_t rsrc_t;
            const size_t pix0 = ((size_t)img * p.OH + oh) * p.OW + ow0 + wc0;
            const rsrc_t ry = __builtin_amdgcn_make_buffer_rsrc(p.y + pix0 * p.yld, 0, 0x7fffffff, 0x00020000);
            const rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.res ? p.res + pix0 * p.rld : p.y), 0, 0x7fffffff, 0x00020000);
            int lane_e = lane;
            asm volatile("" : "+v"(lane_e));                 // (keeps the offsets below out of the registers that live through the tap loops)
            const int l31e = lane_e & 31, lhie = lane_e >> 5;
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const int n = nb * 32 + l31e;
                if (n >= p.Ng) continue;
                const float bv = p.bias ? p.bias[n] : 0.f;
                const unsigned yoff = ((unsigned)(4 * lhie) * (unsigned)p.yld + (unsigned)n) * 4u;
                const unsigned roff = ((unsigned)(4 * lhie) * (unsigned)p.rld + (unsigned)n) * 4u;
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
                if (p.res) {
                    float rs[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        rs[r] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rr, (int)roff, (int)((unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.rld * 4u), 0));
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[r] += rs[r];
                }
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(o[r]), ry, (int)yoff, (int)((unsigned)((r & 3) + 8 * (r >> 2)) * (unsigned)p.yld * 4u), 2);
            }
        } else {
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n = nb * 32 + l31;
            if (n >= p.Ng || oh >= p.OH) continue;
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
            const size_t row0 = ((size_t)img * p.OH + oh) * p.OW;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int ow = ow0 + wc0 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (ow < p.OW) {
                    float vv = o[r];
                    if (p.res) vv += p.res[(row0 + ow) * p.rld + n];
                    __builtin_nontemporal_store(vv, &p.y[(row0 + ow) * p.yld + n]);
                }
            }
        }
        }
    }
    if ((emax >= 0x7f800000u || !(amax < 65504.f)) && p.range_flag) rd_raise_flag(p.range_flag);
}

static bool direct_geom(const ConvParams& p, DirectGeom& g) {
    // channels per pass: all of them up to 64, else the largest of 64 / 48 / 32 that divides Cin
    int cc = p.Cin;
    if (cc > 64) {
        cc = 0;
        for (int cand : {64, 48, 32})
            if (p.Cin % cand == 0) { cc = cand; break; }
        if (!cc) return false;
    }
    const int n32 = (p.Ng + 31) / 32 * 32;
    for (;;) {
        const int ccpad = (cc + 15) / 16 * 16;
        if (ccpad < 32) return false;
        g.CC = cc;
        g.npass = (p.Cin + cc - 1) / cc;
        g.S = ccpad * 2 + 16;
        g.N32 = n32;
        // tile shape: 8 x 32 or 4 x 64 output pixels, whichever wastes fewer of them
        long best = -1;
        for (int wpr : {1, 2}) {
            const int trr = 8 / wpr, tcc = 32 * wpr;
            const int nr = (p.OH + trr - 1) / trr, nc = (p.OW + tcc - 1) / tcc;
            const long padded = (long)nr * trr * nc * tcc;
            const int pw = tcc + p.KW - 1, pp = (trr + p.KH - 1) * pw;
            const size_t lds = (size_t)2 * pp * g.S + (size_t)6 * n32 * g.S;      // patch planes + three (hi, lo) weight slabs
            if (lds > 160 * 1024 - 512 || pw * (ccpad / 4) > 1024) continue;
            if (best < 0 || padded < best) {
                best = padded;
                g.WPR = wpr; g.TR = trr; g.TC = tcc; g.tiles_r = nr; g.tiles_c = nc; g.PW = pw; g.PP = pp;
            }
        }
        if (best >= 0) return true;
        // does not fit: halve the channel chunk if it still divides
        if (cc % 32 == 0 && p.Cin % (cc / 2) == 0) cc /= 2;
        else return false;
    }
}

// geometry the kernel can run
bool conv_direct_h3_supported(const ConvParams& p) {
    if (!p.wh || p.out_mode != OUT_NHWC || p.ascale || p.ln_g) return false;
    if (p.KH * p.KW < 2 || p.KH > 3 || p.KW > 3 || p.SH != 1 || p.SW != 1) return false;
    if (p.Ng > 96 || p.Cin % 8 != 0 || (p.xld % 4) != 0) return false;
    if (2 * ((p.Ng + 31) / 32 * 32) * (((p.Cin > 64 ? 64 : p.Cin) + 15) / 16 * 2) > 3 * 512) return false;   // weight chunks per thread <= 3
    DirectGeom g;
    return direct_geom(p, g);
}
// ... and where launch_conv_igemm_h3 routes to it: 3x3 layers.  The 2x2 stem layers (4 taps, K = 96 / 192) measured level with
// the implicit GEMM (58 / 62 vs 54 / 63 us) and stay there (RD_CONV_DIRECT=2 routes them here too, =0 nothing).
bool conv_direct_h3_applies(const ConvParams& p) {
    static const int mode = [] { const char* e = getenv("RD_CONV_DIRECT"); return e ? atoi(e) : 1; }();
    if (mode == 0) return false;
    if (p.KH * p.KW < 9 && mode != 2) return false;
    return conv_direct_h3_supported(p);
}

void launch_conv_direct_h3(const ConvParams& p, hipStream_t s) {
    DirectGeom g;
    if (!direct_geom(p, g)) return;
    const int nb = g.N32 / 32, ks = ((g.CC + 15) / 16 * 16) / 16;
    const size_t lds = (size_t)2 * g.PP * g.S + (size_t)6 * g.N32 * g.S;
    static const int n_cu = [] {
        int dev = 0, n = 256;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev);
        return n > 0 ? n : 256;
    }();
    static const int fast_epi = [] { const char* e = getenv("RD_DIRECT_FAST_EPI"); return e && e[0] == '0' ? 0 : 1; }();
    g.fast_epi = fast_epi;
    const long ntiles = (long)p.N * g.tiles_r * g.tiles_c;
    // persistent workgroups: as many as fit the chip at once (two per CU when LDS and the 128-VGPR single-block kernels allow)
    const int per_cu = (nb == 1 && lds <= 80 * 1024 - 256) ? 2 : 1;
    const long want = (long)n_cu * per_cu;
    const dim3 grid((unsigned)(ntiles < want ? ntiles : want)), block(512);
    static unsigned long long ok[3][5] = {};
#define RD_DIRECT_CASE(NB_, KS_)                                                                                  \
    if (nb == NB_ && ks == KS_) {                                                                                 \
        rd_allow_dynamic_lds((const void*)conv_direct_h3_kernel<NB_, KS_>, 160 * 1024, ok[NB_ - 1][KS_]);        \
        hipLaunchKernelGGL((conv_direct_h3_kernel<NB_, KS_>), grid, block, lds, s, p, g, (int)ntiles);            \
        return;                                                                                                   \
    }
    RD_DIRECT_CASE(1, 2) RD_DIRECT_CASE(1, 3) RD_DIRECT_CASE(1, 4)
    RD_DIRECT_CASE(2, 2) RD_DIRECT_CASE(2, 3) RD_DIRECT_CASE(2, 4)
    RD_DIRECT_CASE(3, 2) RD_DIRECT_CASE(3, 3) RD_DIRECT_CASE(3, 4)
#undef RD_DIRECT_CASE
}

}  // namespace rd
