// Kernel launch interface of the MI355X page-inference engine (gfx950 only).
// All activations are NHWC fp32; a "view" is (base pointer, channel stride ld) so that channel
// slices of a wider buffer (concat-free dense blocks, FPN concat) are addressed without copies.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdlib>
#include <vector>

namespace rd {

// Kernels that need more than 64 KB of dynamic LDS opt in once per (kernel, device): the attribute belongs to the device
// the call is made on, and one process may drive several GPUs.  `mask` is a static of the launcher, one bit per device.
inline void rd_allow_dynamic_lds(const void* kernel, size_t bytes, unsigned long long& mask) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(mask & bit)) {
        (void)hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        mask |= bit;
    }
}

// CUs the persistent matrix kernels size their grids for.  Their workgroups book a CU's LDS and registers completely (gemm_h1 2 x 80 KB,
// the mixers 146 - 148 KB), so while one of them covers all 256 CUs nothing of another stream - not even a bandwidth-bound depthwise
// launch that needs no matrix pipe - can start: RD_PERSISTENT_CUS=n leaves 256 - n CUs (spread over the XCDs by the dispatcher's round robin)
// to whatever else is in flight.  Default: all of them.
inline int rd_cu_budget(int n_cu) {
    static const int env = [] { const char* e = std::getenv("RD_PERSISTENT_CUS"); return e ? std::atoi(e) : 0; }();
    return env > 0 && env < n_cu ? env : n_cu;
}

enum Act : int { ACT_NONE = 0, ACT_RELU = 1, ACT_GELU = 2, ACT_SILU = 3, ACT_SIGMOID = 4, ACT_HSIG = 5, ACT_HSIG_PADDLE = 6 };
enum OutMode : int { OUT_NHWC = 0, OUT_DECONV2X2 = 1 };

// Dense convolution as implicit GEMM on fp32 MFMA:  Y[M, Ng] = im2col(X)[M, K] * W[Ng, K]^T
//   M = N*OH*OW, K = KH*KW*Cin (ci fastest), Ng = Cout (OUT_NHWC) or 4*Cout (OUT_DECONV2X2).
struct ConvParams {
    const float* x; int xld;
    int N, H, W, Cin;
    const float* w;      // [Ng][K]
    const uint16_t* wh;  // fp16x3 ("h3") mode: the same weights split into hi / lo fp16 matrices [Ng][Kp], Kp = ceil8(K)
    const uint16_t* wl;
    const float* bias;   // [Cout] or nullptr (BN already folded into w / bias)
    float* y; int yld;
    int OH, OW, Cout;
    int KH, KW, SH, SW, PT, PL;
    const float* res; int rld;      // added after the activation, same geometry as y
    const float* ascale;            // optional [N][Cin] multiplier applied to X rows on load (1x1 only)
    const float* ln_g; const float* ln_b;   // skinny path (M <= 32, K <= 512) only: LayerNorm(eps 1e-5) applied to every X row
                                            // before the product (fuses the pre-LN of a transformer decode step)
    int act, out_mode;
    int M, K, Ng;
    unsigned* range_flag = nullptr;   // split-fp16 kernels: raised when an activation leaves the fp16 range
    // single-accumulator split GEMM (kernels_gemm_h1.hip): fragment-ordered image of the power-of-two pre-scaled weights and the
    // inverse of that scale (prepare_gemm_h1_weights), or nullptr / 0
    const uint16_t* w1 = nullptr;
    float w1_inv = 0.f;
    // the small-M (<= 32 rows) fp32 path of launch_conv_igemm: only for layers whose M is a batch size BY CONSTRUCTION (the formula decoder's
    // linears); image layers never set it - their kernel is picked by the layer, not by how many rows a launch happens to hold
    int allow_skinny = 0;
    // one-accumulator direct 3x3 (kernels_conv3x3_h1.hip): slab-ordered fragment image of the pre-scaled weights + the inverse scale
    const uint16_t* w3 = nullptr;
    float w3_inv = 0.f;
    int fast_epi = 1;               // interior tiles of the split implicit-GEMM kernels store through buffer accesses (round 5; RD_CONV_FAST_EPI=0: A/B)
};
void launch_conv_igemm(const ConvParams& p, hipStream_t s);
bool skinny_gemm_applies(int M, int K);   // true when launch_conv_igemm will take the small-M path that can fuse ln_g / ln_b
// fp32-accurate variant on the fp16 matrix cores (3 MFMAs per product, see kernels_conv_h3.hip); needs p.wh / p.wl
bool gemm_h3_dma_applies(const ConvParams& p);
bool gemm_h3_dma_uses16(const ConvParams& p);   // the 16-wavefront kernel (short-K GELU layers, > 2^32-element tensors) or the 8-wavefront one
void launch_gemm_h3_dma(const ConvParams& p, hipStream_t s);
// single-accumulator split GEMM, 64 x 128 per wavefront, two 4-wavefront workgroups per CU (kernels_gemm_h1.hip); needs p.w1 / p.w1s
bool gemm_h1_shape_ok(int K, int cout);       // host-side: which layers get a weight image
bool gemm_h1_applies(const ConvParams& p);
void launch_gemm_h1(const ConvParams& p, hipStream_t s);
float prepare_gemm_h1_weights(const float* w, int N, int K, std::vector<uint16_t>& img);   // returns the inverse scale
void launch_conv_igemm_h3(const ConvParams& p, hipStream_t s);
// RT-DETR-family head operators, preparation only (kernels_rtdetr.hip; nothing in the engine calls them)
void launch_msdeform_attn(const float* value, const int32_t* shapes, const int32_t* start, const float* loc, const float* attn, float* out, int B,
                          int S, int H, int D, int Q, int L, int P, hipStream_t s);
void launch_topk_rows(const float* scores, int rows, int n, int k, float* out_vals, int32_t* out_idx, hipStream_t s);
// stem tail: 3x3 / stride 2 / pad 1 conv (+ bias + act3) -> 1x1 conv (+ bias + act4) in one kernel (kernels_stem34.hip, round 6).
// x NHWC [N][H][W][xld >= Cin]; y NHWC [N][OH][OW][yld >= N2], OH = (H - 1) / 2 + 1, OW likewise; pointers and row strides 16-byte aligned.
struct Stem34Params {
    const float* x; int xld;
    int N, H, W, Cin;
    float* y; int yld;
    int OH, OW, N1, N2;
    const uint16_t* w3; float w3_inv; const float* b3;     // slab-ordered image of the 3x3, bias padded with zeros to 32-wide blocks
    const uint16_t* w4; float w4_inv; const float* b4;     // fragment image of the 1x1 (input channels in the 3x3's C/D register order), bias [N2]
    int act3, act4;
    unsigned* range_flag = nullptr;
    int abl = 0;     // developer ablation bits (results are garbage): 1 no fragment reads / MFMAs, 2 no patch split / write, 4 no patch loads, 8 no epilogue, 16 no weight DMA
};
bool stem34_shape_ok(int cin, int n1, int n2);
bool stem34_enabled();                        // RD_STEM34=1 (default off: measured level with the two-kernel path, profiles/r6_stem34.txt)
void launch_stem34(const Stem34Params& p, hipStream_t s);
void prepare_stem34_weights(const float* w3, const float* w4, int Cin, int N1, int N2, std::vector<uint16_t>& img3, std::vector<uint16_t>& img4,
                            float inv[2]);
// direct 3x3 / stride 1 / pad 1, <= 96 output channels, one accumulator set, two workgroups per CU (kernels_conv3x3_h1.hip, round 6)
bool conv3x3_h1_shape_ok(int kh, int kw, int cin, int cout);   // host-side: which layers get a weight image
bool conv3x3_h1_applies(const ConvParams& p);
void launch_conv3x3_h1(const ConvParams& p, hipStream_t s);
float prepare_conv3x3_h1_weights(const float* w, int N, int Cin, std::vector<uint16_t>& img);   // returns the inverse scale
// direct 2x2 / 3x3 stride-1 convolution for narrow outputs (kernels_conv_direct_h3.hip); launch_conv_igemm_h3 dispatches to it
bool conv_direct_h3_supported(const ConvParams& p);   // geometry
bool conv_direct_h3_applies(const ConvParams& p);     // geometry + routing policy
void launch_conv_direct_h3(const ConvParams& p, hipStream_t s);
// small-K layers (padded K <= 256, <= 96 output channels), operands streamed from global memory (kernels_conv_stream_h3.hip)
bool conv_stream_h3_supported(const ConvParams& p);
bool conv_stream_h3_applies(const ConvParams& p);
void launch_conv_stream_h3(const ConvParams& p, hipStream_t s);
// host-side twin of conv_stream_h3_supported for the weight preparation (no ConvParams yet)
inline bool conv_stream_h3_shape_ok(int kh, int kw, int cin, int cout) {
    return cin % 4 == 0 && cin <= 64 && kh * kw * ((cin + 15) / 16 * 16) <= 256 && cout <= 96;
}
// a short human-readable tag of the tile configuration chosen for p (for the per-op profile)
const char* conv_igemm_config_name(const ConvParams& p);

// First conv of a network: 3x3 stride 2 pad 1, Cin = 3, reads the caller's NCHW (or NHWC u8/f32) image.
struct StemParams {
    const float* x;   // NCHW f32 [N,3,H,W]  (or [N,1,H,W] read three times when in_ch == 1: the reference repeats grey formulas)
    int N, H, W, in_ch;
    const float* w;   // [27][Cout] (kh,kw,ci major; co fastest)
    const float* bias;
    float* y; int yld; int OH, OW, Cout;
    int act;
};
void launch_stem_conv3x3s2(const StemParams& p, hipStream_t s);

// Fused stem front (kernels_stem_fused.hip): stem1 (3x3 s2) + stem2a + stem2b (2x2, pad right / bottom) + the 2x2 / s1 max-pool in
// one kernel, x NCHW -> cat NHWC [N][H2][W2][2 c1] = [pool | stem2b]; split-fp16 matrix cores; c1 in {24, 32, 48}
bool stem_fused_supported(int c1);
void prepare_stem_fused_weights(int c1, const float* w1_stem_layout, const float* w2a_folded, const float* w2b_folded, std::vector<uint16_t>& img);
// line_tab (optional): the recogniser's per-line width table (LineTab below) - every image n is treated as if it were only
// line_tab[4 n] columns wide: e / a / cat beyond its own half-resolution width are the zero padding the next layer expects
void launch_stem_fused(int c1, const float* x, int N, int H, int W, int in_ch, const uint16_t* wimg, const float* bias, float* y, int yld,
                       unsigned* range_flag, hipStream_t s, const int32_t* line_tab = nullptr);

// Per-line widths of a recogniser launch (REC_LINE_WIDTHS): text lines of DIFFERENT reference padded widths share one [B,3,48,W]
// tensor and every line is computed exactly as if it had been padded to its own width only (rapid_ocr.py:404-449 pads a line to
// its chunk-of-6's width; LightSVTR, the SE pooling and the conv borders all see that width).  int32 [B][4]:
//   0: w_in  the line's reference padded width (columns >= w_in of x are zero)
//   1: w2 = (w_in - 1) / 2 + 1 after stem1            2: w4 = (w2 - 1) / 2 + 1 after stem3 = the width of every block
//   3: first token of the line in the token buffer (the line has w4 / 2 tokens)
constexpr int kLineTabStride = 4;

struct DwParams {
    const float* x; int xld;
    int N, H, W, C;
    const float* w;     // [KH*KW][C]
    const float* bias;  // [C]
    float* y; int yld;
    int OH, OW, KH, KW, SH, SW, PT, PL;
    int act;
    const float* res; int rld;  // added after activation
    float* gap_partial;         // optional [N][gap_chunks][C] partial sums of the OUTPUT (SE fusion), or nullptr
    int gap_chunks;
    // ragged rows (the recogniser's batched tail): the tensor is [1][1][W = all tokens][C], token i sits at position
    // tokinfo[i] & 0xffff of a text line of tokinfo[i] >> 16 tokens and the kernel's horizontal taps stop at the line's ends
    const int32_t* tokinfo = nullptr;
    // per-image valid width (LineTab column 2; stride-1 'same' convs only): input columns >= line_w[n * line_w_stride] read as the
    // conv's zero padding and are left out of the SE partial sums
    const int32_t* line_w = nullptr; int line_w_stride = 0;
    int dbg = 0;                // developer (RD_DW_DBG): 1 no stores, 2 no loads - timing only
};
void launch_dwconv(const DwParams& p, hipStream_t s);
// number of per-image partial-sum chunks launch_dwconv writes to p.gap_partial for this geometry (0 = the fused
// global-average-pool is not available for it)
int dwconv_gap_chunks(const DwParams& p);
// kernels_dw_lds.hip: the LDS-DMA-staged 3x3 / stride 1 for the recogniser's 3 / 6 / 12-row maps; launch_dwconv routes to it
bool dwconv_lds_applies(const DwParams& p);
int dwconv_lds_gap_chunks(const DwParams& p);
void launch_dwconv_lds(const DwParams& p, hipStream_t s);
// ... and the 5x5 / 7x7 stride-1 kernel with one channel per lane (needs gap_partial == nullptr: checked at the launch, not by the planner)
bool dwconv_kxk_lds_applies(const DwParams& p);
void launch_dwconv_kxk_lds(const DwParams& p, hipStream_t s);

// 2x2 stride-1 max-pool over an input zero-padded by one pixel on the right/bottom (stem branch b)
void launch_maxpool2x2s1(const float* x, int xld, float* y, int yld, int N, int H, int W, int C, hipStream_t s);
// avg_pool2d(kernel (3,2), stride (3,2)) - rec height collapse
// line_tab: image n writes its w4 / 2 tokens at row line_tab[4 n + 3] of y (compact token buffer) instead of [n][OW]
void launch_avgpool3x2(const float* x, int xld, float* y, int yld, int N, int H, int W, int C, hipStream_t s, const int32_t* line_tab = nullptr);
// y[n, :, w >= line_w[n * stride], :] = 0 (the separate-kernel stem of the fp32 mode under a line table)
void launch_mask_cols(float* y, int yld, int N, int H, int W, int C, const int32_t* line_w, int stride, hipStream_t s);

// Squeeze-excite: deterministic two-stage global average pool, tiny FCs, then y = x * (alpha + s)
void launch_gap_partial(const float* x, int xld, int N, int HW, int C, float* partial, int chunks, hipStream_t s);
struct SeFcParams {
    const float* partial; int chunks; int N, C, Cr; float inv_hw;
    const float* w1; const float* b1;  // [Cr][C], [Cr]
    const float* w2; const float* b2;  // [C][Cr], [C]
    int gate;                          // ACT_HSIG (x/6+.5) or ACT_HSIG_PADDLE (.2x+.5)
    float* scale;                      // [N][C]
    const int32_t* line_w = nullptr; int line_w_stride = 0; int H = 0;   // per-image pooling extent H x line_w[n * stride] instead of 1 / inv_hw
};
void launch_se_fc(const SeFcParams& p, hipStream_t s);
bool det_head_tail_supported(int cin, int cmid, int cout);
void launch_det_head_tail(const float* x, int xld, int N, int H, int W, const float* w1, const float* b1, const float* w2, const float* b2,
                          float* y, hipStream_t s);
void launch_scale_channels(const float* x, int xld, float* y, int yld, const float* scale, float alpha,
                           int N, int HW, int C, hipStream_t s);

// y[n,h,w,:] (+)= x[n,h/f,w/f,:]   (nearest-neighbour upsample by integer factor f)
void launch_upsample(const float* x, int xld, float* y, int yld, int N, int H, int W, int C, int f, int accumulate,
                     hipStream_t s);

void launch_layernorm(const float* x, int xld, float* y, int yld, const float* g, const float* b, int M, int C,
                      float eps, hipStream_t s);
// qkv: [B*T][3*heads*hd] (q|k|v, head-major) -> o: [B*T][heads*hd]
// seg != nullptr: ragged batch - sequence b is seg[2b+1] tokens starting at token seg[2b]; T is then the longest one
int attention_h3_max_t();                     // lines up to this many tokens run on the matrix-core kernel, longer ones on the VALU kernel
bool attention_h3_applies(int T, int hd);     // kernels_attention_h3.hip: the same attention on the split-fp16 matrix cores
void launch_attention_h3(const float* qkv, float* o, int B, int T, int heads, int hd, float scale, hipStream_t s, const int32_t* seg);
void launch_attention(const float* qkv, float* o, int B, int T, int heads, int hd, float scale, hipStream_t s, const int32_t* seg = nullptr);

// y = a + b (same geometry, views)
void launch_add(const float* a, int ald, const float* b, int bld, float* y, int yld, int M, int C, hipStream_t s);

// CTC head statistics from logits: idx[m] = argmax_c z[m][c], prob[m] = 1 / sum_c exp(z[m][c]-max)
void launch_rowmax_softmax(const float* logits, int ld, int M, int C, int32_t* idx, float* prob, hipStream_t s);
// full softmax (only when the caller asks for the reference-shaped [B,T,C] probabilities)
// softmax rows; with idx / prob also numpy's argmax / max of the rows WRITTEN (lowest class among equal values)
void launch_row_softmax(const float* logits, int ld, float* out, int M, int C, hipStream_t s, int32_t* idx = nullptr, float* prob = nullptr);
// Fused CTC head: logits are never materialised.
struct CtcParams {
    const float* x; int xld;   // [M][K]
    const float* w;            // [C][K]
    const float* bias;         // [C]
    int M, K, C;
    float* part;               // workspace [M][nsplit][4] (max, sumexp, idx, pad)
    int nsplit;
    int32_t* idx; float* prob;
    // split-fp16 variant: W' [C][128] as (hi, lo) fp16; range_flag as in ConvParams
    const uint16_t* wh = nullptr; const uint16_t* wl = nullptr;
    unsigned* range_flag = nullptr;
};
void launch_ctc_head(const CtcParams& p, hipStream_t s);   // dispatches on p.wh
int ctc_head_nsplit(int M, int C, bool split_fp16);   // class splits the kernel that launch_ctc_head picks (p.wh set or not) wants

// layout conversion at the C-ABI boundary
void launch_nhwc_to_nchw(const float* x, int xld, float* y, int N, int H, int W, int C, hipStream_t s);
void launch_nchw_to_nhwc(const float* x, float* y, int yld, int N, int C, int H, int W, hipStream_t s);

// image pre-processing: u8 HWC -> bilinear/bicubic resize -> (v*scale - mean)/std -> NCHW f32
struct PreprocParams {
    const uint8_t* src; int H, W;      // HWC, 3 channels
    float* dst; int OH, OW;            // [3][OH][OW] plane of one batch element
    float mean[3], inv_std[3]; float scale;
    int interp;                         // 1 = bilinear, 2 = bicubic (a=-0.75, OpenCV convention)
    int swap_rb;
    int batch = 1;                      // images in this launch: image i reads src + i * src_stride, writes dst + i * dst_stride
    size_t src_stride = 0, dst_stride = 0;
};
void launch_preproc_resize_norm(const PreprocParams& p, hipStream_t s);

// Text-line crops for the recogniser, one launch per rec batch: for crop i, output pixel (ox, oy) of the
// 48 x out_w[i] resized line maps to crop coordinates, then through the 3x3 matrix m (crop -> page, i.e. the
// inverse of the reference's cv2.getPerspectiveTransform in utils/ocr_utils.py:494-536) to a bilinear sample
// of page `page`; columns >= out_w[i] are zero (rapidocr resize_norm_img right-pads with 0).
struct CropDesc {
    int32_t page;     // index into the page batch
    int32_t out_w;    // resized width (<= padded batch width)
    float crop_w, crop_h;
    float m[9];
    int32_t rot90;    // 1: the crop is rotated by 90 deg counter-clockwise first (h/w >= 1.5 rule, ocr_utils.py:531-535)
    int32_t pad_;
};
struct CropBatchParams {
    const uint8_t* pages; int H, W; size_t page_stride;   // [P][H][W][3] u8
    const CropDesc* descs; int n;
    float* dst; int OH, OWp;                               // [n][3][OH][OWp]
    float mean[3], inv_std[3]; float scale; int swap_rb;
};
void launch_crop_resize_norm_batch(const CropBatchParams& p, hipStream_t s);

// Reference-shaped text-line crops (kernels_image.hip): cubic perspective warp to the integer-sized uint8 crop
// (cv2.warpPerspective INTER_CUBIC / BORDER_REPLICATE, utils/ocr_utils.py:523-529), 90-degree rotation of tall crops, linear
// resize to height 48, /255, (x - 0.5) / 0.5, zero right-padding (rapidocr resize_norm_img) - two kernels per rec batch.
struct LineCropDesc {
    int32_t page;          // page index in the batch
    int32_t out_w;         // resized width (<= padded batch width)
    int32_t crop_w, crop_h;   // size of the rectified crop (before the optional rotation)
    int32_t rot90;         // np.rot90 the crop first (h / w >= 2, ocr_utils.py:533-535)
    int32_t scratch_off;   // byte offset of this crop's [crop_h][crop_w][3] uint8 image in the scratch buffer
    double m[9];           // crop pixel (x, y, 1) -> page (X, Y, W), row-major (the inverse of getPerspectiveTransform's matrix)
};
struct LineCropParams {
    const uint8_t* pages; int H, W; size_t page_stride;   // [P][H][W][3] u8
    const LineCropDesc* descs; int n;
    uint8_t* scratch; long max_crop_pixels;                // largest crop_w * crop_h of the batch (grid sizing)
    float* dst; int OH, OWp;                               // [n][3][OH][OWp]
    int swap_rb;
};
int launch_line_crops(const LineCropParams& p, hipStream_t s);
int launch_line_warp(const LineCropParams& p, hipStream_t s);          // stage 1 only (pages -> uint8 crops in the scratch buffer)
int launch_line_resize_norm(const LineCropParams& p, hipStream_t s);   // stage 2 only (scratch crops -> fp32 rec batch tensor)
// CTC greedy decode on the device: idx / prob [B][T] -> per line (row stride row_bytes): int32 n_text_bytes, float32
// confidence (numpy float32 mean of the kept probabilities), int32 n_kept, int32 0, UTF-8 text.  ctab: [n_classes][1 + max_len]
// bytes (length, then the entry's UTF-8 bytes).
// box_score_fast of the DB post-process candidates (kernels_image.hip), used by launch_db_boxes
int launch_db_scores(const float* prob, int B, int H, int W, const void* cand, const int32_t* n_cand, int max_cand, double* scores, hipStream_t s);
// the whole DB post-process on the device (kernels_dbpost.hip): include/rapiddoc_mi355.h rd_db_boxes_device
size_t db_boxes_workspace_bytes(int B, int H, int max_runs, int max_cand);
int launch_db_boxes(const float* prob, int B, int H, int W, const int32_t* src_hw_dev, float thresh, float box_thresh, float unclip_ratio,
                    int dilate, int max_cand, int max_runs, void* ws, size_t ws_bytes, void* out_boxes, int max_out, int32_t* n_out_dev,
                    hipStream_t s);
int launch_ctc_collapse(const int32_t* idx, const float* prob, int B, int T, const int32_t* seg, const uint8_t* ctab, int max_len, int n_classes,
                        uint8_t* out, int row_bytes, uint16_t* kept_cols, float* kept_conf, hipStream_t s);

}  // namespace rd

namespace rd {
void split_weights_h3(const float* w, int rows, int K, std::vector<uint16_t>& hi, std::vector<uint16_t>& lo);
// Fused PPLCNetV4 channel mixer for residual ("rep") blocks (rec_lcnetv4.py:226-236):
//   X' = X * gate            (optional SE gate, per sample and channel)
//   Y  = X' + W2 . GELU(W1 . X' + b1) + b2      W1: [2C][C], W2: [C][2C]  (BN folded)
// The [M][2C] hidden activation never leaves the CU.  C in {48, 96, 192}.
struct MixerParams {
    const float* x; int xld;
    float* y; int yld;
    int M, HW, C;
    const float* gate;   // [N][C] or nullptr
    const float* w1; const float* b1; const float* w2; const float* b2;
    // RD_PRECISION=h3 only: fp16 (hi, lo) splits of w1 [2C][C] and of w2 [C][2C] (hidden columns permuted per 32-chunk)
    const uint16_t* w1h = nullptr; const uint16_t* w1l = nullptr; const uint16_t* w2h = nullptr; const uint16_t* w2l = nullptr;
    unsigned* range_flag = nullptr;   // raised when a split operand leaves the fp16 range (|v| >= 65504)
    int dbg = 0;   // microbenchmark ablation bits (h3 kernel): 1 no GELU, 2 no weight streaming, 4 skip GEMM1, 8 skip GEMM2
    float ws_inv1 = 1.f, ws_inv2 = 1.f;   // ws kernel: inverse power-of-two scales of its weight stream image
    bool ws_pf = false;                   // ws kernel (C = 192): tile prefetch into the X registers, residual folded into the accumulator
    // resident-weights kernel (kernels_mixer_res.hip) only: the block's depthwise 3x3 / stride 1 / pad 1 computed in the tile load (round 6).
    // x is then the block's INPUT [N][dwH][dwW][xld]; dw_w [9][C] (tap-major) and dw_b [C] as Builder::dwconv folds them; dw_line_w as
    // DwParams::line_w (per-image valid width, or nullptr)
    const float* dw_w = nullptr; const float* dw_b = nullptr;
    int dwH = 0, dwW = 0;
    const int32_t* dw_line_w = nullptr; int dw_line_w_stride = 0;
};
bool mixer_fused_supported(int C);
bool mixer_res_fuses_dw(int C);      // kernels_mixer_res.hip: the block's depthwise 3x3 computed in the tile load (round 6)
void launch_mixer_fused_h3(const MixerParams& p, hipStream_t s);
void prepare_mixer_weights_h3(const float* w1, const float* w2, int C, std::vector<uint16_t>& w1h, std::vector<uint16_t>& w1l,
                              std::vector<uint16_t>& w2h, std::vector<uint16_t>& w2l);
void launch_mixer_fused(const MixerParams& p, hipStream_t s);
// round-2 weight-streaming variant (kernels_mixer_ws.hip; C = 96 / 192): activations in registers, LDS = weight stream only,
// persistent 8-wavefront workgroups.  p.w1h carries the stream image built by prepare_mixer_weights_ws.
bool mixer_ws_supported(int C);
bool mixer_ws_preferred(int C);   // where it measures faster than the round-1 kernel
void prepare_mixer_weights_ws(const float* w1, const float* w2, int C, std::vector<uint16_t>& img, float inv[2]);
bool mixer_ws_prefetch();             // RD_WS_PF (default 1): the engine launches the prefetching form of the ws kernel
void launch_mixer_fused_ws(const MixerParams& p, hipStream_t s);
void launch_mixer_debug(const MixerParams& p, int variant, hipStream_t s);
// resident-weights variant for the narrow blocks (kernels_mixer_res.hip; C = 96): all split weights in LDS, 16 independent
// wavefronts per CU.  p.w1h carries the fragment image built by prepare_mixer_weights_res.
bool mixer_res_supported(int C);
void prepare_mixer_weights_res(const float* w1, const float* w2, int C, std::vector<uint16_t>& img, float inv[2]);   // inv -> ws_inv1 / ws_inv2
void launch_mixer_fused_res(const MixerParams& p, hipStream_t s);
}  // namespace rd
