// extern "C" surface of librapiddoc_mi355.so - see include/rapiddoc_mi355.h for the contract.
#include "../../include/rapiddoc_mi355.h"

#include <mutex>
#include <string>

#include "engine.h"

namespace rd {
class FormulaDecoder;
FormulaDecoder* formula_decoder_create(int device, const void* blob, size_t nbytes);
void formula_decoder_destroy(FormulaDecoder* d);
int formula_decoder_decode(FormulaDecoder* d, const float* enc, int B, int S, int max_new, long long* ids, hipStream_t s);
int formula_decoder_max_new(FormulaDecoder* d);
}  // namespace rd

struct rd_handle {
    rd::Engine* eng = nullptr;           // convolutional networks (plan-based)
    rd::FormulaDecoder* dec = nullptr;   // "ppformulanet_head": autoregressive decoder
    int device = 0;
    std::string kind;
    std::string err;
    std::string prof;
};

static thread_local std::string g_create_err;

template <typename F>
static int guarded(rd_handle* h, F&& f) {
    if (!h || (!h->eng && h->kind != "ppformulanet_head")) return 2;
    try {
        f();
        h->err.clear();
        return 0;
    } catch (const std::exception& e) {
        h->err = e.what();
        return 1;
    }
}

extern "C" {

const char* rd_version(void) { return "rapiddoc_mi355 0.3 (gfx950; fp32 results, split-fp16 + fp32 MFMA kernels)"; }

rd_handle* rd_create(int device_id, const char* model_kind) {
    try {
        if (!model_kind) throw rd::Error("model_kind is NULL");
        auto* h = new rd_handle();
        h->device = device_id;
        h->kind = model_kind;
        if (h->kind == "ppformulanet_head") {
            int count = 0;
            if (hipGetDeviceCount(&count) != hipSuccess || count <= 0 || device_id < 0 || device_id >= count) {
                delete h;
                throw rd::Error("no HIP device available (MI355X required; there is no CPU fallback)");
            }
        } else {
            h->eng = new rd::Engine(device_id, model_kind);
        }
        g_create_err.clear();
        return h;
    } catch (const std::exception& e) {
        g_create_err = e.what();
        return nullptr;
    }
}
const char* rd_create_error(void) { return g_create_err.c_str(); }

void rd_destroy(rd_handle* h) {
    if (!h) return;
    delete h->eng;
    if (h->dec) rd::formula_decoder_destroy(h->dec);
    delete h;
}
const char* rd_last_error(rd_handle* h) { return h ? h->err.c_str() : "null handle"; }

int rd_load_weights(rd_handle* h, const void* img, size_t nbytes) {
    return guarded(h, [&] {
        if (h->kind == "ppformulanet_head") {
            RD_CHECK(!h->dec, "weights already loaded for this handle");
            h->dec = rd::formula_decoder_create(h->device, img, nbytes);
        } else {
            h->eng->load_weights(img, nbytes);
        }
    });
}

int rd_query_workspace(rd_handle* h, int B, int H, int W, int flags, size_t* ws_bytes) {
    return guarded(h, [&] {
        RD_CHECK(ws_bytes, "ws_bytes is NULL");
        RD_CHECK(h->eng, "this model kind owns its workspace");
        if (h->eng->kind() == "ppocrv6_rec" && !(flags & rd::REC_STAGE_TAIL)) H = 48;
        *ws_bytes = h->eng->workspace_bytes(B, H, W, flags);
    });
}

int rd_det_forward(rd_handle* h, const float* x, int B, int H, int W, float* prob, void* ws, size_t ws_bytes, void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->eng && h->eng->kind() == "ppocrv6_det", "handle is not a ppocrv6_det model");
        RD_CHECK(x && prob && B > 0, "null input/output");
        h->eng->run(B, H, W, 0, {(void*)x, (void*)prob}, ws, ws_bytes, (hipStream_t)stream);
    });
}

int rd_rec_forward(rd_handle* h, const float* x, int B, int W, int32_t* idx, float* prob, float* full, int flags, void* ws,
                   size_t ws_bytes, void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->eng && h->eng->kind() == "ppocrv6_rec", "handle is not a ppocrv6_rec model");
        RD_CHECK(x && idx && prob && B > 0, "null input/output");
        if (flags & (RD_REC_WANT_SOFTMAX | RD_REC_WANT_LOGITS)) RD_CHECK(full, "full_btc_dev is NULL");
        RD_CHECK(!((flags & RD_REC_WANT_SOFTMAX) && (flags & RD_REC_WANT_LOGITS)), "choose softmax OR logits");
        h->eng->run(B, 48, W, flags, {(void*)x, (void*)idx, (void*)prob, (void*)full}, ws, ws_bytes, (hipStream_t)stream);
    });
}
int rd_rec_token_dim(rd_handle* h) { return (h && h->eng && h->eng->kind() == "ppocrv6_rec") ? h->eng->rec_token_dim() : -1; }
int rd_rec_backbone_forward(rd_handle* h, const float* x, int B, int W, float* tokens, void* ws, size_t ws_bytes, void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->eng && h->eng->kind() == "ppocrv6_rec", "handle is not a ppocrv6_rec model");
        RD_CHECK(x && tokens && B > 0, "null input/output");
        h->eng->run(B, 48, W, rd::REC_STAGE_BACKBONE, {(void*)x, (void*)tokens}, ws, ws_bytes, (hipStream_t)stream);
    });
}
int rd_rec_backbone_forward_lines(rd_handle* h, const float* x, int B, int W, const int32_t* line_tab, float* tokens, void* ws, size_t ws_bytes,
                                  void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->eng && h->eng->kind() == "ppocrv6_rec", "handle is not a ppocrv6_rec model");
        RD_CHECK(x && tokens && line_tab && B > 0, "null input/output");
        h->eng->run(B, 48, W, rd::REC_STAGE_BACKBONE | rd::REC_LINE_WIDTHS, {(void*)x, (void*)tokens, (void*)line_tab}, ws, ws_bytes,
                    (hipStream_t)stream);
    });
}
int rd_rec_tail_forward(rd_handle* h, const float* tokens, int n_tokens, int n_lines, int max_tokens, const int32_t* seg,
                        const int32_t* tokinfo, int32_t* idx, float* prob, void* ws, size_t ws_bytes, void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->eng && h->eng->kind() == "ppocrv6_rec", "handle is not a ppocrv6_rec model");
        RD_CHECK(tokens && seg && tokinfo && idx && prob && n_tokens > 0 && n_lines > 0 && max_tokens > 0, "null input/output");
        h->eng->run(n_lines, max_tokens, n_tokens, rd::REC_STAGE_TAIL,
                    {(void*)tokens, (void*)idx, (void*)prob, nullptr, (void*)seg, (void*)tokinfo}, ws, ws_bytes, (hipStream_t)stream);
    });
}
int rd_rec_seq_len(int W) {
    if (W < 16) return 0;
    const int w1 = (W - 1) / 2 + 1, w2 = (w1 - 1) / 2 + 1;
    return (w2 - 2) / 2 + 1;
}
int rd_rec_num_classes(rd_handle* h) { return (h && h->eng) ? h->eng->n_classes() : -1; }

int rd_backbone_forward(rd_handle* h, const float* x, int B, int H, int W, float* const feats[4], void* ws, size_t ws_bytes,
                        void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->eng && h->eng->kind() == "pphgnetv2_b4", "handle is not a pphgnetv2_b4 model");
        RD_CHECK(x && feats && feats[0] && feats[1] && feats[2] && feats[3], "null input/output");
        h->eng->run(B, H, W, 0, {(void*)x, (void*)feats[0], (void*)feats[1], (void*)feats[2], (void*)feats[3]}, ws, ws_bytes,
                    (hipStream_t)stream);
    });
}

int rd_formula_encoder_forward(rd_handle* h, const float* x, int B, int C, int H, int W, float* enc, void* ws, size_t ws_bytes,
                               void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->eng && h->eng->kind() == "pphgnetv2_b6_formula", "handle is not a pphgnetv2_b6_formula model");
        RD_CHECK(x && enc && B > 0 && (C == 1 || C == 3), "null input/output or channel count not 1/3");
        h->eng->run(B, H, W, C == 1 ? 1 : 0, {(void*)x, (void*)enc}, ws, ws_bytes, (hipStream_t)stream);
    });
}

int rd_formula_decode(rd_handle* h, const float* enc, int B, int S, int max_new_tokens, int64_t* ids, int32_t* n_cols, void* stream) {
    return guarded(h, [&] {
        RD_CHECK(h->kind == "ppformulanet_head" && h->dec, "handle is not a loaded ppformulanet_head model");
        RD_CHECK(enc && ids && n_cols, "null input/output");
        *n_cols = rd::formula_decoder_decode(h->dec, enc, B, S, max_new_tokens, reinterpret_cast<long long*>(ids), (hipStream_t)stream);
    });
}
int rd_formula_max_new_tokens(rd_handle* h) { return (h && h->dec) ? rd::formula_decoder_max_new(h->dec) : -1; }

int rd_preproc_resize_norm(int device_id, const uint8_t* src, int H, int W, int OH, int OW, const float mean[3],
                           const float std[3], float scale, int interp, int swap_rb, float* out, void* stream) {
    if (!src || !out || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || (interp != 1 && interp != 2)) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::PreprocParams p{};
    p.src = src; p.H = H; p.W = W; p.dst = out; p.OH = OH; p.OW = OW;
    for (int i = 0; i < 3; ++i) { p.mean[i] = mean ? mean[i] : 0.f; p.inv_std[i] = 1.f / (std ? std[i] : 1.f); }
    p.scale = scale; p.interp = interp; p.swap_rb = swap_rb;
    rd::launch_preproc_resize_norm(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int rd_preproc_resize_norm_batch(int device_id, const uint8_t* src, int P, int H, int W, int OH, int OW, const float mean[3],
                                 const float std[3], float scale, int interp, int swap_rb, float* out, void* stream) {
    if (!src || !out || P <= 0 || P > 65535 || H <= 0 || W <= 0 || OH <= 0 || OW <= 0 || (interp != 1 && interp != 2)) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::PreprocParams p{};
    p.src = src; p.H = H; p.W = W; p.dst = out; p.OH = OH; p.OW = OW;
    for (int i = 0; i < 3; ++i) { p.mean[i] = mean ? mean[i] : 0.f; p.inv_std[i] = 1.f / (std ? std[i] : 1.f); }
    p.scale = scale; p.interp = interp; p.swap_rb = swap_rb;
    p.batch = P; p.src_stride = (size_t)H * W * 3; p.dst_stride = (size_t)3 * OH * OW;
    rd::launch_preproc_resize_norm(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int rd_crop_resize_norm_batch(int device_id, const uint8_t* pages, int P, int H, int W, const rd_crop_desc* descs, int n,
                              int out_h, int out_w_padded, const float mean[3], const float std[3], float scale, int swap_rb,
                              float* out, void* stream) {
    static_assert(sizeof(rd_crop_desc) == sizeof(rd::CropDesc), "rd_crop_desc layout");
    if (!pages || !descs || !out || P <= 0 || n < 0 || out_h <= 0 || out_w_padded <= 0) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::CropBatchParams p{};
    p.pages = pages; p.H = H; p.W = W; p.page_stride = (size_t)H * W * 3;
    p.descs = reinterpret_cast<const rd::CropDesc*>(descs); p.n = n;
    p.dst = out; p.OH = out_h; p.OWp = out_w_padded;
    for (int i = 0; i < 3; ++i) { p.mean[i] = mean ? mean[i] : 0.f; p.inv_std[i] = 1.f / (std ? std[i] : 1.f); }
    p.scale = scale; p.swap_rb = swap_rb;
    rd::launch_crop_resize_norm_batch(p, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int rd_line_crops_batch(int device_id, const uint8_t* pages, int P, int H, int W, const rd_line_crop_desc* descs, int n,
                        int64_t max_crop_pixels, uint8_t* scratch, int out_h, int out_w_padded, int swap_rb, float* out, void* stream) {
    static_assert(sizeof(rd_line_crop_desc) == sizeof(rd::LineCropDesc), "rd_line_crop_desc layout");
    if (!pages || !descs || !out || !scratch || P <= 0 || n < 0 || out_h <= 0 || out_w_padded <= 0 || max_crop_pixels <= 0) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::LineCropParams p{};
    p.pages = pages; p.H = H; p.W = W; p.page_stride = (size_t)H * W * 3;
    p.descs = reinterpret_cast<const rd::LineCropDesc*>(descs); p.n = n;
    p.scratch = scratch; p.max_crop_pixels = (long)max_crop_pixels;
    p.dst = out; p.OH = out_h; p.OWp = out_w_padded; p.swap_rb = swap_rb;
    if (rd::launch_line_crops(p, (hipStream_t)stream) != 0) return 1;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int rd_line_warp_batch(int device_id, const uint8_t* pages, int P, int H, int W, const rd_line_crop_desc* descs, int n,
                       int64_t max_crop_pixels, uint8_t* scratch, void* stream) {
    if (!pages || !descs || !scratch || P <= 0 || n < 0 || max_crop_pixels <= 0) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::LineCropParams p{};
    p.pages = pages; p.H = H; p.W = W; p.page_stride = (size_t)H * W * 3;
    p.descs = reinterpret_cast<const rd::LineCropDesc*>(descs); p.n = n;
    p.scratch = scratch; p.max_crop_pixels = (long)max_crop_pixels;
    if (rd::launch_line_warp(p, (hipStream_t)stream) != 0) return 1;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
int rd_line_resize_norm_batch(int device_id, const rd_line_crop_desc* descs, int n, const uint8_t* scratch, int out_h, int out_w_padded,
                              int swap_rb, float* out, void* stream) {
    if (!descs || !scratch || !out || n < 0 || out_h <= 0 || out_w_padded <= 0) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::LineCropParams p{};
    p.descs = reinterpret_cast<const rd::LineCropDesc*>(descs); p.n = n;
    p.scratch = const_cast<uint8_t*>(scratch);
    p.dst = out; p.OH = out_h; p.OWp = out_w_padded; p.swap_rb = swap_rb;
    if (rd::launch_line_resize_norm(p, (hipStream_t)stream) != 0) return 1;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int rd_ctc_collapse(int device_id, const int32_t* idx, const float* prob, int B, int T, const uint8_t* ctab, int max_len, int n_classes,
                    uint8_t* out, int row_bytes, void* stream) {
    if (!idx || !prob || !ctab || !out || B < 0 || T <= 0 || max_len <= 0 || n_classes <= 0) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    if (rd::launch_ctc_collapse(idx, prob, B, T, nullptr, ctab, max_len, n_classes, out, row_bytes, nullptr, nullptr, (hipStream_t)stream) != 0) return 1;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int rd_ctc_collapse_lines(int device_id, const int32_t* idx, const float* prob, int n_lines, const int32_t* seg, int max_tokens,
                          const uint8_t* ctab, int max_len, int n_classes, uint8_t* out, int row_bytes, uint16_t* kept_cols, float* kept_conf,
                          void* stream) {
    if (!idx || !prob || !seg || !ctab || !out || n_lines < 0 || max_tokens <= 0 || max_len <= 0 || n_classes <= 0) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    if (rd::launch_ctc_collapse(idx, prob, n_lines, max_tokens, seg, ctab, max_len, n_classes, out, row_bytes, kept_cols, kept_conf,
                                (hipStream_t)stream) != 0)
        return 1;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- RT-DETR-family head operators (preparation only, parity unpinned: include/rapiddoc_mi355.h) ----------------------------------
int rd_msdeform_attn(int device_id, const float* value, const int32_t* shapes, const int32_t* level_start, const float* loc, const float* attn,
                     float* out, int B, int S, int H, int D, int Q, int L, int P, void* stream) {
    if (!value || !shapes || !level_start || !loc || !attn || !out || B < 0 || S <= 0 || H <= 0 || D <= 0 || Q < 0 || L <= 0 || P <= 0 || H * D > 1024)
        return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::launch_msdeform_attn(value, shapes, level_start, loc, attn, out, B, S, H, D, Q, L, P, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

int rd_topk_rows(int device_id, const float* scores, int rows, int n, int k, float* out_vals, int32_t* out_idx, void* stream) {
    if (!scores || !out_vals || !out_idx || rows < 0 || n <= 0 || k <= 0 || k > 1024 || k > n) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    rd::launch_topk_rows(scores, rows, n, k, out_vals, out_idx, (hipStream_t)stream);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

size_t rd_encoder_layer_workspace(int M, int Dm, int F) {
    if (M <= 0 || Dm <= 0 || F <= 0) return 0;
    return ((size_t)M * Dm * 3 /* x + pos, attention output, post-norm-1 */ + (size_t)M * 3 * Dm /* q | k | v */ + (size_t)M * F) * sizeof(float);
}

// One post-norm transformer encoder layer over B sequences of T tokens (M = B * T rows of Dm):
//   a = MHA(q = k = x + pos, v = x);  y1 = LN1(x + a Wo^T + bo);  out = LN2(y1 + W2 act(W1 y1 + b1) + b2)
// in_w [3 Dm][Dm] / in_b [3 Dm] = the packed q | k | v projection (nn.MultiheadAttention's in_proj), head_dim = Dm / heads in {16, 32}.
// Composed from the engine's kernels: fp32-MFMA GEMMs (launch_conv_igemm on raw [N][K] weights), launch_attention, launch_layernorm.
int rd_encoder_layer(int device_id, const float* x, const float* pos, int B, int T, int Dm, int heads, int F, int act, const float* in_w,
                     const float* in_b, const float* out_w, const float* out_b, const float* ln1_g, const float* ln1_b, const float* w1,
                     const float* b1, const float* w2, const float* b2, const float* ln2_g, const float* ln2_b, float eps, float* out, void* ws,
                     size_t ws_bytes, void* stream) {
    const int M = B * T;
    if (!x || !in_w || !out_w || !w1 || !w2 || !ln1_g || !ln1_b || !ln2_g || !ln2_b || !out || !ws || B <= 0 || T <= 0 || heads <= 0 ||
        Dm % heads != 0 || Dm % 4 != 0 || F % 4 != 0 || ws_bytes < rd_encoder_layer_workspace(M, Dm, F))
        return 1;
    const int hd = Dm / heads;
    if (hd != 16 && hd != 32) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    hipStream_t s = (hipStream_t)stream;
    float* xp = (float*)ws;
    float* att = xp + (size_t)M * Dm;
    float* y1 = att + (size_t)M * Dm;
    float* qkv = y1 + (size_t)M * Dm;
    float* ff = qkv + (size_t)M * 3 * Dm;
    auto gemm = [&](const float* a, int K, const float* w, const float* b, int N, float* y, int yld, int a_act, const float* res) {
        rd::ConvParams p{};
        p.x = a; p.xld = K; p.N = 1; p.H = 1; p.W = M; p.Cin = K;
        p.w = w; p.bias = b; p.y = y; p.yld = yld; p.OH = 1; p.OW = M; p.Cout = N;
        p.KH = p.KW = p.SH = p.SW = 1;
        p.res = res; p.rld = N;
        p.act = a_act; p.out_mode = rd::OUT_NHWC;
        p.M = M; p.K = K; p.Ng = N;
        rd::launch_conv_igemm(p, s);
    };
    try {
        const float* qk_in = x;
        if (pos) {
            rd::launch_add(x, Dm, pos, Dm, xp, Dm, M, Dm, s);
            qk_in = xp;
        }
        gemm(qk_in, Dm, in_w, in_b, 2 * Dm, qkv, 3 * Dm, rd::ACT_NONE, nullptr);                                   // q | k from x + pos
        gemm(x, Dm, in_w + (size_t)2 * Dm * Dm, in_b ? in_b + 2 * Dm : nullptr, Dm, qkv + 2 * Dm, 3 * Dm, rd::ACT_NONE, nullptr);   // v from x
        rd::launch_attention(qkv, att, B, T, heads, hd, 1.0f / std::sqrt((float)hd), s);
        gemm(att, Dm, out_w, out_b, Dm, xp, Dm, rd::ACT_NONE, x);                                                   // x + a Wo^T + bo
        rd::launch_layernorm(xp, Dm, y1, Dm, ln1_g, ln1_b, M, Dm, eps, s);
        gemm(y1, Dm, w1, b1, F, ff, F, act, nullptr);
        gemm(ff, F, w2, b2, Dm, xp, Dm, rd::ACT_NONE, y1);
        rd::launch_layernorm(xp, Dm, out, Dm, ln2_g, ln2_b, M, Dm, eps, s);
    } catch (...) {
        return 1;
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

size_t rd_db_boxes_workspace(int B, int H, int W, int max_runs, int max_candidates) {
    (void)W;
    if (B <= 0 || H <= 0 || max_runs <= 0 || max_candidates <= 0) return 0;
    return rd::db_boxes_workspace_bytes(B, H, max_runs, max_candidates);
}
int rd_db_boxes_device(int device_id, const float* prob, int B, int H, int W, const int32_t* src_hw_dev, float thresh, float box_thresh,
                       float unclip_ratio, int use_dilation, int max_candidates, int max_runs, void* ws, size_t ws_bytes, rd_text_box* out,
                       int max_out, int32_t* n_out, void* stream) {
    if (!prob || !src_hw_dev || !ws || !out || !n_out || B < 0 || H <= 0 || W <= 0) return 1;
    if (hipSetDevice(device_id) != hipSuccess) return 1;
    if (rd::launch_db_boxes(prob, B, H, W, src_hw_dev, thresh, box_thresh, unclip_ratio, use_dilation, max_candidates, max_runs, ws, ws_bytes,
                            out, max_out, n_out, (hipStream_t)stream) != 0)
        return 1;
    return hipGetLastError() == hipSuccess ? 0 : 1;
}

// ---- developer micro-benchmarks (not part of the public header): time one kernel on caller-provided buffers
float rd_debug_time_mixer(int C, int M, int variant, int iters, float* x, float* y, float* w1, float* b1, float* w2, float* b2) {
    rd::MixerParams p{};
    p.x = x; p.xld = C; p.y = y; p.yld = C; p.M = M; p.HW = M; p.C = C; p.gate = nullptr;
    p.w1 = w1; p.b1 = b1; p.w2 = w2; p.b2 = b2;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    void* hbuf[4] = {nullptr, nullptr, nullptr, nullptr};
    int ws_abl = 0;                       // 1000 + 256 * bits: ablations of the ws kernel (C = 192, results are garbage)
    if (variant >= 1000) { ws_abl = (variant - 1000) & ~0xff; variant = 200 + ((variant - 1000) & 0xff); }
    if (variant >= 600) variant -= 300;   // (600+ = 300+: keeps the resident-weights variants clear of the 400..599 PF range)
    if (variant >= 300 && variant < 400) {  // resident-weights split mixer (kernels_mixer_res.hip)
        std::vector<float> hw1((size_t)2 * C * C), hw2((size_t)2 * C * C);
        (void)hipMemcpy(hw1.data(), w1, hw1.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hw2.data(), w2, hw2.size() * 4, hipMemcpyDeviceToHost);
        std::vector<uint16_t> img;
        float inv[2];
        rd::prepare_mixer_weights_res(hw1.data(), hw2.data(), C, img, inv);
        p.ws_inv1 = inv[0]; p.ws_inv2 = inv[1];
        (void)hipMalloc(&hbuf[0], img.size() * 2);
        (void)hipMemcpy(hbuf[0], img.data(), img.size() * 2, hipMemcpyHostToDevice);
        p.w1h = (const uint16_t*)hbuf[0];
    } else if (variant >= 200) {  // weight-streaming split mixer (kernels_mixer_ws.hip); 400+: the prefetching form (PF)
        const bool pf = variant >= 400 && variant < 600;
        if (pf) variant -= 200;
        p.ws_pf = pf;
        p.dbg = (variant - 200) | ws_abl;
        std::vector<float> hw1((size_t)2 * C * C), hw2((size_t)2 * C * C);
        (void)hipMemcpy(hw1.data(), w1, hw1.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hw2.data(), w2, hw2.size() * 4, hipMemcpyDeviceToHost);
        std::vector<uint16_t> img;
        float inv[2];
        rd::prepare_mixer_weights_ws(hw1.data(), hw2.data(), C, img, inv);
        p.ws_inv1 = inv[0]; p.ws_inv2 = inv[1];
        (void)hipMalloc(&hbuf[0], img.size() * 2);
        (void)hipMemcpy(hbuf[0], img.data(), img.size() * 2, hipMemcpyHostToDevice);
        p.w1h = (const uint16_t*)hbuf[0];
    } else if (variant >= 100) {  // fp16x3 mixer (+ ablation bits)
        p.dbg = variant - 100;
        std::vector<float> hw1((size_t)2 * C * C), hw2((size_t)2 * C * C);
        (void)hipMemcpy(hw1.data(), w1, hw1.size() * 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(hw2.data(), w2, hw2.size() * 4, hipMemcpyDeviceToHost);
        std::vector<uint16_t> v[4];
        rd::prepare_mixer_weights_h3(hw1.data(), hw2.data(), C, v[0], v[1], v[2], v[3]);
        for (int i = 0; i < 4; ++i) {
            (void)hipMalloc(&hbuf[i], v[i].size() * 2);
            (void)hipMemcpy(hbuf[i], v[i].data(), v[i].size() * 2, hipMemcpyHostToDevice);
        }
        p.w1h = (const uint16_t*)hbuf[0]; p.w1l = (const uint16_t*)hbuf[1];
        p.w2h = (const uint16_t*)hbuf[2]; p.w2l = (const uint16_t*)hbuf[3];
    }
    auto go = [&] {
        if (variant >= 300 && !p.ws_pf) rd::launch_mixer_fused_res(p, nullptr);
        else if (variant >= 200) rd::launch_mixer_fused_ws(p, nullptr);
        else if (variant >= 100) rd::launch_mixer_fused_h3(p, nullptr);
        else rd::launch_mixer_debug(p, variant, nullptr);
    };
    go();
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) go();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    for (void* b : hbuf) if (b) (void)hipFree(b);
    return ms / iters;
}
// h1 image of device weights w [N][K] (single-accumulator split GEMM, kernels_gemm_h1.hip): uploaded into fresh device buffers
static bool debug_h1_image(const float* w_dev, int N, int K, void** img_dev, float* inv) {
    if (!rd::gemm_h1_shape_ok(K, N)) return false;
    std::vector<float> hw((size_t)N * K);
    if (hipMemcpy(hw.data(), w_dev, hw.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return false;
    std::vector<uint16_t> img;
    *inv = rd::prepare_gemm_h1_weights(hw.data(), N, K, img);
    if (hipMalloc(img_dev, img.size() * 2) != hipSuccess) return false;
    (void)hipMemcpy(*img_dev, img.data(), img.size() * 2, hipMemcpyHostToDevice);
    return true;
}
float rd_debug_time_gemm(int M, int K, int N, int act, int iters, float* x, float* w, float* b, float* y, void* wh, void* wl) {
    rd::ConvParams p{};
    p.wh = (const uint16_t*)wh; p.wl = (const uint16_t*)wl;
    p.x = x; p.xld = K; p.N = 1; p.H = 1; p.W = M; p.Cin = K; p.w = w; p.bias = b; p.y = y; p.yld = N;
    p.OH = 1; p.OW = M; p.Cout = N; p.KH = p.KW = p.SH = p.SW = 1; p.act = act; p.out_mode = rd::OUT_NHWC;
    p.M = M; p.K = K; p.Ng = N;
    void* img = nullptr;
    float inv = 0.f;
    if (wh && debug_h1_image(w, N, K, &img, &inv)) { p.w1 = (const uint16_t*)img; p.w1_inv = inv; }   // as the engine prepares it
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto go = [&] { if (wh) rd::launch_conv_igemm_h3(p, nullptr); else rd::launch_conv_igemm(p, nullptr); };
    go();
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) go();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (img) (void)hipFree(img);
    return ms / iters;
}
// developer entry: the single-accumulator split GEMM alone, with strides, residual and the range flag:
// y[M][yld] = act(x[M][xld(K used)] * w[N][K]^T + b) + res[M][rld].  Returns ms per launch, or -1 when the kernel does not take the shape.
// *range_out (optional) receives 1 when the kernel raised its range flag.
float rd_debug_gemm_h1(int M, int K, int N, int act, int iters, float* x, int xld, float* w, float* b, float* res, int rld, float* y, int yld,
                       int* range_out) {
    rd::ConvParams p{};
    p.x = x; p.xld = xld; p.N = 1; p.H = 1; p.W = M; p.Cin = K; p.w = w; p.bias = b; p.y = y; p.yld = yld;
    p.OH = 1; p.OW = M; p.Cout = N; p.KH = p.KW = p.SH = p.SW = 1; p.act = act; p.out_mode = rd::OUT_NHWC;
    p.res = res; p.rld = rld;
    p.M = M; p.K = K; p.Ng = N;
    void* img = nullptr;
    float inv = 0.f;
    if (!debug_h1_image(w, N, K, &img, &inv)) return -1.f;
    p.w1 = (const uint16_t*)img; p.w1_inv = inv;
    unsigned* flag = nullptr;
    (void)hipHostMalloc((void**)&flag, sizeof(unsigned), hipHostMallocMapped);
    *flag = 0;
    p.range_flag = flag;
    float ms = -1.f;
    if (rd::gemm_h1_applies(p)) {
        hipEvent_t e0, e1;
        (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        rd::launch_gemm_h1(p, nullptr);
        (void)hipEventRecord(e0, nullptr);
        for (int i = 0; i < iters; ++i) rd::launch_gemm_h1(p, nullptr);
        (void)hipEventRecord(e1, nullptr);
        (void)hipEventSynchronize(e1);
        (void)hipEventElapsedTime(&ms, e0, e1);
        (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
        ms = iters > 0 ? ms / iters : 0.f;
    }
    (void)hipDeviceSynchronize();
    if (range_out) *range_out = (int)*flag;
    (void)hipHostFree(flag);
    (void)hipFree(img);
    return ms;
}

// developer entry: one dense convolution on prepared operands (x NHWC fp32 [N][H][W][Cin]; w folded [Cout][K], k = (kh*KW+kw)*Cin+ci;
// wh / wl its fp16 split with rows padded to Kp = ceil32(K), or null for the fp32 MFMA kernels; y NHWC [N][OH][OW][Cout]).
// Returns ms per launch (iters timed launches after one untimed).  *used_direct: in = 1 forces the direct k x k kernel when it
// supports the geometry, in = 2 the small-K streaming kernel; out = 1 / 2 when the direct / streaming kernel ran.
float rd_debug_conv(int N, int H, int W, int Cin, int Cout, int KH, int KW, int S, int PT, int PL, int PB, int PR, int act, int iters,
                    float* x, float* w, void* wh, void* wl, float* bias, float* res, float* y, int* used_direct) {
    rd::ConvParams p{};
    p.x = x; p.xld = Cin; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.w = w; p.bias = bias; p.y = y; p.yld = Cout;
    p.wh = (const uint16_t*)wh; p.wl = (const uint16_t*)wl;
    p.OH = (H + PT + PB - KH) / S + 1; p.OW = (W + PL + PR - KW) / S + 1; p.Cout = Cout;
    p.KH = KH; p.KW = KW; p.SH = p.SW = S; p.PT = PT; p.PL = PL; p.act = act; p.out_mode = rd::OUT_NHWC;
    p.res = res; p.rld = Cout;
    p.M = N * p.OH * p.OW; p.K = KH * KW * Cin; p.Ng = Cout;
    // the one-accumulator 3x3 kernel needs its own weight image, prepared here the way the engine prepares it (RD_CONV3X3_H1=0 or
    // *used_direct = 1 / 2 on entry: the older kernels).  Reported as *used_direct = 3.
    void* img3 = nullptr;
    if (wh && !(used_direct && *used_direct) && rd::conv3x3_h1_shape_ok(KH, KW, Cin, Cout)) {
        std::vector<float> hw((size_t)Cout * p.K);
        if (hipMemcpy(hw.data(), w, hw.size() * 4, hipMemcpyDeviceToHost) == hipSuccess) {
            std::vector<uint16_t> img;
            const float inv = rd::prepare_conv3x3_h1_weights(hw.data(), Cout, Cin, img);
            if (hipMalloc(&img3, img.size() * 2) == hipSuccess) {
                (void)hipMemcpy(img3, img.data(), img.size() * 2, hipMemcpyHostToDevice);
                p.w3 = (const uint16_t*)img3;
                p.w3_inv = inv;
            }
        }
    }
    const bool c3 = wh && rd::conv3x3_h1_applies(p);
    const bool force = used_direct && *used_direct == 1 && wh && rd::conv_direct_h3_supported(p);
    const bool force_stream = used_direct && *used_direct == 2 && wh && rd::conv_stream_h3_supported(p);
    if (used_direct) *used_direct = c3 ? 3 : force_stream ? 2 : (force || (wh && !rd::conv_stream_h3_applies(p) && rd::conv_direct_h3_applies(p))) ? 1
                                    : (wh && rd::conv_stream_h3_applies(p)) ? 2 : 0;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    auto go = [&] {
        if (force_stream) rd::launch_conv_stream_h3(p, nullptr);
        else if (force) rd::launch_conv_direct_h3(p, nullptr);
        else if (wh) rd::launch_conv_igemm_h3(p, nullptr);
        else rd::launch_conv_igemm(p, nullptr);
    };
    go();
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) go();
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (img3) (void)hipFree(img3);
    return iters > 0 ? ms / iters : 0.f;
}

// developer entry: the fused stem tail (kernels_stem34.hip).  x NHWC fp32 [N][H][W][xld >= Cin]; w3 [N1][9 Cin] with k = (kh * 3 + kw) * Cin + ci,
// w4 [N2][N1], biases [N1] / [N2] (device pointers); y NHWC [N][OH][OW][yld >= N2].  Images are prepared here the way the engine prepares them.
// Returns ms per launch, -1 when the shape is not covered.
float rd_debug_stem34(int N, int H, int W, int Cin, int xld, int N1, int N2, int yld, int act3, int act4, int iters, float* x, float* w3, float* b3,
                      float* w4, float* b4, float* y, unsigned* range_flag) {
    if (!rd::stem34_shape_ok(Cin, N1, N2)) return -1.f;
    std::vector<float> h3((size_t)N1 * 9 * Cin), h4((size_t)N2 * N1), hb3((size_t)((N1 + 31) / 32) * 32, 0.f);
    if (hipMemcpy(h3.data(), w3, h3.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1.f;
    if (hipMemcpy(h4.data(), w4, h4.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1.f;
    if (hipMemcpy(hb3.data(), b3, (size_t)N1 * 4, hipMemcpyDeviceToHost) != hipSuccess) return -1.f;
    std::vector<uint16_t> i3, i4;
    float inv[2];
    rd::prepare_stem34_weights(h3.data(), h4.data(), Cin, N1, N2, i3, i4, inv);
    void *d3 = nullptr, *d4 = nullptr, *db3 = nullptr;
    if (hipMalloc(&d3, i3.size() * 2) != hipSuccess || hipMalloc(&d4, i4.size() * 2) != hipSuccess || hipMalloc(&db3, hb3.size() * 4) != hipSuccess) return -1.f;
    (void)hipMemcpy(d3, i3.data(), i3.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(d4, i4.data(), i4.size() * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(db3, hb3.data(), hb3.size() * 4, hipMemcpyHostToDevice);
    rd::Stem34Params p{};
    p.x = x; p.xld = xld; p.N = N; p.H = H; p.W = W; p.Cin = Cin; p.y = y; p.yld = yld;
    p.OH = (H - 1) / 2 + 1; p.OW = (W - 1) / 2 + 1; p.N1 = N1; p.N2 = N2;
    p.w3 = (const uint16_t*)d3; p.w3_inv = inv[0]; p.b3 = (const float*)db3;
    p.w4 = (const uint16_t*)d4; p.w4_inv = inv[1]; p.b4 = b4;
    p.act3 = act3; p.act4 = act4; p.range_flag = range_flag;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd::launch_stem34(p, nullptr);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) rd::launch_stem34(p, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(d3); (void)hipFree(d4); (void)hipFree(db3);
    return iters > 0 ? ms / iters : 0.f;
}

// developer entry: one depthwise convolution through launch_dwconv (x NHWC fp32 [N][H][W][C], w [KH*KW][C], bias [C] or null,
// res NHWC or null, line_w int32 [N] valid widths or null, gap = [N][chunks][C] partial sums of the output or null).
// *gap_chunks receives the chunk count the launcher uses for this geometry.  Returns ms per launch.
float rd_debug_dwconv(int N, int H, int W, int C, int K, int SH, int act, int iters, float* x, float* w, float* bias, float* res, float* y,
                      const int32_t* line_w, float* gap, int* gap_chunks) {
    rd::DwParams p{};
    p.x = x; p.xld = C; p.N = N; p.H = H; p.W = W; p.C = C; p.w = w; p.bias = bias; p.y = y; p.yld = C;
    p.KH = p.KW = K; p.SH = SH; p.SW = 1; p.PT = p.PL = K / 2;
    p.OH = (H + 2 * p.PT - K) / SH + 1; p.OW = W; p.act = act; p.res = res; p.rld = C;
    const int chunks = rd::dwconv_gap_chunks(p);
    if (gap_chunks) *gap_chunks = chunks;
    p.gap_partial = chunks > 0 ? gap : nullptr; p.gap_chunks = chunks;
    p.line_w = line_w; p.line_w_stride = 1;
    if (gap_chunks && rd::dwconv_kxk_lds_applies(p)) *gap_chunks = -1;      // (tests: the one-channel-per-lane 5x5 / 7x7 kernel takes this call)
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd::launch_dwconv(p, nullptr);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) rd::launch_dwconv(p, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return iters > 0 ? ms / iters : 0.f;
}

// developer timing of the fused CTC head on prepared weights: wp = W' [C][128] fp32 (bias in column K), wh / wl its fp16 split
// (null: fp32 MFMA kernel); part = workspace of M * 64 * 4 floats.  Returns ms per launch; *nsplit_out = the split count used.
float rd_debug_time_ctc(int M, int K, int Ccls, int iters, float* x, float* wp, void* wh, void* wl, float* part, int32_t* idx, float* prob,
                        int nsplit_override, int* nsplit_out) {
    rd::CtcParams p{};
    p.x = x; p.xld = K; p.w = wp; p.M = M; p.K = K; p.C = Ccls; p.part = part;
    p.nsplit = nsplit_override > 0 ? nsplit_override : rd::ctc_head_nsplit(M, Ccls, wh != nullptr);
    p.idx = idx; p.prob = prob;
    p.wh = (const uint16_t*)wh; p.wl = (const uint16_t*)wl;
    if (nsplit_out) *nsplit_out = p.nsplit;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    rd::launch_ctc_head(p, nullptr);
    (void)hipEventRecord(e0, nullptr);
    for (int i = 0; i < iters; ++i) rd::launch_ctc_head(p, nullptr);
    (void)hipEventRecord(e1, nullptr);
    (void)hipEventSynchronize(e1);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return ms / iters;
}

int rd_set_precision(rd_handle* h, const char* mode) {
    return guarded(h, [&] {
        RD_CHECK(h->eng, "precision modes apply to the network engines");
        const std::string m = mode ? mode : "";
        RD_CHECK(m == "auto" || m == "fp32" || m == "h3", "precision must be auto, fp32 or h3");
        h->eng->set_precision(m == "h3" ? rd::Engine::PREC_H3 : m == "fp32" ? rd::Engine::PREC_FP32 : rd::Engine::PREC_AUTO);
    });
}
int rd_range_status(rd_handle* h, void* stream) {
    int out = 0;
    const int rc = guarded(h, [&] { if (h->eng) out = h->eng->take_range_flag((hipStream_t)stream); });
    return rc != 0 ? -1 : out;
}

int rd_plan_stats(rd_handle* h, uint64_t* plans_built, uint64_t* graph_captures, uint64_t* graph_replays) {
    if (!h || !h->eng) return 1;
    if (plans_built) *plans_built = h->eng->plans_built();
    if (graph_captures) *graph_captures = h->eng->graph_captures();
    if (graph_replays) *graph_replays = h->eng->graph_replays();
    return 0;
}

int rd_set_profiling(rd_handle* h, int on) {
    return guarded(h, [&] { if (h->eng) h->eng->set_profiling(on != 0); });
}
const char* rd_profile_json(rd_handle* h) {
    if (!h || !h->eng) return "[]";
    h->prof = h->eng->profile_json();
    return h->prof.c_str();
}

}  // extern "C"
