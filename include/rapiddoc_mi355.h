/* rapiddoc_mi355 - C ABI of the MI355X-native page-inference engine (gfx950).
 *
 * Drop-in boundary for RapidDoc's per-page neural forward passes.  Each entry point replaces what one
 * `InferSession.__call__` of the reference does (paths relative to the RapidDoc source tree):
 *
 *   rd_det_forward       <- OCR det session:   rapid_doc/model/ocr/torch.py:171-192 (`maps`),
 *                           called from rapid_doc/model/ocr/rapid_ocr.py:528 (`text_detector.session`)
 *   rd_rec_forward       <- OCR rec session:   rapid_doc/model/ocr/torch.py:171-192 (`softmax(ctc_logits)`),
 *                           called from rapid_doc/model/ocr/rapid_ocr.py:443 (`text_recognizer.session`), plus the
 *                           argmax/max half of rapidocr's CTCLabelDecode (rapid_ocr.py:444-449)
 *   rd_backbone_forward  <- the PPHGNetV2-B4 backbone inside the PP-DocLayout ONNX graph:
 *                           rapid_doc/model/layout/rapid_layout_self/inference_engine/onnxruntime/main.py:61-78
 *                           (network definition: .../formula/rapid_formula_self/networks/backbones/rec_pphgnetv2.py:1445-1477)
 *   rd_preproc_resize_norm <- PPPreProcess: .../model_handler/pp_doclayout/pre_process.py:22-42
 *   rd_load_weights      <- `_load_state_dict` + `load_state_dict`: rapid_doc/model/ocr/torch.py:82-110
 *
 * Conventions: every pointer named *_dev is a DEVICE pointer on the handle's GPU (PyTorch-ROCm
 * `tensor.data_ptr()` works); images are NCHW float32 exactly like the numpy arrays the reference hands to
 * its sessions; `stream` is a hipStream_t passed as void* (NULL = default stream); functions return 0 on
 * success, non-zero on failure with the message in rd_last_error(handle).  `ws_dev` may be NULL, in which
 * case the handle owns (and grows) its own workspace.  One handle = one network on one device; a handle is
 * not thread-safe, different handles are independent.  There is no CPU fallback: rd_create fails if no
 * HIP device is present.
 */
#ifndef RAPIDDOC_MI355_H
#define RAPIDDOC_MI355_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rd_handle rd_handle;

/* rd_rec_forward flags */
#define RD_REC_UNFUSED_CTC 1  /* materialise logits, then row statistics (validation path)              */
#define RD_REC_WANT_SOFTMAX 2 /* also write softmax probabilities [B,T,C] (the reference session output) */
#define RD_REC_WANT_LOGITS 4  /* also write raw logits [B,T,C]                                          */

const char* rd_version(void);
/* model_kind: "ppocrv6_det" | "ppocrv6_rec" | "pphgnetv2_b4" | "pphgnetv2_b6_formula" | "ppformulanet_head".  NULL on failure -> rd_create_error(). */
rd_handle* rd_create(int device_id, const char* model_kind);
const char* rd_create_error(void);
void rd_destroy(rd_handle* h);
const char* rd_last_error(rd_handle* h);

/* HOST pointer to a .safetensors byte image using the reference's tensor names (a leading "model." is ignored). */
int rd_load_weights(rd_handle* h, const void* safetensors_image, size_t nbytes);

/* bytes of device workspace one call with this geometry needs (H is ignored for rec: always 48) */
int rd_query_workspace(rd_handle* h, int B, int H, int W, int flags, size_t* ws_bytes);

/* PP-OCRv6 det: x [B,3,H,W] (H, W multiples of 32) -> DB probability map [B,1,H,W] */
int rd_det_forward(rd_handle* h, const float* x_nchw_dev, int B, int H, int W, float* prob_b1hw_dev, void* ws_dev,
                   size_t ws_bytes, void* stream);

/* PP-OCRv6 rec: x [B,3,48,W] (any W >= 16) -> per time step (T = rd_rec_seq_len(W): two stride-2 convs and the (3,2)
 * average pool, W/8 when W is a multiple of 8) argmax class and its softmax probability; full_btc_dev is only written with RD_REC_WANT_SOFTMAX / RD_REC_WANT_LOGITS (else may be NULL). */
int rd_rec_forward(rd_handle* h, const float* x_nchw_dev, int B, int W, int32_t* idx_bt_dev, float* prob_bt_dev,
                   float* full_btc_dev, int flags, void* ws_dev, size_t ws_bytes, void* stream);
int rd_rec_num_classes(rd_handle* h);
/* The same network in two stages, for a pipeline that recognises the lines of many batches (rapid_ocr.py:430-449 loops over
 * `rec_batch_num` chunks and calls the session once per chunk): the neck and the CTC head of one chunk are ~25 launches on a few
 * thousand tokens - launch latency - so the chunks run only the backbone, each writing its tokens into ONE buffer, and the tail
 * runs once over all of them.  The results are those of rd_rec_forward chunk by chunk.
 *   rd_rec_backbone_forward: x [B,3,48,W] -> tokens [B][T = rd_rec_seq_len(W)][rd_rec_token_dim()] float32
 *   rd_rec_tail_forward:     tokens [n_tokens][dim] of n_lines text lines (any mix of lengths) -> idx / prob [n_tokens];
 *                            seg_dev = int32 [n_lines][2] (first token, tokens) per line, tokinfo_dev = int32 [n_tokens]
 *                            (position in its line | tokens of the line << 16), max_tokens = the longest line (< 32768)
 * rd_query_workspace(h, B, 0, W, RD_REC_STAGE_BACKBONE) resp. (h, n_lines, max_tokens, n_tokens, RD_REC_STAGE_TAIL) size them. */
#define RD_REC_STAGE_BACKBONE 8
#define RD_REC_STAGE_TAIL 16
int rd_rec_token_dim(rd_handle* h);
int rd_rec_backbone_forward(rd_handle* h, const float* x_nchw_dev, int B, int W, float* tokens_btd_dev, void* ws_dev, size_t ws_bytes,
                            void* stream);
int rd_rec_tail_forward(rd_handle* h, const float* tokens_dev, int n_tokens, int n_lines, int max_tokens, const int32_t* seg_dev,
                        const int32_t* tokinfo_dev, int32_t* idx_dev, float* prob_dev, void* ws_dev, size_t ws_bytes, void* stream);
/* The backbone stage over text lines of DIFFERENT reference padded widths in one launch.  The reference pads every line to the
 * width of its own chunk of `rec_batch_num` = 6 lines (rapid_ocr.py:404-449: imgW = int(48 * max w/h of the chunk)) and the
 * network's output for a line depends on that width (conv borders, SE pooling extent, LightSVTR attention over the padded columns),
 * so GPU-sized batches of mixed chunks must still give every line ITS width: x [B,3,48,W] holds line b in columns [0, w_b) (zeros
 * beyond), line_tab_dev = int32 [B][4] = (w_b, (w_b - 1) / 2 + 1, ((w_b - 1) / 2) / 2 + 1, first token of the line in tokens_dev);
 * line b writes rd_rec_seq_len(w_b) tokens there.  Results per line == rd_rec_backbone_forward on [1,3,48,w_b].
 * rd_query_workspace(h, B, 0, W, RD_REC_STAGE_BACKBONE | RD_REC_LINE_WIDTHS) sizes the workspace. */
#define RD_REC_LINE_WIDTHS 32
int rd_rec_backbone_forward_lines(rd_handle* h, const float* x_nchw_dev, int B, int W, const int32_t* line_tab_dev, float* tokens_dev,
                                  void* ws_dev, size_t ws_bytes, void* stream);
/* number of CTC time steps the rec network emits for input width W (stride-2 stem x2, then avg-pool (3,2)) */
int rd_rec_seq_len(int W);

/* PPHGNetV2-B4 (PP-DocLayout backbone): x [B,3,H,W] -> 4 NCHW feature maps, strides 4/8/16/32,
 * channels 128/512/1024/2048. */
int rd_backbone_forward(rd_handle* h, const float* x_nchw_dev, int B, int H, int W, float* const feats_dev[4],
                        void* ws_dev, size_t ws_bytes, void* stream);

/* PP-FormulaNet_plus encoder (PPHGNetV2_B6_Formula): x [B,C,H,W], C = 1 (grey, replicated to 3 channels like
 * rapid_doc/model/formula/rapid_formula_self/networks/backbones/rec_pphgnetv2.py:1626-1628) or 3; H, W multiples of 32.
 * enc_dev: [B, (H/32)*(W/32), 2048] = `last_hidden_state` of the reference backbone (:1629-1633).
 * model_kind "pphgnetv2_b6_formula". */
int rd_formula_encoder_forward(rd_handle* h, const float* x_nchw_dev, int B, int C, int H, int W, float* enc_dev,
                               void* ws_dev, size_t ws_bytes, void* stream);

/* PP-FormulaNet_plus decoder head (model_kind "ppformulanet_head"; load the full model's .safetensors / state dict, only
 * `head.*` tensors are used): greedy decode with KV cache of all B formulas at once.  Replaces PPFormulaNet_Head.forward ->
 * generate_export (rapid_doc/model/formula/rapid_formula_self/networks/heads/rec_ppformulanet_head.py:1054-1176,1367-1380).
 * enc_dev [B,S,2048] = encoder states; ids_dev [B][max_new_tokens+1] int64 (start token 0 first, pad 1 after EOS 2);
 * *n_cols = number of leading columns the reference would return (it stops when every sequence has produced EOS). */
int rd_formula_decode(rd_handle* h, const float* enc_dev, int B, int S, int max_new_tokens, int64_t* ids_dev, int32_t* n_cols,
                      void* stream);
/* largest max_new_tokens the loaded positional table supports (2560 for the shipped PP-FormulaNet_plus-M) */
int rd_formula_max_new_tokens(rd_handle* h);

/* u8 HWC (3 channels) device image -> resize to OHxOW -> (v*scale - mean[c]) / std[c] -> CHW float32.
 * interp: 1 = cv2.INTER_LINEAR, 2 = cv2.INTER_CUBIC, both in OpenCV's 8-bit fixed-point arithmetic (11-bit coefficients,
 * uint8 result) - bit-equal to oracle/cv2_ops.py; against cv2 itself parity is unpinned (package absent at build time). */
int rd_preproc_resize_norm(int device_id, const uint8_t* hwc_u8_dev, int H, int W, int OH, int OW, const float mean[3],
                           const float std[3], float scale, int interp, int swap_rb, float* out_chw_dev, void* stream);

/* The same for P images of one size in ONE launch: pages_u8_dev [P][H][W][3] -> out [P][3][OH][OW] (the reference's per-page
 * loop over PPPreProcess / DetPreProcess, rapid_layout_self/main.py:41-56, rapid_ocr.py:474-536). */
int rd_preproc_resize_norm_batch(int device_id, const uint8_t* pages_u8_dev, int P, int H, int W, int OH, int OW, const float mean[3],
                                 const float std[3], float scale, int interp, int swap_rb, float* out_nchw_dev, void* stream);

/* Text-line crops for one rec batch (replaces per-line cv2.warpPerspective + rapidocr resize_norm_img:
 * rapid_doc/utils/ocr_utils.py:494-536, rapid_doc/model/ocr/rapid_ocr.py:436-440).  pages_u8_dev: [P][H][W][3];
 * descs_dev: n device-resident rd_crop_desc; out: [n][3][out_h][out_w_padded] float32, zero right-padded. */
typedef struct rd_crop_desc {
    int32_t page;          /* page index in the batch */
    int32_t out_w;         /* resized width of this line (<= out_w_padded) */
    float crop_w, crop_h;  /* size of the rectified crop in page pixels */
    float m[9];            /* crop (x, y, 1) -> page (x, y, w) homography, row-major */
    int32_t rot90;         /* rotate the crop 90 deg CCW first (tall boxes) */
    int32_t pad_;
} rd_crop_desc;
int rd_crop_resize_norm_batch(int device_id, const uint8_t* pages_u8_dev, int P, int H, int W,
                              const rd_crop_desc* descs_dev, int n, int out_h, int out_w_padded, const float mean[3],
                              const float std[3], float scale, int swap_rb, float* out_nchw_dev, void* stream);

/* Reference-shaped text-line crops for one rec batch, in OpenCV's 8-bit arithmetic: cv2.warpPerspective(INTER_CUBIC,
 * BORDER_REPLICATE) of every line to its integer-sized uint8 crop (rapid_doc/utils/ocr_utils.py:494-536), np.rot90 for tall
 * crops, then rapidocr's resize_norm_img (linear resize to height out_h, /255, (x - 0.5) / 0.5, zero right-padding; called
 * from rapid_doc/model/ocr/rapid_ocr.py:436-440).  scratch_u8_dev: device buffer holding the packed uint8 crops
 * (desc.scratch_off, crop_w * crop_h * 3 bytes each); max_crop_pixels: largest crop_w * crop_h among the n lines.
 * out: [n][3][out_h][out_w_padded] float32.  swap_rb = 1 when the pages are RGB (rapidocr works on BGR). */
typedef struct rd_line_crop_desc {
    int32_t page;          /* page index in the batch */
    int32_t out_w;         /* resized width of this line (<= out_w_padded) */
    int32_t crop_w, crop_h;/* rectified crop size in page pixels (before the optional rotation) */
    int32_t rot90;         /* rotate the crop 90 deg CCW first (tall boxes) */
    int32_t scratch_off;   /* byte offset of this crop in scratch_u8_dev */
    double m[9];           /* crop (x, y, 1) -> page (X, Y, W) homography, row-major, float64 like cv2's */
} rd_line_crop_desc;
int rd_line_crops_batch(int device_id, const uint8_t* pages_u8_dev, int P, int H, int W, const rd_line_crop_desc* descs_dev,
                        int n, int64_t max_crop_pixels, uint8_t* scratch_u8_dev, int out_h, int out_w_padded, int swap_rb,
                        float* out_nchw_dev, void* stream);

/* The two stages of rd_line_crops_batch as separate calls, for callers whose text lines come from SEVERAL source images of
 * different sizes but are recognised together (the reference pools every line of a page batch per language before it sorts
 * and chunks them, rapid_doc/backend/pipeline/analyze_utils.py:216-252): rd_line_warp_batch once per source image array
 * (stage 1: cubic warp of its lines into the shared scratch buffer), then rd_line_resize_norm_batch once per rec batch
 * (stage 2: rot90 / linear resize to out_h / normalise / zero right-pad; reads the scratch buffer only). */
int rd_line_warp_batch(int device_id, const uint8_t* pages_u8_dev, int P, int H, int W, const rd_line_crop_desc* descs_dev, int n,
                       int64_t max_crop_pixels, uint8_t* scratch_u8_dev, void* stream);
int rd_line_resize_norm_batch(int device_id, const rd_line_crop_desc* descs_dev, int n, const uint8_t* scratch_u8_dev, int out_h,
                              int out_w_padded, int swap_rb, float* out_nchw_dev, void* stream);

/* CTC greedy decode on the device (replaces rapidocr CTCLabelDecode's per-line loop, called from
 * rapid_doc/model/ocr/rapid_ocr.py:444-449): idx / prob [B][T] as rd_rec_forward wrote them -> per line, at out + b * row_bytes:
 * int32 n_text_bytes, float32 confidence (the float32 np.mean of the kept max-probabilities, bit for bit), int32 n_kept,
 * int32 0, then the UTF-8 text.  char_table_dev: [n_classes][1 + max_len] bytes = (length, UTF-8 bytes) of every dictionary
 * entry, entry 0 = blank.  row_bytes >= 16 + T * max_len. */
int rd_ctc_collapse(int device_id, const int32_t* idx_bt_dev, const float* prob_bt_dev, int B, int T, const uint8_t* char_table_dev,
                    int max_len, int n_classes, uint8_t* out_dev, int row_bytes, void* stream);
/* The same over RAGGED lines as rd_rec_tail_forward leaves them: line b = seg_dev[2 b + 1] tokens (<= max_tokens <= 65535) starting at
 * token seg_dev[2 b] of idx_dev / prob_dev.  kept_cols_dev / kept_conf_dev (each may be NULL): uint16 / float32 [n_lines][max_tokens],
 * the time step and the probability of every kept character in order (n_kept of them, header field 2) - the `selection` and the
 * `conf_list` rapidocr's CTCLabelDecode hands to get_word_info / WordInfo when return_word_box is set (RapidDoc's patched
 * get_word_info and cal_ocr_word_box: rapid_doc/model/ocr/ocr_patch.py:264-389; table OCR default, analyze_utils.py:308). */
int rd_ctc_collapse_lines(int device_id, const int32_t* idx_dev, const float* prob_dev, int n_lines, const int32_t* seg_dev, int max_tokens,
                          const uint8_t* char_table_dev, int max_len, int n_classes, uint8_t* out_dev, int row_bytes, uint16_t* kept_cols_dev,
                          float* kept_conf_dev, void* stream);

/* DB post-process (HOST pointers, runs on the host like the reference's): probability maps [B][H][W] -> text boxes.
 * Replaces rapidocr DBPostProcess.__call__ as patched in rapid_doc/model/ocr/ocr_patch.py:223-241 (box_type "quad",
 * score_mode "fast"), called from rapid_doc/model/ocr/rapid_ocr.py:537-538.  src_hw[b] = (height, width) of the image
 * the boxes are scaled to.  out[b*max_out + i], i < n_out[b]: corners tl,tr,br,bl in source pixels + score. */
typedef struct rd_text_box {
    float pts[8];
    float score;
} rd_text_box;
int rd_db_postprocess(const float* prob_host, int B, int H, int W, const int32_t* src_hw, float thresh, float box_thresh,
                      float unclip_ratio, int use_dilation, int max_candidates, rd_text_box* out, int max_out,
                      int32_t* n_out, int n_threads);

/* The same post-process with NOTHING on the host (round 3): rows -> runs in raster order -> regions (union-find over
 * row-adjacent runs) -> convex hull of each region's row extremes -> min-area rectangle -> box_score_fast -> unclip / rescale /
 * filter_det_res, all on `stream`; the caller needs one device-to-host copy of n_out_dev and out_dev.  Boxes and their order
 * equal rd_db_postprocess's.  src_hw_dev: int32 [B][2] on the device.  n_out_dev: int32 [B + 1]; entry B is an overflow flag
 * (a page had more than max_runs bitmap runs: repeat that batch with rd_db_postprocess).  ws_dev: rd_db_boxes_workspace()
 * bytes of device scratch.  The scores are sums in double precision reduced in a fixed tree order; the host path adds the same
 * terms serially, so a candidate whose score sits within one ulp of box_thresh may be kept on one path and dropped on the other. */
size_t rd_db_boxes_workspace(int B, int H, int W, int max_runs, int max_candidates);
int rd_db_boxes_device(int device_id, const float* prob_dev, int B, int H, int W, const int32_t* src_hw_dev, float thresh, float box_thresh,
                       float unclip_ratio, int use_dilation, int max_candidates, int max_runs, void* ws_dev, size_t ws_bytes,
                       rd_text_box* out_dev, int max_out, int32_t* n_out_dev, void* stream);

/* PP-DocLayout post-process, rectangle mode (HOST pointers).  Replaces PPPostProcess.__call__ with
 * layout_shape_mode="rect": rapid_doc/model/layout/rapid_layout_self/model_handler/pp_doclayout/post_process.py:20-243.
 * boxes: [n][ncol] float32 rows (cls, score, x0, y0, x1, y1[, order...]) in original-image pixels (ncol 6, 7 or 8).
 * out: [n][6] kept rows, out_order[i] = the reference's 1-based "order" field, *n_out = kept count. */
typedef struct rd_layout_post_cfg {
    int32_t n_classes;
    int32_t image_index, formula_index; /* index of the "image" / "formula" label, -1 if absent          */
    int32_t thresh_is_dict;             /* 0: thresh[0] for every class; 1: thresh[class] (0.5 if missing) */
    const float* thresh;
    int32_t layout_nms;
    int32_t merge_kind;                 /* 0 none/"union", 1 "large", 2 "small", 3 per-class table         */
    const int8_t* merge_per_class;      /* [n_classes]: 0 union, 1 large, 2 small (merge_kind 3)            */
    int32_t unclip_kind;                /* 0 none, 1 one (w, h) ratio pair, 2 per-class pairs               */
    const float* unclip;                /* [2] or [n_classes][2]                                           */
    const uint8_t* unclip_present;      /* [n_classes] (unclip_kind 2)                                     */
} rd_layout_post_cfg;
int rd_layout_postprocess(const float* boxes, int n, int ncol, int img_w, int img_h, const rd_layout_post_cfg* cfg,
                          float* out, int32_t* out_order, int32_t* n_out);
/* The rows of `boxes` that reach the polygon stage of PPPostProcess.__call__ (post_process.py:213-218: after threshold / NMS /
 * big-image filter / containment merge / reading-order sort, before unclip and clip): sel_boxes [n][6], sel_src[i] = row index in
 * `boxes` (the detector's masks follow the boxes through those steps, :46-211).  Position i is out_order[k] - 1 of the row
 * rd_layout_postprocess writes for the same box. */
int rd_layout_postprocess_select(const float* boxes, int n, int ncol, int img_w, int img_h, const rd_layout_post_cfg* cfg,
                                 float* sel_boxes, int32_t* sel_src, int32_t* n_sel);

/* RT-DETR-family head operators - PREPARATION ONLY, PARITY UNPINNED (rapiddoc_amd/csrc/kernels_rtdetr.hip).  PP-DocLayout's neck / decoder
 * are ONNX files that are not part of the offline reference tree (rapid_layout_self/inference_engine/onnxruntime/main.py:61-78 only loads
 * them), so the graph cannot be read: these three entries implement operators whose definition does not depend on it, are tested against
 * fp64 restatements of their published definitions (tests/test_gpu_rtdetr_ops.py), and are wired into nothing.  Device pointers.
 *   rd_msdeform_attn   multi-scale deformable attention sampling (Deformable DETR eq. 3): value [B][S][H][D], shapes int32 [L][2] = (h, w),
 *                      level_start int32 [L], loc [B][Q][H][L][P][2] = (x, y) in [0, 1], attn [B][Q][H][L][P] -> out [B][Q][H * D];
 *                      bilinear samples as grid_sample(align_corners = False, zero padding); H * D <= 1024
 *   rd_topk_rows       per row of n scores: the k <= 1024 largest, descending, equal values in ascending index order, NaN above +inf
 *                      (the (queries x classes) selection in front of PPPostProcess, pp_doclayout/main.py:88-139)
 *   rd_encoder_layer   one post-norm transformer encoder layer (AIFI): MHA(q = k = x + pos, v = x) + residual + LayerNorm, FFN + residual +
 *                      LayerNorm over B sequences of T tokens; in_w [3 Dm][Dm] packed q | k | v; head_dim 16 or 32; `act` = rd activation
 *                      code (2 = GELU, 1 = ReLU); workspace of rd_encoder_layer_workspace(B * T, Dm, F) bytes */
int rd_msdeform_attn(int device_id, const float* value, const int32_t* shapes, const int32_t* level_start, const float* loc, const float* attn,
                     float* out, int B, int S, int H, int D, int Q, int L, int P, void* stream);
int rd_topk_rows(int device_id, const float* scores, int rows, int n, int k, float* out_vals, int32_t* out_idx, void* stream);
size_t rd_encoder_layer_workspace(int M, int Dm, int F);
int rd_encoder_layer(int device_id, const float* x, const float* pos, int B, int T, int Dm, int heads, int F, int act, const float* in_w,
                     const float* in_b, const float* out_w, const float* out_b, const float* ln1_g, const float* ln1_b, const float* w1,
                     const float* b1, const float* w2, const float* b2, const float* ln2_g, const float* ln2_b, float eps, float* out, void* ws,
                     size_t ws_bytes, void* stream);

/* Raster / polygon primitives of the polygon branch (HOST pointers; rapiddoc_amd/csrc/polygon_ops.cpp).  Each replaces one
 * OpenCV / shapely call of the reference (PARITY UNPINNED: neither library is available offline; restated from their published
 * algorithms):
 *   rd_find_external_contours  cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)        post_process.py:409
 *   rd_contour_area            cv2.contourArea                                                     :414
 *   rd_arc_length              cv2.arcLength(cnt, closed)                                          :415
 *   rd_approx_poly_dp          cv2.approxPolyDP(cnt, epsilon, closed)                              :416
 *   rd_min_area_rect_points    cv2.boxPoints(cv2.minAreaRect(points))  (corner order unspecified)  :553-554
 *   rd_polygon_area / rd_polygon_intersection_area   shapely Polygon.area / intersection(...).area :704-711
 *   rd_fill_poly               cv2.fillPoly(mask, [polygon], value), u8 single channel    rapid_doc/utils/model_utils.py:114
 * Points are [n][2] (x, y).  Return 0 = ok, 1 = bad arguments, 2 = output capacity (the needed counts are written). */
int rd_find_external_contours(const uint8_t* mask, int h, int w, int32_t* pts_out, int max_pts, int32_t* counts_out, int max_contours,
                              int32_t* n_contours, int32_t* n_pts);
double rd_contour_area(const int32_t* pts, int n);
double rd_arc_length(const int32_t* pts, int n, int closed);
int rd_approx_poly_dp(const int32_t* pts, int n, double epsilon, int closed, int32_t* out, int32_t* n_out);
int rd_min_area_rect_points(const float* pts, int n, float* out8);
double rd_polygon_area(const double* a, int na);
double rd_polygon_intersection_area(const double* a, int na, const double* b, int nb);
int rd_fill_poly(uint8_t* img, int h, int w, const int32_t* pts, int n, int value);

/* Host: chunk sizes of the recogniser's THROUGHPUT mode (the engine's own scheduling; rapidocr's fixed rec_batch_num chunks,
 * rapid_doc/model/ocr/rapid_ocr.py:430-440, are the `strict` mode of rapiddoc_amd.pipeline.PagePipeline).  wpad_sorted[i] = padded
 * width of the i-th line of the aspect-sorted list (int(48 * max(320/48, w/h)) rounded up to the width multiple); a chunk is a run of
 * that list padded to its last line.  rd_rec_plan_chunks cuts the list so that the summed rd_rec_chunk_cost (estimated microseconds
 * of a backbone forward: whole rounds of workgroup tiles on n_cu compute units per persistent kernel + a linear term + a per-forward
 * constant) is minimal; candidate sizes n_min, n_min + n_step, ... <= n_max, the last chunk any size.  Returns 0 on success. */
double rd_rec_chunk_cost(int n_lines, int wpad, int n_cu);
int rd_rec_plan_chunks(const int32_t* wpad_sorted, int n, int n_min, int n_max, int n_step, int n_cu, int32_t* sizes_out, int max_out,
                       int32_t* n_out);

/* Arithmetic of the dense layers of a network handle; every mode returns fp32 tensors with fp32-level error
 * (the reference runs the same layers through onnxruntime / torch fp32: rapid_doc/model/ocr/.../torch.py:58-76).
 *   "auto" (default)  split-fp16 matrix cores ((hi, lo) operand splitting, 3 fp16 MFMAs per product, fp32 accumulate,
 *                     measured error vs fp64 at or below the fp32 MFMA path's) for the fused PPLCNetV4 channel mixers,
 *                     the CTC head, 1x1 convs with K >= 96 and N >= 96, k x k convs with K >= 96 and N >= 24 and the
 *                     small-K stem layers; fp32 MFMA for the rest (narrow / short layers, M < 2048).
 *   "fp32"            native fp32 MFMA only.
 *   "h3"              every dense layer split (experimental; needs RD_PRECISION=h3 in the environment at rd_load_weights).
 * Split operands must stay inside the fp16 range (|v| < 65504).  The kernels never return a silently wrong answer:
 * they raise a flag instead, which rd_range_status() returns (1) and clears after synchronising `stream`; the caller
 * then switches the handle to "fp32" and repeats the forward call.  rd_range_status returns -1 on error. */
int rd_set_precision(rd_handle* h, const char* mode);
int rd_range_status(rd_handle* h, void* stream);

/* Cache behaviour of a handle since rd_create: per-shape plans built, hipGraph captures and hipGraph replays (each may be NULL).
 * A forward on a new (B, H, W, flags) builds a plan on the host; the second time a (plan, pointers, stream) triple shows up it is
 * captured, from the third on replayed.  bench.py reports the plans a non-repeating document stream builds inside its timed region. */
int rd_plan_stats(rd_handle* h, uint64_t* plans_built, uint64_t* graph_captures, uint64_t* graph_replays);

/* per-op HIP-event timing of the NEXT forward calls; rd_profile_json returns the last call's table as a JSON
 * array [{"name","kind","cfg","flops","bytes","ms"}, ...] owned by the handle. */
int rd_set_profiling(rd_handle* h, int on);
const char* rd_profile_json(rd_handle* h);

#ifdef __cplusplus
}
#endif
#endif
