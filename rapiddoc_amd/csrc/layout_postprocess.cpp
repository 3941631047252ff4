// PP-DocLayout post-process, rectangle mode: score threshold -> class-aware greedy NMS -> page-sized "image" filter ->
// containment merge -> reading-order sort -> unclip -> clip to the page.  Host-side C++: in the reference this is
// numpy + O(n^2) Python loops per page (rapid_doc/model/layout/rapid_layout_self/model_handler/pp_doclayout/
// post_process.py:20-243 `PPPostProcess.__call__`, :948-979 `nms`, :981-1022 `is_contained`/`check_containment`,
// :611-662 `unclip_boxes`, :566-608 `restructured_boxes`), which becomes the bottleneck once the network takes ~ms.
// Arithmetic is done in float32 exactly where numpy does it in float32 so that kept boxes / coordinates are
// bit-identical to the reference (pinned by tests/golden/layout_post_seed*.json, minted from the reference class).
// The polygon branch (masks -> polygon_points, post_process.py:213-218,425-535) runs between steps 5 and 6 on the Python side
// (rapiddoc_amd/layout_polygon.py over polygon_ops.cpp); rd_layout_postprocess_select hands it the rows that reach that point.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <numeric>
#include <vector>

#include "../../include/rapiddoc_mi355.h"

namespace {

struct Box { float v[8]; int32_t src; };

static inline float iou_plus1(const float* a, const float* b) {  // post_process.py:921-946 (+1 pixel convention)
    const float x1 = std::max(a[0], b[0]), y1 = std::max(a[1], b[1]);
    const float x2 = std::min(a[2], b[2]), y2 = std::min(a[3], b[3]);
    const float iw = x2 - x1 + 1.f, ih = y2 - y1 + 1.f;
    const float inter = (iw > 0.f ? iw : 0.f) * (ih > 0.f ? ih : 0.f);
    const float a1 = (a[2] - a[0] + 1.f) * (a[3] - a[1] + 1.f);
    const float a2 = (b[2] - b[0] + 1.f) * (b[3] - b[1] + 1.f);
    return inter / (a1 + a2 - inter);
}

static inline bool is_contained(const Box& b1, const Box& b2) {  // post_process.py:981-994
    const float x1 = b1.v[2], y1 = b1.v[3], x2 = b1.v[4], y2 = b1.v[5];
    const float area = (x2 - x1) * (y2 - y1);
    const float xi1 = std::max(x1, b2.v[2]), yi1 = std::max(y1, b2.v[3]);
    const float xi2 = std::min(x2, b2.v[4]), yi2 = std::min(y2, b2.v[5]);
    const float iw = std::max(0.f, xi2 - xi1), ih = std::max(0.f, yi2 - yi1);
    const float r = area > 0.f ? (iw * ih) / area : 0.f;
    return r >= 0.9f;
}

static void check_containment(const std::vector<Box>& b, int formula_index, int category_index, int mode,
                              std::vector<int>& contains, std::vector<int>& contained) {
    const int n = (int)b.size();
    contains.assign(n, 0);
    contained.assign(n, 0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            if (i == j) continue;
            if (formula_index >= 0 && b[i].v[0] == (float)formula_index && b[j].v[0] != (float)formula_index) continue;
            bool test;
            if (category_index >= 0 && mode != 0)
                test = (mode == 1 && b[j].v[0] == (float)category_index) || (mode == 2 && b[i].v[0] == (float)category_index);
            else
                test = true;
            if (test && is_contained(b[i], b[j])) { contained[i] = 1; contains[j] = 1; }
        }
}


// steps 1 - 5: the rows that survive threshold / NMS / big-image filter / containment merge, in reading order
static std::vector<Box> select_rows(const float* boxes_in, int n, int ncol, int img_w, int img_h, const rd_layout_post_cfg* cfg) {
    std::vector<Box> bx;
    auto get = [&](int i) { Box b{}; for (int c = 0; c < ncol; ++c) b.v[c] = boxes_in[(size_t)i * ncol + c]; b.src = i; return b; };
    // 1. score threshold (float: original order; dict: grouped by ascending class id, np.unique + vstack)
    if (!cfg->thresh_is_dict) {
        for (int i = 0; i < n; ++i) {
            Box b = get(i);
            if (b.v[1] > cfg->thresh[0] && b.v[0] > -1.f) bx.push_back(b);
        }
    } else {
        std::vector<float> cats;
        for (int i = 0; i < n; ++i) cats.push_back(boxes_in[(size_t)i * ncol]);
        std::sort(cats.begin(), cats.end());
        cats.erase(std::unique(cats.begin(), cats.end()), cats.end());
        for (float cat : cats) {
            const int ci = (int)cat;
            const float th = (ci >= 0 && ci < cfg->n_classes && (float)ci == cat) ? cfg->thresh[ci] : 0.5f;
            for (int i = 0; i < n; ++i) {
                Box b = get(i);
                if (b.v[0] == cat && b.v[1] > th && b.v[0] > -1.f) bx.push_back(b);
            }
        }
    }
    // 2. greedy NMS: IoU >= 0.6 suppresses within a class, >= 0.98 across classes (post_process.py:76,948-979).
    //    Boxes with EQUAL scores: the reference orders them with np.argsort's default kind reversed (:954), which is not a stable sort
    //    (introsort, or x86-simd-sort where numpy dispatches to it), so their relative order there depends on the numpy build and the
    //    CPU; here they keep their input order.  Seen with a stand-in detector emitting exact ties (tests/golden/make_golden_layout_trace.py
    //    keeps its scores distinct for that reason); float32 scores of a real detector tie only by accident.
    if (cfg->layout_nms && !bx.empty()) {
        std::vector<int> idx(bx.size());
        std::iota(idx.begin(), idx.end(), 0);
        std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return bx[a].v[1] > bx[b].v[1]; });
        std::vector<int> keep;
        std::vector<char> dead(bx.size(), 0);
        for (size_t a = 0; a < idx.size(); ++a) {
            const int cur = idx[a];
            if (dead[cur]) continue;
            keep.push_back(cur);
            for (size_t c = a + 1; c < idx.size(); ++c) {
                const int o = idx[c];
                if (dead[o]) continue;
                const float th = bx[cur].v[0] == bx[o].v[0] ? 0.6f : 0.98f;
                if (!(iou_plus1(bx[cur].v + 2, bx[o].v + 2) < th)) dead[o] = 1;
            }
        }
        std::vector<Box> kept;
        for (int k : keep) kept.push_back(bx[k]);
        bx.swap(kept);
    }
    // 3. drop page-sized "image" boxes (post_process.py:81-121)
    if (bx.size() > 1) {
        const float area_thres = img_w > img_h ? 0.82f : 0.93f;
        const float lim = (float)((double)area_thres * ((double)img_w * (double)img_h));
        std::vector<Box> f;
        for (const Box& b : bx) {
            if (cfg->image_index >= 0 && b.v[0] == (float)cfg->image_index) {
                const float xmin = std::max(0.f, b.v[2]), ymin = std::max(0.f, b.v[3]);
                const float xmax = std::min((float)img_w, b.v[4]), ymax = std::min((float)img_h, b.v[5]);
                if ((xmax - xmin) * (ymax - ymin) <= lim) f.push_back(b);
            } else {
                f.push_back(b);
            }
        }
        if (!f.empty()) bx.swap(f);
    }
    // 4. containment merge (post_process.py:123-190)
    if (cfg->merge_kind == 1 || cfg->merge_kind == 2) {
        std::vector<int> contains, contained;
        check_containment(bx, cfg->formula_index, -1, 0, contains, contained);
        std::vector<Box> f;
        for (size_t i = 0; i < bx.size(); ++i)
            if (cfg->merge_kind == 1 ? contained[i] == 0 : (contains[i] == 0 || contained[i] == 1)) f.push_back(bx[i]);
        bx.swap(f);
    } else if (cfg->merge_kind == 3) {
        std::vector<char> keep(bx.size(), 1);
        std::vector<int> contains, contained;
        for (int c = 0; c < cfg->n_classes; ++c) {
            const int mode = cfg->merge_per_class[c];
            if (mode != 1 && mode != 2) continue;
            check_containment(bx, cfg->formula_index, c, mode, contains, contained);
            for (size_t i = 0; i < bx.size(); ++i)
                keep[i] &= mode == 1 ? contained[i] == 0 : (contains[i] == 0 || contained[i] == 1);
        }
        std::vector<Box> f;
        for (size_t i = 0; i < bx.size(); ++i)
            if (keep[i]) f.push_back(bx[i]);
        bx.swap(f);
    }
    if (bx.empty()) return bx;
    // 5. reading-order sort for the 7 / 8 column outputs (post_process.py:195-211)
    if (ncol == 8)
        std::stable_sort(bx.begin(), bx.end(), [](const Box& a, const Box& b) {
            return a.v[6] < b.v[6] || (a.v[6] == b.v[6] && -a.v[7] < -b.v[7]);
        });
    else if (ncol == 7)
        std::stable_sort(bx.begin(), bx.end(), [](const Box& a, const Box& b) { return a.v[6] < b.v[6]; });
    return bx;
}

}  // namespace

// The rows that reach the polygon stage (after the reading-order sort, before unclip / clip): sel_boxes [n][6] and, per row, its
// row index in `boxes` (so that the caller can carry the detector's masks along, post_process.py:46-211).  *n_sel rows are written;
// position i here is `out_order[k] - 1` of the row rd_layout_postprocess writes for it.
extern "C" int rd_layout_postprocess_select(const float* boxes_in, int n, int ncol, int img_w, int img_h, const rd_layout_post_cfg* cfg,
                                            float* sel_boxes, int32_t* sel_src, int32_t* n_sel) {
    if (!cfg || !sel_boxes || !sel_src || !n_sel || n < 0 || (ncol != 6 && ncol != 7 && ncol != 8) || (n > 0 && !boxes_in)) return 1;
    const std::vector<Box> bx = select_rows(boxes_in, n, ncol, img_w, img_h, cfg);
    for (size_t i = 0; i < bx.size(); ++i) {
        for (int c = 0; c < 6; ++c) sel_boxes[i * 6 + c] = bx[i].v[c];
        sel_src[i] = bx[i].src;
    }
    *n_sel = (int32_t)bx.size();
    return 0;
}

extern "C" int rd_layout_postprocess(const float* boxes_in, int n, int ncol, int img_w, int img_h, const rd_layout_post_cfg* cfg,
                                     float* out, int32_t* out_order, int32_t* n_out) {
    if (!cfg || !out || !out_order || !n_out || n < 0 || (ncol != 6 && ncol != 7 && ncol != 8) || (n > 0 && !boxes_in)) return 1;
    *n_out = 0;
    const std::vector<Box> bx = select_rows(boxes_in, n, ncol, img_w, img_h, cfg);
    // 6. unclip (post_process.py:611-662) + 7. clip / drop degenerate (post_process.py:566-608)
    int k = 0;
    for (size_t i = 0; i < bx.size(); ++i) {
        Box b = bx[i];
        bool do_unclip = cfg->unclip_kind == 1;
        float rw = 1.f, rh = 1.f;
        if (cfg->unclip_kind == 1) { rw = cfg->unclip[0]; rh = cfg->unclip[1]; }
        if (cfg->unclip_kind == 2) {
            const int ci = (int)b.v[0];
            if (ci >= 0 && ci < cfg->n_classes && (float)ci == b.v[0] && cfg->unclip_present[ci]) {
                do_unclip = true; rw = cfg->unclip[2 * ci]; rh = cfg->unclip[2 * ci + 1];
            }
        }
        if (do_unclip) {
            const float w = b.v[4] - b.v[2], h = b.v[5] - b.v[3];
            const float nw = w * rw, nh = h * rh;
            const float cx = b.v[2] + w / 2.f, cy = b.v[3] + h / 2.f;
            b.v[2] = cx - nw / 2.f; b.v[3] = cy - nh / 2.f; b.v[4] = cx + nw / 2.f; b.v[5] = cy + nh / 2.f;
        }
        const float xmin = std::max(0.f, b.v[2]), ymin = std::max(0.f, b.v[3]);
        const float xmax = std::min((float)img_w, b.v[4]), ymax = std::min((float)img_h, b.v[5]);
        if (xmax <= xmin || ymax <= ymin) continue;
        float* o = out + (size_t)k * 6;
        o[0] = b.v[0]; o[1] = b.v[1]; o[2] = xmin; o[3] = ymin; o[4] = xmax; o[5] = ymax;
        out_order[k] = (int32_t)i + 1;
        ++k;
    }
    *n_out = k;
    return 0;
}
