// DB post-process: probability map -> text boxes.  Host-side C++ (the reference runs this step on the host too,
// inside the third-party rapidocr package on top of OpenCV + pyclipper + shapely, none of which is vendored):
//   rapidocr DBPostProcess.__call__ as patched in rapid_doc/model/ocr/ocr_patch.py:223-241 (box_type "quad"),
//   constructed at ocr_patch.py:141-154 (thresh .3, box_thresh from model_init.py:73 / :22, unclip 1.8 / 1.6,
//   max_candidates 1000, 2x2 dilation, score_mode "fast").
// The arithmetic restated here is the public PaddleOCR algorithm those calls implement:
//   bitmap = pred > thresh -> dilate 2x2 -> connected regions -> min-area rectangle of each region (min side >= 3)
//   -> box_score_fast (mean probability inside the rectangle) >= box_thresh -> unclip by area*ratio/perimeter
//   -> min-area rectangle again (min side >= 5) -> scale to the source image, round, clip
//   -> filter_det_res (clockwise order, clip, drop boxes with a side <= 3 px).
// PARITY UNPINNED: the reference holds no vectors for this step and cv2/pyclipper are not available to mint any;
// tests pin it against an independent restatement (oracle/dbpost.py) and analytic known answers.
// Differences to OpenCV that are known and accepted (all sub-pixel): regions are 8-connected components whose OUTER
// boundary is used (RETR_LIST would also return hole borders as extra candidates); the polygon fill of
// box_score_fast uses an inclusive point-in-convex-quad test instead of cv2.fillPoly's line rasteriser; the
// JT_ROUND offset of a rectangle is replaced by its exact min-area rectangle (the rectangle grown by `distance`).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

#include "../../include/rapiddoc_mi355.h"

#include "db_geom.h"

namespace {

using rd_db::Cand;
using rd_db::P2;
using rd_db::Rect;
using rd_db::cross;

// Andrew monotone chain over (x, y)-sorted points; returns the hull in the canonical order of db_geom.h (counter-clockwise
// in (x, y) numbers, i.e. visually clockwise with y down, starting at the smallest (x, y)), no duplicates
static std::vector<P2> convex_hull(std::vector<P2> pts) {
    std::sort(pts.begin(), pts.end(), [](const P2& a, const P2& b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    pts.erase(std::unique(pts.begin(), pts.end(), [](const P2& a, const P2& b) { return a.x == b.x && a.y == b.y; }), pts.end());
    const int n = (int)pts.size();
    if (n < 3) return pts;
    std::vector<P2> h(2 * n);
    int k = 0;
    for (int i = 0; i < n; ++i) {
        while (k >= 2 && cross(h[k - 2], h[k - 1], pts[i]) <= 0) --k;
        h[k++] = pts[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
        while (k >= t && cross(h[k - 2], h[k - 1], pts[i]) <= 0) --k;
        h[k++] = pts[i];
    }
    h.resize(k - 1);
    return h;
}

static double box_score_fast(const float* pred, int H, int W, const P2 box[4]) {
    double xmn = 1e300, xmx = -1e300, ymn = 1e300, ymx = -1e300;
    for (int i = 0; i < 4; ++i) {
        xmn = std::min(xmn, box[i].x); xmx = std::max(xmx, box[i].x);
        ymn = std::min(ymn, box[i].y); ymx = std::max(ymx, box[i].y);
    }
    const int x0 = std::min(std::max((int)std::floor(xmn), 0), W - 1), x1 = std::min(std::max((int)std::ceil(xmx), 0), W - 1);
    const int y0 = std::min(std::max((int)std::floor(ymn), 0), H - 1), y1 = std::min(std::max((int)std::ceil(ymx), 0), H - 1);
    // polygon vertices relative to (x0, y0), truncated to int32 like `.astype("int32")`
    long qx[4], qy[4];
    for (int i = 0; i < 4; ++i) { qx[i] = (long)(box[i].x - x0); qy[i] = (long)(box[i].y - y0); }
    // orientation
    long area2 = 0;
    for (int i = 0; i < 4; ++i) area2 += qx[i] * qy[(i + 1) & 3] - qx[(i + 1) & 3] * qy[i];
    const int sgn = area2 >= 0 ? 1 : -1;
    double sum = 0;
    long cnt = 0;
    for (int y = y0; y <= y1; ++y)
        for (int x = x0; x <= x1; ++x) {
            const long px = x - x0, py = y - y0;
            bool in = true;
            for (int i = 0; i < 4 && in; ++i) {
                const long cr = (qx[(i + 1) & 3] - qx[i]) * (py - qy[i]) - (qy[(i + 1) & 3] - qy[i]) * (px - qx[i]);
                in = (cr * sgn) >= 0;
            }
            if (in) { sum += pred[(size_t)y * W + x]; ++cnt; }
        }
    return cnt ? sum / cnt : 0.0;
}

struct Params { float thresh, box_thresh, unclip_ratio; int use_dilation, max_candidates, min_size; };

// border pixels (or any point set with the same convex hull) of one region -> min-area rectangle candidate
static bool make_candidate(const std::vector<P2>& border, const Params& pr, Cand& c) {
    const std::vector<P2> h = convex_hull(border);
    return rd_db::make_candidate_hull(h.data(), (int)h.size(), pr.min_size, c);
}

static int finish_candidate(const Cand& c, double score, int H, int W, int src_h, int src_w, const Params& pr, rd_text_box* out) {
    static_assert(sizeof(rd_db::TextBox) == sizeof(rd_text_box), "rd_text_box layout");
    return rd_db::finish_candidate(c, score, H, W, src_h, src_w, pr.box_thresh, pr.unclip_ratio, pr.min_size,
                                   reinterpret_cast<rd_db::TextBox*>(out));
}

static int process_one(const float* pred, int H, int W, int src_h, int src_w, const Params& pr, rd_text_box* out, int max_out) {
    const size_t n = (size_t)H * W;
    std::vector<uint8_t> bin(n), bm(n);
    for (size_t i = 0; i < n; ++i) bin[i] = pred[i] > pr.thresh;
    if (pr.use_dilation) {  // cv2.dilate, 2x2 ones, anchor (1,1): max over (y-1..y, x-1..x)
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                uint8_t v = bin[(size_t)y * W + x];
                if (x > 0) v |= bin[(size_t)y * W + x - 1];
                if (y > 0) {
                    v |= bin[(size_t)(y - 1) * W + x];
                    if (x > 0) v |= bin[(size_t)(y - 1) * W + x - 1];
                }
                bm[(size_t)y * W + x] = v;
            }
    } else {
        bm = bin;
    }
    std::vector<int32_t> label(n, 0);
    std::vector<int32_t> stack;
    int n_out = 0, n_cand = 0;
    std::vector<P2> border;
    for (int y = 0; y < H && n_out < max_out; ++y)
        for (int x = 0; x < W && n_out < max_out; ++x) {
            const size_t s = (size_t)y * W + x;
            if (!bm[s] || label[s]) continue;
            // flood fill the 8-connected region, collecting its border pixels
            border.clear();
            stack.clear();
            stack.push_back((int32_t)s);
            label[s] = 1;
            while (!stack.empty()) {
                const int32_t q = stack.back();
                stack.pop_back();
                const int qy = q / W, qx = q - qy * W;
                bool edge = qx == 0 || qy == 0 || qx == W - 1 || qy == H - 1;
                for (int dy = -1; dy <= 1; ++dy)
                    for (int dx = -1; dx <= 1; ++dx) {
                        if (!dx && !dy) continue;
                        const int ny = qy + dy, nx = qx + dx;
                        if (ny < 0 || nx < 0 || ny >= H || nx >= W) continue;
                        const size_t t = (size_t)ny * W + nx;
                        if (!bm[t]) {
                            if (!dx || !dy) edge = true;
                            continue;
                        }
                        if (!label[t]) { label[t] = 1; stack.push_back((int32_t)t); }
                    }
                if (edge) border.push_back({(double)qx, (double)qy});
            }
            if (++n_cand > pr.max_candidates) return n_out;
            Cand c;
            if (!make_candidate(border, pr, c)) continue;
            const double score = box_score_fast(pred, H, W, c.box);
            n_out += finish_candidate(c, score, H, W, src_h, src_w, pr, out + n_out);
        }
    return n_out;
}

}  // namespace

// ---- device-assisted path: the GPU thresholds / dilates the maps and emits the horizontal RUNS of the bitmap
// (rd_db_runs), scores the candidate rectangles (rd_db_scores); the host only sees a few thousand runs per page.
struct Run { int16_t y, x0, x1, pad; };

// runs of one page -> 8-connected regions in raster order of their first pixel -> candidates (same rectangles as the
// flood fill above: a region's convex hull is the hull of its run end points)
static int candidates_from_runs(std::vector<Run>& runs, int max_cand, const Params& pr, rd_db_candidate* out) {
    std::sort(runs.begin(), runs.end(), [](const Run& a, const Run& b) { return a.y < b.y || (a.y == b.y && a.x0 < b.x0); });
    const int n = (int)runs.size();
    std::vector<int> parent(n);
    for (int i = 0; i < n; ++i) parent[i] = i;
    auto find = [&](int i) { while (parent[i] != i) { parent[i] = parent[parent[i]]; i = parent[i]; } return i; };
    int prev_b = 0, prev_e = 0;   // [prev_b, prev_e): runs of the previous row
    for (int i = 0; i < n;) {
        int j = i;
        while (j < n && runs[j].y == runs[i].y) ++j;
        if (prev_e > prev_b && runs[prev_b].y == runs[i].y - 1) {
            int q = prev_b;
            for (int k = i; k < j; ++k) {
                while (q < prev_e && runs[q].x1 + 1 < runs[k].x0) ++q;           // runs entirely to the left
                for (int t = q; t < prev_e && runs[t].x0 <= runs[k].x1 + 1; ++t) {  // 8-connected: columns may differ by one
                    const int a = find(t), b = find(k);
                    if (a != b) parent[std::max(a, b)] = std::min(a, b);           // root = earliest run = first pixel in raster order
                }
            }
        }
        prev_b = i;
        prev_e = j;
        i = j;
    }
    // regions in order of their root run (runs are sorted in raster order, the root is the smallest index)
    std::vector<std::vector<P2>> pts;
    std::vector<int> slot(n, -1);
    for (int i = 0; i < n; ++i) {
        const int r = find(i);
        if (slot[r] < 0) { slot[r] = (int)pts.size(); pts.emplace_back(); }
        pts[slot[r]].push_back({(double)runs[i].x0, (double)runs[i].y});
        if (runs[i].x1 != runs[i].x0) pts[slot[r]].push_back({(double)runs[i].x1, (double)runs[i].y});
    }
    int n_out = 0, n_cand = 0;
    for (auto& border : pts) {
        if (++n_cand > pr.max_candidates || n_out >= max_cand) break;
        Cand c;
        if (!make_candidate(border, pr, c)) continue;
        for (int k = 0; k < 4; ++k) {
            out[n_out].box[2 * k] = c.box[k].x; out[n_out].box[2 * k + 1] = c.box[k].y;
            out[n_out].rect[2 * k] = c.r.c[k].x; out[n_out].rect[2 * k + 1] = c.r.c[k].y;
        }
        out[n_out].w = c.r.w;
        out[n_out].h = c.r.h;
        ++n_out;
    }
    return n_out;
}

extern "C" int rd_db_candidates(const void* runs_host, const int32_t* n_runs, int B, int max_runs, int max_candidates,
                                rd_db_candidate* out, int max_out, int32_t* n_out) {
    if (!runs_host || !n_runs || !out || !n_out || B < 0 || max_runs <= 0 || max_out <= 0) return 1;
    Params pr{0.f, 0.f, 0.f, 0, max_candidates > 0 ? max_candidates : 1000, 3};
    for (int b = 0; b < B; ++b) {
        if (n_runs[b] < 0 || n_runs[b] > max_runs) return 2;      // the device buffer overflowed: caller falls back to the host path
        const Run* r = reinterpret_cast<const Run*>(runs_host) + (size_t)b * max_runs;
        std::vector<Run> runs(r, r + n_runs[b]);
        n_out[b] = candidates_from_runs(runs, max_out, pr, out + (size_t)b * max_out);
    }
    return 0;
}

extern "C" int rd_db_finish(const rd_db_candidate* cand, const double* scores, const int32_t* n_cand, int B, int max_cand, int H, int W,
                            const int32_t* src_hw, float box_thresh, float unclip_ratio, rd_text_box* out, int max_out, int32_t* n_out) {
    if (!cand || !scores || !n_cand || !src_hw || !out || !n_out || B < 0) return 1;
    Params pr{0.f, box_thresh, unclip_ratio, 0, 0, 3};
    for (int b = 0; b < B; ++b) {
        int n = 0;
        for (int i = 0; i < n_cand[b] && n < max_out; ++i) {
            const rd_db_candidate& q = cand[(size_t)b * max_cand + i];
            Cand c;
            for (int k = 0; k < 4; ++k) {
                c.box[k] = {q.box[2 * k], q.box[2 * k + 1]};
                c.r.c[k] = {q.rect[2 * k], q.rect[2 * k + 1]};
            }
            c.r.w = q.w;
            c.r.h = q.h;
            n += finish_candidate(c, scores[(size_t)b * max_cand + i], H, W, src_hw[2 * b], src_hw[2 * b + 1], pr, out + (size_t)b * max_out + n);
        }
        n_out[b] = n;
    }
    return 0;
}

extern "C" int rd_db_postprocess(const float* prob_host, int B, int H, int W, const int32_t* src_hw, float thresh, float box_thresh,
                                 float unclip_ratio, int use_dilation, int max_candidates, rd_text_box* out, int max_out,
                                 int32_t* n_out, int n_threads) {
    if (!prob_host || !src_hw || !out || !n_out || B < 0 || H <= 0 || W <= 0 || max_out <= 0) return 1;
    Params pr{thresh, box_thresh, unclip_ratio, use_dilation, max_candidates > 0 ? max_candidates : 1000, 3};
    auto work = [&](int b) {
        n_out[b] = process_one(prob_host + (size_t)b * H * W, H, W, src_hw[2 * b], src_hw[2 * b + 1], pr, out + (size_t)b * max_out, max_out);
    };
    const int nt = std::max(1, std::min(n_threads > 0 ? n_threads : (int)std::thread::hardware_concurrency(), B));
    if (nt <= 1) {
        for (int b = 0; b < B; ++b) work(b);
    } else {
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; ++t)
            pool.emplace_back([&, t] { for (int b = t; b < B; b += nt) work(b); });
        for (auto& th : pool) th.join();
    }
    return 0;
}
