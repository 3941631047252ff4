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
// Contours (cv2.findContours(RETR_LIST), Suzuki-Abe border following, 8-connected foreground / 4-connected background):
//   * one OUTER border per 8-connected region of the bitmap;
//   * one HOLE border per 4-connected background component that does not reach the image frame: the region pixels that have a
//     pixel of that hole among their 4 neighbours.  RETR_LIST returns hole borders as contours like any other, so the reference
//     scores them as candidates too (a 3 x 3 hole inside a text blob yields an extra ~9-px box there); round 2 dropped them.
//   Candidates are taken in raster order of the contour's start pixel (outer border: the region's first pixel; hole border:
//   the pixel left of the hole's first pixel), the first `max_candidates` count.  (OpenCV hands the list back in reverse
//   discovery order; that only matters to which 1000 survive on a map with more than 1000 contours.)
// Differences to OpenCV that are known and accepted (all sub-pixel): the polygon fill of box_score_fast uses an inclusive
// point-in-convex-quad test instead of cv2.fillPoly's line rasteriser; the JT_ROUND offset of a rectangle is replaced by its
// exact min-area rectangle (the rectangle grown by `distance`).
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
    struct Contour { int key_y, key_x; std::vector<P2> pts; };
    std::vector<Contour> contours;
    // outer borders: flood fill every 8-connected region, collecting its border pixels
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t s = (size_t)y * W + x;
            if (!bm[s] || label[s]) continue;
            Contour c{y, x, {}};
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
                if (edge) c.pts.push_back({(double)qx, (double)qy});
            }
            contours.push_back(std::move(c));
        }
    // hole borders: 4-connected background components that do not reach the image frame
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) {
            const size_t s = (size_t)y * W + x;
            if (bm[s] || label[s]) continue;
            Contour c{y, x - 1, {}};
            bool open = false;
            stack.clear();
            stack.push_back((int32_t)s);
            label[s] = 1;
            while (!stack.empty()) {
                const int32_t q = stack.back();
                stack.pop_back();
                const int qy = q / W, qx = q - qy * W;
                if (qx == 0 || qy == 0 || qx == W - 1 || qy == H - 1) open = true;
                const int ny4[4] = {qy - 1, qy + 1, qy, qy}, nx4[4] = {qx, qx, qx - 1, qx + 1};
                for (int k = 0; k < 4; ++k) {
                    if (ny4[k] < 0 || nx4[k] < 0 || ny4[k] >= H || nx4[k] >= W) continue;
                    const size_t t = (size_t)ny4[k] * W + nx4[k];
                    if (bm[t]) c.pts.push_back({(double)nx4[k], (double)ny4[k]});        // a region pixel on this hole's border
                    else if (!label[t]) { label[t] = 1; stack.push_back((int32_t)t); }
                }
            }
            if (!open) contours.push_back(std::move(c));
        }
    std::stable_sort(contours.begin(), contours.end(), [](const Contour& a, const Contour& b) {
        return a.key_y < b.key_y || (a.key_y == b.key_y && a.key_x < b.key_x);
    });
    int n_out = 0, n_cand = 0;
    for (const Contour& ct : contours) {
        if (n_out >= max_out || ++n_cand > pr.max_candidates) break;
        Cand c;
        if (!make_candidate(ct.pts, pr, c)) continue;
        const double score = box_score_fast(pred, H, W, c.box);
        n_out += finish_candidate(c, score, H, W, src_h, src_w, pr, out + n_out);
    }
    return n_out;
}

}  // namespace

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
