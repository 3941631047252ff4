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

namespace {

struct P2 { double x, y; };

static double cross(const P2& o, const P2& a, const P2& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }

// Andrew monotone chain; returns hull in counter-clockwise order (y down => visually clockwise), no duplicates
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

struct Rect { P2 c[4]; double w, h; };

// minimum-area enclosing rectangle of a point set (rotating calipers over hull edges) - cv2.minAreaRect + boxPoints
static bool min_area_rect(const std::vector<P2>& pts, Rect& out) {
    std::vector<P2> h = convex_hull(pts);
    const int n = (int)h.size();
    if (n == 0) return false;
    if (n == 1) {
        for (auto& c : out.c) c = h[0];
        out.w = out.h = 0;
        return true;
    }
    double best = 1e300;
    for (int i = 0; i < n; ++i) {
        const P2 a = h[i], b = h[(i + 1) % n];
        double ex = b.x - a.x, ey = b.y - a.y;
        const double len = std::sqrt(ex * ex + ey * ey);
        if (len == 0) continue;
        ex /= len; ey /= len;
        double mn_u = 1e300, mx_u = -1e300, mn_v = 1e300, mx_v = -1e300;
        for (const P2& p : h) {
            const double u = (p.x - a.x) * ex + (p.y - a.y) * ey;
            const double v = -(p.x - a.x) * ey + (p.y - a.y) * ex;
            mn_u = std::min(mn_u, u); mx_u = std::max(mx_u, u);
            mn_v = std::min(mn_v, v); mx_v = std::max(mx_v, v);
        }
        const double area = (mx_u - mn_u) * (mx_v - mn_v);
        if (area < best) {
            best = area;
            const double us[4] = {mn_u, mx_u, mx_u, mn_u}, vs[4] = {mn_v, mn_v, mx_v, mx_v};
            for (int k = 0; k < 4; ++k) out.c[k] = {a.x + us[k] * ex - vs[k] * ey, a.y + us[k] * ey + vs[k] * ex};
            out.w = mx_u - mn_u;
            out.h = mx_v - mn_v;
        }
        if (n == 2) break;
    }
    return true;
}

// PaddleOCR get_mini_boxes ordering: sort by x; left pair by y -> (tl, bl); right pair by y -> (tr, br)
static void order_mini_box(const P2 in[4], P2 out[4]) {
    P2 p[4] = {in[0], in[1], in[2], in[3]};
    std::stable_sort(p, p + 4, [](const P2& a, const P2& b) { return a.x < b.x; });
    int i1, i2, i3, i4;
    if (p[1].y > p[0].y) { i1 = 0; i4 = 1; } else { i1 = 1; i4 = 0; }
    if (p[3].y > p[2].y) { i2 = 2; i3 = 3; } else { i2 = 3; i3 = 2; }
    out[0] = p[i1]; out[1] = p[i2]; out[2] = p[i3]; out[3] = p[i4];
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

struct Cand { Rect r; P2 box[4]; };

// border pixels (or any point set with the same convex hull) of one region -> min-area rectangle candidate
static bool make_candidate(const std::vector<P2>& border, const Params& pr, Cand& c) {
    if (!min_area_rect(border, c.r)) return false;
    if (std::min(c.r.w, c.r.h) < pr.min_size) return false;
    order_mini_box(c.r.c, c.box);
    return true;
}

// score filter, unclip, second min-area rectangle, scale to the source image, filter_det_res.  Returns 1 if a box was written.
static int finish_candidate(const Cand& c, double score, int H, int W, int src_h, int src_w, const Params& pr, rd_text_box* out) {
    if (pr.box_thresh > score) return 0;
    // unclip: Polygon(box).area * ratio / Polygon(box).length; pyclipper works on integer coordinates
    const Rect& r = c.r;
    const P2* box = c.box;
    const double area = r.w * r.h, perim = 2.0 * (r.w + r.h);
    if (perim <= 0) return 0;
    const double dist = area * pr.unclip_ratio / perim;
    std::vector<P2> ip(4);
    for (int i = 0; i < 4; ++i) ip[i] = {(double)(long)box[i].x, (double)(long)box[i].y};
    Rect ri;
    if (!min_area_rect(ip, ri) || ri.w <= 0 || ri.h <= 0) return 0;
    // grow the rectangle by `dist` on every side (== min-area rect of the round-join offset polygon)
    P2 cen = {0, 0};
    for (auto& c : ri.c) { cen.x += c.x * 0.25; cen.y += c.y * 0.25; }
    double ux = ri.c[1].x - ri.c[0].x, uy = ri.c[1].y - ri.c[0].y;
    double vx = ri.c[3].x - ri.c[0].x, vy = ri.c[3].y - ri.c[0].y;
    const double ul = std::sqrt(ux * ux + uy * uy), vl = std::sqrt(vx * vx + vy * vy);
    ux /= ul; uy /= ul; vx /= vl; vy /= vl;
    const double hu = ul * 0.5 + dist, hv = vl * 0.5 + dist;
    P2 ex[4] = {{cen.x - hu * ux - hv * vx, cen.y - hu * uy - hv * vy}, {cen.x + hu * ux - hv * vx, cen.y + hu * uy - hv * vy},
                {cen.x + hu * ux + hv * vx, cen.y + hu * uy + hv * vy}, {cen.x - hu * ux + hv * vx, cen.y - hu * uy + hv * vy}};
    if (std::min(2 * hu, 2 * hv) < pr.min_size + 2) return 0;
    P2 eb[4];
    order_mini_box(ex, eb);
    // scale to the source image: np.clip(np.round(x / width * dest_width), 0, dest_width) -> int32
    long bx[4], by[4];
    for (int i = 0; i < 4; ++i) {
        bx[i] = (long)std::min(std::max(std::nearbyint(eb[i].x / W * src_w), 0.0), (double)src_w);
        by[i] = (long)std::min(std::max(std::nearbyint(eb[i].y / H * src_h), 0.0), (double)src_h);
    }
    // filter_det_res: order_points_clockwise, clip to the image, drop tiny boxes
    int idx[4] = {0, 1, 2, 3};
    std::stable_sort(idx, idx + 4, [&](int a, int b) { return bx[a] < bx[b]; });
    int l0 = idx[0], l1 = idx[1], r0 = idx[2], r1 = idx[3];
    if (by[l1] < by[l0]) std::swap(l0, l1);
    if (by[r1] < by[r0]) std::swap(r0, r1);
    const int ord[4] = {l0, r0, r1, l1};  // tl, tr, br, bl
    float pts[8];
    for (int i = 0; i < 4; ++i) {
        pts[2 * i] = (float)std::min(std::max(bx[ord[i]], 0L), (long)src_w - 1);
        pts[2 * i + 1] = (float)std::min(std::max(by[ord[i]], 0L), (long)src_h - 1);
    }
    const int rw = (int)std::sqrt((pts[0] - pts[2]) * (pts[0] - pts[2]) + (pts[1] - pts[3]) * (pts[1] - pts[3]));
    const int rh = (int)std::sqrt((pts[0] - pts[6]) * (pts[0] - pts[6]) + (pts[1] - pts[7]) * (pts[1] - pts[7]));
    if (rw <= 3 || rh <= 3) return 0;
    std::memcpy(out->pts, pts, sizeof(pts));
    out->score = (float)score;
    return 1;
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
