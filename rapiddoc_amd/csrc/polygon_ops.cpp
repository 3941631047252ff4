// Raster / polygon primitives of the layout post-process's POLYGON branch (masks -> polygon_points) and of the polygon-masked
// region crop.  Host-side C++: in the reference these are OpenCV / GEOS (shapely) calls made from Python,
//   rapid_doc/model/layout/rapid_layout_self/model_handler/pp_doclayout/post_process.py
//     :409-417  mask2polygon              cv2.findContours(RETR_EXTERNAL, CHAIN_APPROX_SIMPLE), contourArea, arcLength, approxPolyDP
//     :468-470  extract_polygon_points_by_masks   cv2.resize(INTER_NEAREST)     (done with numpy indexing on the Python side)
//     :553-554  convert_polygon_to_quad   cv2.minAreaRect + cv2.boxPoints
//     :692-711  calculate_polygon_overlap_ratio   shapely Polygon.intersection / union / area
//   rapid_doc/utils/model_utils.py:109-118  crop_img     cv2.fillPoly(mask, [polygon], 1)
// PARITY UNPINNED: cv2 and shapely are not installed here and the reference holds no vectors for these calls; every function is
// restated from the library's published algorithm (named at the function) and tested against analytic known answers
// (tests/test_polygon_ops.py).  What IS pinned is everything the reference does around them: tests/golden/make_golden_polygon.py
// runs the reference's own post_process.py with these functions plugged in as `cv2` / `shapely`.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../include/rapiddoc_mi355.h"

#include "db_geom.h"

namespace {

struct Pt { int x, y; };

// OpenCV's 8-neighbour codes: 0 = E, 1 = NE, 2 = N, 3 = NW, 4 = W, 5 = SW, 6 = S, 7 = SE (y points down)
static const int DX[8] = {1, 1, 0, -1, -1, -1, 0, 1};
static const int DY[8] = {0, -1, -1, -1, 0, 1, 1, 1};

// Suzuki-Abe border following of ONE outer border starting at (x0, y0) (the component's first pixel in raster order, so its
// west neighbour is background), as OpenCV's contour fetcher walks it: first non-zero neighbour clockwise from west, then
// counter-clockwise search from the direction after the one we came from; stop when the first step is about to repeat.
// CHAIN_APPROX_SIMPLE: a point is written only where the direction changes.
static void trace_outer_border(const uint8_t* img, int H, int W, int x0, int y0, std::vector<Pt>& out) {
    auto at = [&](int x, int y) -> int { return (x >= 0 && y >= 0 && x < W && y < H) ? img[(size_t)y * W + x] != 0 : 0; };
    int s = 4;
    const int s_stop = 4;
    int x1 = x0, y1 = y0;
    do {
        s = (s - 1) & 7;
        x1 = x0 + DX[s];
        y1 = y0 + DY[s];
    } while (!at(x1, y1) && s != s_stop);
    if (s == s_stop) {               // a single pixel
        out.push_back({x0, y0});
        return;
    }
    int x3 = x0, y3 = y0, prev_s = s ^ 4;
    int px = x0, py = y0;            // the point being written (== (x3, y3))
    for (;;) {
        int x4, y4;
        for (;;) {
            ++s;
            x4 = x3 + DX[s & 7];
            y4 = y3 + DY[s & 7];
            if (at(x4, y4)) break;
        }
        s &= 7;
        if (s != prev_s) {
            out.push_back({px, py});
            prev_s = s;
        }
        px += DX[s];
        py += DY[s];
        if (x4 == x0 && y4 == y0 && x3 == x1 && y3 == y1) break;
        x3 = x4;
        y3 = y4;
        s = (s + 4) & 7;
    }
}

// x of the 16.16 fixed-point polygon edges, as cv2's CollectPolyEdges / FillEdgeCollection keep them
struct Edge { int y0, y1; int64_t x, dx; };
static constexpr int XY_SHIFT = 16;
static constexpr int64_t XY_ONE = 1 << XY_SHIFT;

static inline void put(uint8_t* img, int H, int W, int x, int y, uint8_t v) {
    if (x >= 0 && y >= 0 && x < W && y < H) img[(size_t)y * W + x] = v;
}

// cv2's 8-connected LineIterator, drawn left to right
static void draw_line8(uint8_t* img, int H, int W, Pt a, Pt b, uint8_t v) {
    if (b.x < a.x) std::swap(a, b);
    int dx = b.x - a.x, dy = b.y - a.y;
    const int ystep = dy < 0 ? -1 : 1;
    dy = dy < 0 ? -dy : dy;
    const bool steep = dy > dx;
    const int major = steep ? dy : dx, minor = steep ? dx : dy;
    int err = major - 2 * minor;
    int x = a.x, y = a.y;
    put(img, H, W, x, y, v);
    for (int i = 0; i < major; ++i) {
        const bool both = err < 0;
        err += -2 * minor + (both ? 2 * major : 0);
        if (steep) { y += ystep; if (both) x += 1; }
        else       { x += 1;     if (both) y += ystep; }
        put(img, H, W, x, y, v);
    }
}

// crossings of the line x = xm with the non-vertical edges of a polygon, ascending
static void crossings(const double* p, int n, double xm, std::vector<double>& ys) {
    ys.clear();
    for (int i = 0; i < n; ++i) {
        const double ax = p[2 * i], ay = p[2 * i + 1], bx = p[2 * ((i + 1) % n)], by = p[2 * ((i + 1) % n) + 1];
        if (ax == bx) continue;
        const double lo = std::min(ax, bx), hi = std::max(ax, bx);
        if (xm <= lo || xm >= hi) continue;
        ys.push_back(ay + (by - ay) * ((xm - ax) / (bx - ax)));
    }
    std::sort(ys.begin(), ys.end());
}

// measure of (union of [a0,a1],[a2,a3],...) intersected with the same for b (even-odd interiors along one vertical line)
static double overlap_1d(const std::vector<double>& a, const std::vector<double>& b) {
    double s = 0;
    size_t i = 0, j = 0;
    while (i + 1 < a.size() && j + 1 < b.size()) {
        const double lo = std::max(a[i], b[j]), hi = std::min(a[i + 1], b[j + 1]);
        if (hi > lo) s += hi - lo;
        if (a[i + 1] < b[j + 1]) i += 2; else j += 2;
    }
    return s;
}

static void slab_xs(const double* a, int na, const double* b, int nb, std::vector<double>& xs) {
    xs.clear();
    for (int i = 0; i < na; ++i) xs.push_back(a[2 * i]);
    for (int i = 0; i < nb; ++i) xs.push_back(b[2 * i]);
    auto add_crossings = [&](const double* p, int n, const double* q, int m, bool same) {
        for (int i = 0; i < n; ++i) {
            const double x1 = p[2 * i], y1 = p[2 * i + 1], x2 = p[2 * ((i + 1) % n)], y2 = p[2 * ((i + 1) % n) + 1];
            for (int j = same ? i + 1 : 0; j < m; ++j) {
                const double x3 = q[2 * j], y3 = q[2 * j + 1], x4 = q[2 * ((j + 1) % m)], y4 = q[2 * ((j + 1) % m) + 1];
                const double d = (x2 - x1) * (y4 - y3) - (y2 - y1) * (x4 - x3);
                if (d == 0) continue;
                const double t = ((x3 - x1) * (y4 - y3) - (y3 - y1) * (x4 - x3)) / d;
                const double u = ((x3 - x1) * (y2 - y1) - (y3 - y1) * (x2 - x1)) / d;
                if (t >= 0 && t <= 1 && u >= 0 && u <= 1) xs.push_back(x1 + t * (x2 - x1));
            }
        }
    };
    add_crossings(a, na, b, nb, false);
    add_crossings(a, na, a, na, true);       // self-crossings: the even-odd interior changes there too
    if (b != a) add_crossings(b, nb, b, nb, true);
    std::sort(xs.begin(), xs.end());
    xs.erase(std::unique(xs.begin(), xs.end()), xs.end());
}

}  // namespace

// cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE): the outer border of every 8-connected component that does not lie
// inside a hole of another one (its first pixel's west neighbour belongs to background that is 4-connected to the image frame).
// Returned last-found first, the order OpenCV's list has (it only matters to which of several equal-area contours a caller's
// `max(...)` picks).  pts_out: [max_pts][2] (x, y); counts_out[c] = points of contour c.
// Returns 0, 1 (bad arguments) or 2 (capacity: *n_contours / *n_pts hold what is needed).
extern "C" int rd_find_external_contours(const uint8_t* mask, int h, int w, int32_t* pts_out, int max_pts, int32_t* counts_out,
                                         int max_contours, int32_t* n_contours, int32_t* n_pts) {
    if (!mask || h <= 0 || w <= 0 || !n_contours || !n_pts || max_pts < 0 || max_contours < 0) return 1;
    // One raster pass over row RUNS instead of per-pixel flood fills: foreground runs are joined with the runs of the previous row
    // they touch (8-connected: x ranges may differ by one), background runs with those they overlap (4-connected); node 0 of the
    // background forest is "outside the image".  A component is external iff the background run left of its first pixel is outside.
    struct Run { int x0, x1, id; };                       // [x0, x1)
    std::vector<int> fg_parent, bg_parent(1, 0);
    auto find = [](std::vector<int>& p, int a) { while (p[a] != a) { p[a] = p[p[a]]; a = p[a]; } return a; };
    auto unite = [&](std::vector<int>& p, int a, int b) { a = find(p, a); b = find(p, b); if (a != b) p[a > b ? a : b] = a > b ? b : a; };
    struct Start { int x, y, fg, bg_left; };              // first pixel of a foreground run whose component it may start
    std::vector<Start> starts;
    std::vector<Run> prev_fg, prev_bg, cur_fg, cur_bg;
    for (int y = 0; y < h; ++y) {
        const uint8_t* row = mask + (size_t)y * w;
        cur_fg.clear();
        cur_bg.clear();
        for (int x = 0; x < w;) {
            const bool fg = row[x] != 0;
            int e = x + 1;
            while (e < w && (row[e] != 0) == fg) ++e;
            if (fg) {
                const int id = (int)fg_parent.size();
                fg_parent.push_back(id);
                cur_fg.push_back({x, e, id});
            } else {
                const int id = (int)bg_parent.size();
                bg_parent.push_back(id);
                cur_bg.push_back({x, e, id});
                if (x == 0 || e == w || y == 0 || y == h - 1) unite(bg_parent, id, 0);
            }
            x = e;
        }
        size_t j = 0;
        for (const Run& r : cur_fg) {                        // 8-connected: touches [x0 - 1, x1 + 1) of the previous row
            while (j < prev_fg.size() && prev_fg[j].x1 < r.x0) ++j;
            for (size_t k = j; k < prev_fg.size() && prev_fg[k].x0 <= r.x1; ++k) unite(fg_parent, r.id, prev_fg[k].id);
        }
        j = 0;
        for (const Run& r : cur_bg) {                        // 4-connected: overlaps [x0, x1) of the previous row
            while (j < prev_bg.size() && prev_bg[j].x1 <= r.x0) ++j;
            for (size_t k = j; k < prev_bg.size() && prev_bg[k].x0 < r.x1; ++k) unite(bg_parent, r.id, prev_bg[k].id);
        }
        size_t b = 0;
        for (const Run& r : cur_fg) {                        // the background run that ends where this foreground run starts
            while (b < cur_bg.size() && cur_bg[b].x1 < r.x0) ++b;
            starts.push_back({r.x0, y, r.id, r.x0 == 0 ? 0 : cur_bg[b].id});
        }
        prev_fg.swap(cur_fg);
        prev_bg.swap(cur_bg);
    }
    // the first run (raster order) of every component is its start pixel; later runs of the same component are dropped
    std::vector<char> taken(fg_parent.size(), 0);
    std::vector<std::vector<Pt>> found;
    for (const Start& st : starts) {
        const int root = find(fg_parent, st.fg);
        if (taken[root]) continue;
        taken[root] = 1;
        if (find(bg_parent, st.bg_left) != 0) continue;      // inside a hole of another component
        found.emplace_back();
        trace_outer_border(mask, h, w, st.x, st.y, found.back());
    }
    std::reverse(found.begin(), found.end());
    size_t total = 0;
    for (const auto& c : found) total += c.size();
    *n_contours = (int32_t)found.size();
    *n_pts = (int32_t)total;
    if ((int)found.size() > max_contours || (int64_t)total > max_pts || (!found.empty() && (!pts_out || !counts_out))) return 2;
    size_t o = 0;
    for (size_t c = 0; c < found.size(); ++c) {
        counts_out[c] = (int32_t)found[c].size();
        for (const Pt& p : found[c]) { pts_out[2 * o] = p.x; pts_out[2 * o + 1] = p.y; ++o; }
    }
    return 0;
}

// cv2.contourArea(contour) of integer points: |shoelace| / 2, products and sum in double
extern "C" double rd_contour_area(const int32_t* pts, int n) {
    if (!pts || n <= 0) return 0.0;
    double a = 0;
    double px = (float)pts[2 * (n - 1)], py = (float)pts[2 * (n - 1) + 1];
    for (int i = 0; i < n; ++i) {
        const double x = (float)pts[2 * i], y = (float)pts[2 * i + 1];
        a += px * y - py * x;
        px = x;
        py = y;
    }
    return std::fabs(a * 0.5);
}

// cv2.arcLength(curve, closed): float32 segment lengths summed in double
extern "C" double rd_arc_length(const int32_t* pts, int n, int closed) {
    if (!pts || n <= 1) return 0.0;
    double per = 0;
    const int last = closed ? n - 1 : 0;
    float px = (float)pts[2 * last], py = (float)pts[2 * last + 1];
    for (int i = 0; i < n; ++i) {
        const float x = (float)pts[2 * i], y = (float)pts[2 * i + 1];
        const float dx = x - px, dy = y - py;
        per += std::sqrt(dx * dx + dy * dy);
        px = x;
        py = y;
    }
    return per;
}

// cv2.approxPolyDP(curve, epsilon, closed) for integer points: OpenCV's Douglas-Peucker - for a closed curve the two starting
// points are found by three rounds of "farthest point from the current start", the recursion runs on an explicit stack of index
// ranges, and a last pass drops points that lie within sqrt(eps^2 / 2) of the segment joining their neighbours.
// out: room for n points.  Returns 0 or 1 (bad arguments).
extern "C" int rd_approx_poly_dp(const int32_t* pts, int n, double epsilon, int closed, int32_t* out, int32_t* n_out) {
    if (!n_out || n < 0 || (n > 0 && (!pts || !out)) || epsilon < 0) return 1;
    *n_out = 0;
    if (n == 0) return 0;
    auto src = [&](int i) { return Pt{pts[2 * i], pts[2 * i + 1]}; };
    std::vector<Pt> dst;
    dst.reserve(n);
    struct Range { int start, end; };
    std::vector<Range> stack;
    const int count = n;
    const double eps = epsilon * epsilon;
    Range slice{0, 0}, right{0, 0};
    Pt start_pt{-1000000, -1000000}, end_pt{0, 0}, pt{0, 0};
    int pos = 0;
    bool le_eps = false;
    auto read = [&](Pt& p, int& at) { p = src(at); if (++at >= count) at = 0; };

    if (!closed) {
        right.start = count;
        end_pt = src(0);
        start_pt = src(count - 1);
        if (start_pt.x != end_pt.x || start_pt.y != end_pt.y) {
            slice.start = 0;
            slice.end = count - 1;
            stack.push_back(slice);
        } else {
            closed = 1;          // the curve's ends coincide: OpenCV treats it as closed from here on
        }
    }
    const int is_closed0 = closed;
    if (closed) {
        right.start = 0;
        for (int it = 0; it < 3; ++it) {
            double max_dist = 0;
            pos = (pos + right.start) % count;
            read(start_pt, pos);
            for (int j = 1; j < count; ++j) {
                read(pt, pos);
                const double dx = pt.x - start_pt.x, dy = pt.y - start_pt.y;
                const double dist = dx * dx + dy * dy;
                if (dist > max_dist) { max_dist = dist; right.start = j; }
            }
            le_eps = max_dist <= eps;
        }
        if (!le_eps) {
            right.end = slice.start = pos % count;
            slice.end = right.start = (right.start + slice.start) % count;
            stack.push_back(right);
            stack.push_back(slice);
        } else {
            dst.push_back(start_pt);
        }
    }
    while (!stack.empty()) {
        slice = stack.back();
        stack.pop_back();
        end_pt = src(slice.end);
        pos = slice.start;
        read(start_pt, pos);
        if (pos != slice.end) {
            double max_dist = 0;
            const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
            while (pos != slice.end) {
                read(pt, pos);
                const double dist = std::fabs((pt.y - start_pt.y) * dx - (pt.x - start_pt.x) * dy);
                if (dist > max_dist) { max_dist = dist; right.start = (pos + count - 1) % count; }
            }
            le_eps = max_dist * max_dist <= eps * (dx * dx + dy * dy);
        } else {
            le_eps = true;
            start_pt = src(slice.start);
        }
        if (le_eps) {
            dst.push_back(start_pt);
        } else {
            right.end = slice.end;
            slice.end = right.start;
            stack.push_back(right);
            stack.push_back(slice);
        }
    }
    if (!is_closed0) dst.push_back(src(count - 1));

    // final clean-up: remove points on [almost] straight lines
    int new_count = (int)dst.size();
    const int cnt = new_count;
    auto read_dst = [&](Pt& p, int& at) { p = dst[at]; if (++at >= cnt) at = 0; };
    pos = is_closed0 ? cnt - 1 : 0;
    read_dst(start_pt, pos);
    int wpos = pos;
    read_dst(pt, pos);
    const int open = !is_closed0;
    for (int i = open; i < cnt - open && new_count > 2; ++i) {
        read_dst(end_pt, pos);
        const double dx = end_pt.x - start_pt.x, dy = end_pt.y - start_pt.y;
        const double dist = std::fabs((pt.x - start_pt.x) * dy - (pt.y - start_pt.y) * dx);
        const double inner = (double)(pt.x - start_pt.x) * (end_pt.x - pt.x) + (double)(pt.y - start_pt.y) * (end_pt.y - pt.y);
        if (dist * dist <= 0.5 * eps * (dx * dx + dy * dy) && dx != 0 && dy != 0 && inner >= 0) {
            --new_count;
            dst[wpos] = start_pt = end_pt;
            if (++wpos >= cnt) wpos = 0;
            read_dst(pt, pos);
            ++i;
            continue;
        }
        dst[wpos] = start_pt = pt;
        if (++wpos >= cnt) wpos = 0;
        pt = end_pt;
    }
    if (!is_closed0) dst[wpos] = pt;
    for (int i = 0; i < new_count; ++i) { out[2 * i] = dst[i].x; out[2 * i + 1] = dst[i].y; }
    *n_out = new_count;
    return 0;
}

// cv2.boxPoints(cv2.minAreaRect(points)) for float32 points: the four corners of the minimum-area enclosing rectangle (rotating
// calipers over the convex hull's edges, db_geom.h), as float32.  The corner ORDER is this file's own (the rectangle's cycle
// starting at an arbitrary corner); the reference's only caller re-orders the corners by angle around their centre.
extern "C" int rd_min_area_rect_points(const float* pts, int n, float* out8) {
    if (!pts || !out8 || n <= 0) return 1;
    std::vector<rd_db::P2> p(n), hull(n + 1);
    for (int i = 0; i < n; ++i) p[i] = {(double)pts[2 * i], (double)pts[2 * i + 1]};
    std::sort(p.begin(), p.end(), [](const rd_db::P2& a, const rd_db::P2& b) { return a.y < b.y || (a.y == b.y && a.x < b.x); });
    p.erase(std::unique(p.begin(), p.end(), [](const rd_db::P2& a, const rd_db::P2& b) { return a.x == b.x && a.y == b.y; }), p.end());
    const int hn = rd_db::hull_from_yx_sorted(p.data(), (int)p.size(), hull.data());
    rd_db::Rect r;
    if (!rd_db::min_area_rect_hull(hull.data(), hn, r)) return 1;
    for (int k = 0; k < 4; ++k) { out8[2 * k] = (float)r.c[k].x; out8[2 * k + 1] = (float)r.c[k].y; }
    return 0;
}

// cv2.fillPoly(img, [pts], value) on a single-channel u8 image, line type 8, no shift: every edge is drawn with the 8-connected
// line iterator, then the interior is filled scanline by scanline from 16.16 fixed-point edge positions that advance by a truncated
// per-row increment (even-odd pairing of the active edges; pixels ceil(x_left) .. floor(x_right)).
extern "C" int rd_fill_poly(uint8_t* img, int h, int w, const int32_t* pts, int n, int value) {
    if (!img || h <= 0 || w <= 0 || n < 0 || (n > 0 && !pts)) return 1;
    if (n == 0) return 0;
    const uint8_t v = (uint8_t)value;
    std::vector<Edge> edges;
    Pt p0{pts[2 * (n - 1)], pts[2 * (n - 1) + 1]};
    for (int i = 0; i < n; ++i) {
        const Pt p1{pts[2 * i], pts[2 * i + 1]};
        draw_line8(img, h, w, p0, p1, v);
        if (p0.y != p1.y) {
            const int64_t x0 = (int64_t)p0.x << XY_SHIFT, x1 = (int64_t)p1.x << XY_SHIFT;
            Edge e;
            e.dx = (x1 - x0) / (p1.y - p0.y);
            if (p0.y < p1.y) { e.y0 = p0.y; e.y1 = p1.y; e.x = x0; }
            else             { e.y0 = p1.y; e.y1 = p0.y; e.x = x1; }
            edges.push_back(e);
        }
        p0 = p1;
    }
    if (edges.empty()) return 0;
    std::sort(edges.begin(), edges.end(), [](const Edge& a, const Edge& b) {
        return a.y0 != b.y0 ? a.y0 < b.y0 : (a.x != b.x ? a.x < b.x : a.dx < b.dx);
    });
    int y_max = edges[0].y1;
    for (const Edge& e : edges) y_max = std::max(y_max, e.y1);
    y_max = std::min(y_max, h);
    std::vector<Edge> active;
    size_t next = 0;
    for (int y = edges[0].y0; y < y_max; ++y) {
        active.erase(std::remove_if(active.begin(), active.end(), [&](const Edge& e) { return e.y1 == y; }), active.end());
        while (next < edges.size() && edges[next].y0 == y) active.push_back(edges[next++]);
        std::stable_sort(active.begin(), active.end(), [](const Edge& a, const Edge& b) { return a.x < b.x; });
        for (size_t k = 0; k + 1 < active.size(); k += 2) {
            if (y >= 0) {
                int x1 = (int)((active[k].x + XY_ONE - 1) >> XY_SHIFT), x2 = (int)(active[k + 1].x >> XY_SHIFT);
                if (x1 < w && x2 >= 0) {
                    x1 = std::max(x1, 0);
                    x2 = std::min(x2, w - 1);
                    for (int x = x1; x <= x2; ++x) img[(size_t)y * w + x] = v;
                }
            }
        }
        for (Edge& e : active) e.x += e.dx;
    }
    return 0;
}

// Area of a polygon's even-odd interior ([n][2] doubles).  For a simple polygon this is |shoelace| / 2 = shapely's Polygon.area.
extern "C" double rd_polygon_area(const double* a, int na) {
    return rd_polygon_intersection_area(a, na, a, na);
}

// shapely Polygon(a).intersection(Polygon(b)).area for two polygons given as vertex cycles, neither needing to be convex:
// between two consecutive x of (vertices + edge crossings) no two edges cross, so the common area of the slab is
// its width times the common length of the two interiors on the slab's middle line.  (Even-odd interiors: what the reference's
// `make_valid` - buffer(0) - does to a self-crossing ring is not reproduced.)
extern "C" double rd_polygon_intersection_area(const double* a, int na, const double* b, int nb) {
    if (!a || !b || na < 3 || nb < 3) return 0.0;
    std::vector<double> xs, ya, yb;
    slab_xs(a, na, b, nb, xs);
    double area = 0;
    for (size_t i = 0; i + 1 < xs.size(); ++i) {
        const double xm = 0.5 * (xs[i] + xs[i + 1]);
        crossings(a, na, xm, ya);
        if (b == a) { area += (xs[i + 1] - xs[i]) * overlap_1d(ya, ya); continue; }
        crossings(b, nb, xm, yb);
        area += (xs[i + 1] - xs[i]) * overlap_1d(ya, yb);
    }
    return area;
}
