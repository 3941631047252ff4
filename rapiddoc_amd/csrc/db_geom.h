// Geometry of the DB post-process shared by the host path (db_postprocess.cpp) and the device path (kernels_dbpost.hip):
// the SAME source compiled for both, so that a rectangle is the same bits whichever side computed it (no FMA contraction
// on either side: x86-64 has none by default, the device file switches it off).
//   rapidocr DBPostProcess as patched in rapid_doc/model/ocr/ocr_patch.py:223-241: get_mini_boxes (cv2.minAreaRect +
//   boxPoints + corner ordering), unclip (pyclipper offset by area * ratio / perimeter), rescale, filter_det_res.
// PARITY UNPINNED (cv2 / pyclipper absent offline): restated from the public PaddleOCR algorithm, see db_postprocess.cpp.
#pragma once
#include <cmath>
#include <cstdint>

#if defined(__HIPCC__)
#define RD_HD __host__ __device__ inline
#else
#define RD_HD inline
#endif

namespace rd_db {

struct P2 { double x, y; };
struct Rect { P2 c[4]; double w, h; };
struct Cand { Rect r; P2 box[4]; };
struct TextBox { float pts[8]; float score; };     // == rd_text_box

RD_HD double cross(const P2& o, const P2& a, const P2& b) { return (a.x - o.x) * (b.y - o.y) - (a.y - o.y) * (b.x - o.x); }
RD_HD double dmin(double a, double b) { return a < b ? a : b; }
RD_HD double dmax(double a, double b) { return a > b ? a : b; }

// Canonical vertex order of a convex polygon given as a cycle: counter-clockwise in (x, y) numbers (positive shoelace
// area; y points down, so visually clockwise), starting at the lexicographically smallest (x, y) vertex - the order
// Andrew's monotone chain over (x, y)-sorted points returns.  In place; `h` has n >= 3 vertices.
RD_HD void canonical_cycle(P2* h, int n) {
    double a2 = 0;
    for (int i = 0; i < n; ++i) a2 += h[i].x * h[(i + 1) % n].y - h[(i + 1) % n].x * h[i].y;
    if (a2 < 0)
        for (int i = 0, j = n - 1; i < j; ++i, --j) { const P2 t = h[i]; h[i] = h[j]; h[j] = t; }
    int s = 0;
    for (int i = 1; i < n; ++i)
        if (h[i].x < h[s].x || (h[i].x == h[s].x && h[i].y < h[s].y)) s = i;
    if (s == 0) return;
    // rotate left by s: three reversals
    for (int i = 0, j = s - 1; i < j; ++i, --j) { const P2 t = h[i]; h[i] = h[j]; h[j] = t; }
    for (int i = s, j = n - 1; i < j; ++i, --j) { const P2 t = h[i]; h[i] = h[j]; h[j] = t; }
    for (int i = 0, j = n - 1; i < j; ++i, --j) { const P2 t = h[i]; h[i] = h[j]; h[j] = t; }
}

// Convex hull of points that are sorted by (y, x) with duplicates removed (the raster order of a region's row extremes).
// `out` needs room for n + 1 points; returns the vertex count, vertices in canonical order (see canonical_cycle).
// n < 3 returns the points themselves sorted by (x, y) - what the (x, y)-sorted chain returns for them.
RD_HD int hull_from_yx_sorted(const P2* pts, int n, P2* out) {
    if (n < 3) {
        for (int i = 0; i < n; ++i) out[i] = pts[i];
        if (n == 2 && (out[1].x < out[0].x || (out[1].x == out[0].x && out[1].y < out[0].y))) { const P2 t = out[0]; out[0] = out[1]; out[1] = t; }
        return n;
    }
    // Andrew's chain in the swapped plane (u = y, v = x): cross_uv = -cross_xy, so "pop while cross_uv <= 0" reads cross_xy >= 0
    int k = 0;
    for (int i = 0; i < n; ++i) {
        while (k >= 2 && cross(out[k - 2], out[k - 1], pts[i]) >= 0) --k;
        out[k++] = pts[i];
    }
    for (int i = n - 2, t = k + 1; i >= 0; --i) {
        while (k >= t && cross(out[k - 2], out[k - 1], pts[i]) >= 0) --k;
        out[k++] = pts[i];
    }
    k -= 1;                       // the last point repeats the first
    if (k >= 3) canonical_cycle(out, k);
    else if (k == 2 && (out[1].x < out[0].x || (out[1].x == out[0].x && out[1].y < out[0].y))) { const P2 t = out[0]; out[0] = out[1]; out[1] = t; }
    return k;
}

// minimum-area enclosing rectangle of a convex polygon in canonical order (rotating calipers over its edges, first minimum
// wins) - cv2.minAreaRect + boxPoints
RD_HD bool min_area_rect_hull(const P2* h, int n, Rect& out) {
    if (n == 0) return false;
    if (n == 1) {
        for (int k = 0; k < 4; ++k) out.c[k] = h[0];
        out.w = out.h = 0;
        return true;
    }
    double best = 1e300;
    for (int i = 0; i < n; ++i) {
        const P2 a = h[i], b = h[(i + 1) % n];
        double ex = b.x - a.x, ey = b.y - a.y;
        const double len = sqrt(ex * ex + ey * ey);
        if (len == 0) continue;
        ex /= len; ey /= len;
        double mn_u = 1e300, mx_u = -1e300, mn_v = 1e300, mx_v = -1e300;
        for (int j = 0; j < n; ++j) {
            const double u = (h[j].x - a.x) * ex + (h[j].y - a.y) * ey;
            const double v = -(h[j].x - a.x) * ey + (h[j].y - a.y) * ex;
            mn_u = dmin(mn_u, u); mx_u = dmax(mx_u, u);
            mn_v = dmin(mn_v, v); mx_v = dmax(mx_v, v);
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

// stable insertion sort of 4 indices by key (what std::stable_sort does for 4 elements, written out for the device)
template <typename Less>
RD_HD void stable_sort4(int idx[4], Less less) {
    for (int i = 1; i < 4; ++i) {
        const int v = idx[i];
        int j = i - 1;
        while (j >= 0 && less(v, idx[j])) { idx[j + 1] = idx[j]; --j; }
        idx[j + 1] = v;
    }
}

// PaddleOCR get_mini_boxes ordering: sort by x; left pair by y -> (tl, bl); right pair by y -> (tr, br)
RD_HD void order_mini_box(const P2 in[4], P2 out[4]) {
    int idx[4] = {0, 1, 2, 3};
    stable_sort4(idx, [&](int a, int b) { return in[a].x < in[b].x; });
    const P2 p[4] = {in[idx[0]], in[idx[1]], in[idx[2]], in[idx[3]]};
    int i1, i2, i3, i4;
    if (p[1].y > p[0].y) { i1 = 0; i4 = 1; } else { i1 = 1; i4 = 0; }
    if (p[3].y > p[2].y) { i2 = 2; i3 = 3; } else { i2 = 3; i3 = 2; }
    out[0] = p[i1]; out[1] = p[i2]; out[2] = p[i3]; out[3] = p[i4];
}

// hull of one region -> min-area rectangle candidate (rejected when its short side is below min_size)
RD_HD bool make_candidate_hull(const P2* hull, int n, int min_size, Cand& c) {
    if (!min_area_rect_hull(hull, n, c.r)) return false;
    if (dmin(c.r.w, c.r.h) < min_size) return false;
    order_mini_box(c.r.c, c.box);
    return true;
}

// score filter, unclip, second min-area rectangle, scale to the source image, filter_det_res.  Returns 1 if a box was written.
RD_HD int finish_candidate(const Cand& c, double score, int H, int W, int src_h, int src_w, float box_thresh, float unclip_ratio,
                           int min_size, TextBox* out) {
    if (box_thresh > score) return 0;
    // unclip: Polygon(box).area * ratio / Polygon(box).length; pyclipper works on integer coordinates
    const Rect& r = c.r;
    const double area = r.w * r.h, perim = 2.0 * (r.w + r.h);
    if (perim <= 0) return 0;
    const double dist = area * unclip_ratio / perim;
    P2 ip[4], ih[5];
    for (int i = 0; i < 4; ++i) ip[i] = {(double)(long)c.box[i].x, (double)(long)c.box[i].y};
    // hull of the 4 truncated corners: sort by (y, x), drop duplicates, chain
    {
        int idx[4] = {0, 1, 2, 3};
        stable_sort4(idx, [&](int a, int b) { return ip[a].y < ip[b].y || (ip[a].y == ip[b].y && ip[a].x < ip[b].x); });
        P2 sp[4];
        int m = 0;
        for (int i = 0; i < 4; ++i) {
            const P2 q = ip[idx[i]];
            if (m == 0 || q.x != sp[m - 1].x || q.y != sp[m - 1].y) sp[m++] = q;
        }
        P2 tmp[5];
        const int nh = hull_from_yx_sorted(sp, m, tmp);
        Rect ri;
        if (!min_area_rect_hull(tmp, nh, ri) || ri.w <= 0 || ri.h <= 0) return 0;
        for (int i = 0; i < 4; ++i) ih[i] = ri.c[i];
        ih[4] = {ri.w, ri.h};
    }
    // grow the rectangle by `dist` on every side (== min-area rect of the round-join offset polygon)
    P2 cen = {0, 0};
    for (int i = 0; i < 4; ++i) { cen.x += ih[i].x * 0.25; cen.y += ih[i].y * 0.25; }
    double ux = ih[1].x - ih[0].x, uy = ih[1].y - ih[0].y;
    double vx = ih[3].x - ih[0].x, vy = ih[3].y - ih[0].y;
    const double ul = sqrt(ux * ux + uy * uy), vl = sqrt(vx * vx + vy * vy);
    ux /= ul; uy /= ul; vx /= vl; vy /= vl;
    const double hu = ul * 0.5 + dist, hv = vl * 0.5 + dist;
    const P2 ex[4] = {{cen.x - hu * ux - hv * vx, cen.y - hu * uy - hv * vy}, {cen.x + hu * ux - hv * vx, cen.y + hu * uy - hv * vy},
                      {cen.x + hu * ux + hv * vx, cen.y + hu * uy + hv * vy}, {cen.x - hu * ux + hv * vx, cen.y - hu * uy + hv * vy}};
    if (dmin(2 * hu, 2 * hv) < min_size + 2) return 0;
    P2 eb[4];
    order_mini_box(ex, eb);
    // scale to the source image: np.clip(np.round(x / width * dest_width), 0, dest_width) -> int32
    long bx[4], by[4];
    for (int i = 0; i < 4; ++i) {
        bx[i] = (long)dmin(dmax(nearbyint(eb[i].x / W * src_w), 0.0), (double)src_w);
        by[i] = (long)dmin(dmax(nearbyint(eb[i].y / H * src_h), 0.0), (double)src_h);
    }
    // filter_det_res: order_points_clockwise, clip to the image, drop tiny boxes
    int idx[4] = {0, 1, 2, 3};
    stable_sort4(idx, [&](int a, int b) { return bx[a] < bx[b]; });
    int l0 = idx[0], l1 = idx[1], r0 = idx[2], r1 = idx[3];
    if (by[l1] < by[l0]) { const int t = l0; l0 = l1; l1 = t; }
    if (by[r1] < by[r0]) { const int t = r0; r0 = r1; r1 = t; }
    const int ord[4] = {l0, r0, r1, l1};  // tl, tr, br, bl
    float pts[8];
    for (int i = 0; i < 4; ++i) {
        const long cx = bx[ord[i]] < 0 ? 0 : (bx[ord[i]] > (long)src_w - 1 ? (long)src_w - 1 : bx[ord[i]]);
        const long cy = by[ord[i]] < 0 ? 0 : (by[ord[i]] > (long)src_h - 1 ? (long)src_h - 1 : by[ord[i]]);
        pts[2 * i] = (float)cx;
        pts[2 * i + 1] = (float)cy;
    }
    const int rw = (int)sqrtf((pts[0] - pts[2]) * (pts[0] - pts[2]) + (pts[1] - pts[3]) * (pts[1] - pts[3]));
    const int rh = (int)sqrtf((pts[0] - pts[6]) * (pts[0] - pts[6]) + (pts[1] - pts[7]) * (pts[1] - pts[7]));
    if (rw <= 3 || rh <= 3) return 0;
    for (int i = 0; i < 8; ++i) out->pts[i] = pts[i];
    out->score = (float)score;
    return 1;
}

}  // namespace rd_db
