// DB post-process ENTIRELY on the device: probability maps [B][H][W] in HBM -> finished text boxes, one D2H.
//   rapidocr DBPostProcess.__call__ as patched in rapid_doc/model/ocr/ocr_patch.py:223-241, called from
//   rapid_doc/model/ocr/rapid_ocr.py:537-538 - what db_postprocess.cpp does on the host, same arithmetic (db_geom.h is the
//   same source for both), box for box.
// Round 2 kept region labelling and rectangle fitting on the host: three blocking copies between the det forward and the first
// text-line crop.  Here:
//   db_row_runs_kernel<false>   per row: threshold, 2x2 dilation, number of horizontal runs of the bitmap
//   db_row_scan_kernel          per page: exclusive scan of the row counts -> where each row's runs start (RASTER ORDER, no
//                               atomics: the candidate order of the host path is the order of the regions' first pixels)
//   db_row_runs_kernel<true>    per row: the runs, written at their raster position
//   db_regions_kernel           per page, one workgroup: union-find over row-adjacent runs (8-connectivity: columns may differ
//                               by one; lock-free min-root linking, so a region's root is its first run in raster order), regions
//                               a second union-find over the GAPS between the runs of a row (4-connectivity, node 0 = everything
//                               that reaches the image frame) finds the holes: cv2.findContours(RETR_LIST) returns their borders
//                               as contours too.  Contours (outer borders and hole borders) ranked by their start pixel, per
//                               (contour, row) the extreme columns by integer atomicMin / atomicMax (deterministic), then one thread
//                               per contour: convex hull of the row extremes (the hull of a point set is the hull of its rows' end
//                               points), min-area rectangle, min-side filter -> candidates
//   db_scores_kernel            (kernels_image.hip) box_score_fast of every candidate
//   db_finish_kernel            per page: score filter, unclip, rescale, filter_det_res, order-preserving compaction
// A page whose bitmap has more runs than the buffer holds raises an overflow flag: the caller repeats the batch on the host path.
#pragma clang fp contract(off)
#include <hip/hip_runtime.h>

#include <cstdint>

#include "db_geom.h"
#include "rd_kernels.h"

namespace {

using rd_db::Cand;
using rd_db::P2;

struct DbRun { int16_t y, x0, x1, pad; };
struct DbCand { double box[8], rect[8], w, h; };     // == DbCand of kernels_image.hip (db_scores_kernel)

// block-wide exclusive scan of one int per thread (1024 threads max); returns the exclusive prefix, *total = block sum
template <int NT>
__device__ int block_exclusive_scan(int v, int* smem, int* total) {
    const int tid = threadIdx.x;
    smem[tid] = v;
    __syncthreads();
    for (int off = 1; off < NT; off <<= 1) {
        const int t = tid >= off ? smem[tid - off] : 0;
        __syncthreads();
        smem[tid] += t;
        __syncthreads();
    }
    const int incl = smem[tid];
    *total = smem[NT - 1];
    __syncthreads();
    return incl - v;
}

// ---- rows -> runs
template <bool WRITE>
__global__ void __launch_bounds__(256) db_row_runs_kernel(const float* __restrict__ prob, int H, int W, float thresh, int dilate,
                                                          int32_t* __restrict__ row_cnt, const int32_t* __restrict__ row_off,
                                                          DbRun* __restrict__ runs, int max_runs) {
    extern __shared__ unsigned char rowm[];     // dilated bitmap of this row, then [256] ints of scan scratch
    int* scan = reinterpret_cast<int*>(rowm + ((W + 3) & ~3));
    const int y = blockIdx.x, b = blockIdx.y;
    const float* p1 = prob + ((size_t)b * H + y) * W;
    const float* p0 = p1 - W;
    for (int x = threadIdx.x; x < W; x += 256) {
        bool v = p1[x] > thresh;
        if (dilate) {      // cv2.dilate 2x2, anchor (1,1): max over (y-1..y, x-1..x)
            if (x > 0) v = v || p1[x - 1] > thresh;
            if (y > 0) {
                v = v || p0[x] > thresh;
                if (x > 0) v = v || p0[x - 1] > thresh;
            }
        }
        rowm[x] = v;
    }
    __syncthreads();
    // thread t owns the contiguous columns [t * per, (t + 1) * per): run starts inside are counted, then ranked by a block scan
    const int per = (W + 255) / 256;
    const int xa = threadIdx.x * per, xb = min(W, xa + per);
    int mine = 0;
    for (int x = xa; x < xb; ++x) mine += rowm[x] && (x == 0 || !rowm[x - 1]);
    int total;
    const int before = block_exclusive_scan<256>(mine, scan, &total);
    if (!WRITE) {
        if (threadIdx.x == 0) row_cnt[(size_t)b * H + y] = total;
        return;
    }
    int slot = row_off[(size_t)b * (H + 1) + y] + before;
    for (int x = xa; x < xb; ++x) {
        if (rowm[x] && (x == 0 || !rowm[x - 1])) {
            int x1 = x;
            while (x1 + 1 < W && rowm[x1 + 1]) ++x1;
            if (slot < max_runs) runs[(size_t)b * max_runs + slot] = DbRun{(int16_t)y, (int16_t)x, (int16_t)x1, 0};
            ++slot;
        }
    }
}

__global__ void __launch_bounds__(1024) db_row_scan_kernel(const int32_t* __restrict__ row_cnt, int H, int32_t* __restrict__ row_off,
                                                           int32_t* __restrict__ n_runs, int max_runs, int32_t* __restrict__ overflow) {
    __shared__ int smem[1024];
    const int b = blockIdx.x;
    int carry = 0;
    for (int y0 = 0; y0 < H; y0 += 1024) {
        const int y = y0 + threadIdx.x;
        const int v = y < H ? row_cnt[(size_t)b * H + y] : 0;
        int total;
        const int ex = block_exclusive_scan<1024>(v, smem, &total);
        if (y < H) row_off[(size_t)b * (H + 1) + y] = carry + ex;
        carry += total;
    }
    if (threadIdx.x == 0) {
        row_off[(size_t)b * (H + 1) + H] = carry;
        n_runs[b] = carry;
        if (carry > max_runs) atomicOr(overflow, 1);
    }
}

// ---- runs -> regions -> candidates
// (contour, row) slots per run: a region has at most one row per run; a hole of g gaps has at most g + 2 rows (its border reaches
// one row above and below) and there are fewer gaps than runs
constexpr int kExt = 4;
struct RegionsWs {
    const DbRun* runs; const int32_t* row_off; const int32_t* n_runs;
    int32_t* parent;        // [B][max_runs]
    int32_t* comp_of;       // [B][max_runs]  contour index of a ROOT run, -1 otherwise
    int32_t* gparent;       // [B][max_runs + 1]  gaps: node i + 1 = the gap between runs i and i + 1 of one row; node 0 = outside
    int32_t* gcomp_of;      // [B][max_runs + 1]  contour index of a hole's ROOT gap, -1 otherwise
    int32_t* ext_l; int32_t* ext_r;   // [B][kExt * max_runs]  per (contour, row) extreme columns
    int32_t* comp_y0; int32_t* comp_y1; int32_t* comp_off;   // [B][max_cand + 1]
    int2* pts; P2* hullbuf;           // [B][2 * kExt * max_runs], [B][2 * kExt * max_runs + max_cand]
    DbCand* cand; int32_t* n_cand;    // [B][max_cand], [B]
    int max_runs, max_cand, H, min_size;
};

// Values that other wavefronts change with atomics (which execute in L2) are read with agent-scope atomic loads: a plain load
// may be served from a line the CU's vector L1 cached before the atomic.
__device__ __forceinline__ int lda(const int32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ __forceinline__ int uf_find(int32_t* parent, int i) {
    int p = lda(&parent[i]);
    while (p != i) {
        i = p;
        p = lda(&parent[i]);
    }
    return i;
}
__device__ __forceinline__ void uf_union(int32_t* parent, int a, int b) {
    // lock-free: link the larger root under the smaller one; a root only ever changes to a smaller index
    for (;;) {
        a = uf_find(parent, a);
        b = uf_find(parent, b);
        if (a == b) return;
        if (a > b) { const int t = a; a = b; b = t; }
        const int old = atomicMin(&parent[b], a);
        if (old == b) return;
        b = old;
    }
}

// first run of [lo, hi) (one row: sorted, disjoint) whose x1 >= x
__device__ __forceinline__ int first_run_reaching(const DbRun* runs, int lo, int hi, int x) {
    while (lo < hi) {
        const int m = (lo + hi) >> 1;
        if (runs[m].x1 < x) lo = m + 1;
        else hi = m;
    }
    return lo;
}

__global__ void __launch_bounds__(1024) db_regions_kernel(RegionsWs w) {
    __shared__ int smem[1024];
    __shared__ int s_ncomp;
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(w.n_runs[b], w.max_runs);
    const DbRun* runs = w.runs + (size_t)b * w.max_runs;
    const int32_t* row_off = w.row_off + (size_t)b * (w.H + 1);
    int32_t* parent = w.parent + (size_t)b * w.max_runs;
    int32_t* comp_of = w.comp_of + (size_t)b * w.max_runs;
    int32_t* gparent = w.gparent + (size_t)b * (w.max_runs + 1);
    int32_t* gcomp_of = w.gcomp_of + (size_t)b * (w.max_runs + 1);
    int32_t* ext_l = w.ext_l + (size_t)b * kExt * w.max_runs;
    int32_t* ext_r = w.ext_r + (size_t)b * kExt * w.max_runs;
    int32_t* cy0 = w.comp_y0 + (size_t)b * (w.max_cand + 1);
    int32_t* cy1 = w.comp_y1 + (size_t)b * (w.max_cand + 1);
    int32_t* coff = w.comp_off + (size_t)b * (w.max_cand + 1);
    if (w.n_runs[b] > w.max_runs) {          // overflow: the host path takes this batch
        if (tid == 0) w.n_cand[b] = 0;
        return;
    }
    // the gap behind run i (node i + 1) exists when run i + 1 lies in the same row
    auto has_gap = [&](int i) { return i + 1 < n && runs[i + 1].y == runs[i].y; };
    for (int i = tid; i < n; i += 1024) { parent[i] = i; comp_of[i] = -1; }
    for (int i = tid; i <= n; i += 1024) { gparent[i] = i; gcomp_of[i] = -1; }
    __syncthreads();
    // 1. union every run with the runs of the previous row it touches (columns may differ by one: 8-connectivity); union every
    //    gap with the background of the rows above and below it (same columns only: the background is 4-connected)
    for (int i = tid; i < n; i += 1024) {
        const DbRun r = runs[i];
        if (r.y > 0) {
            const int lo = row_off[r.y - 1], hi = row_off[r.y];
            for (int t = first_run_reaching(runs, lo, hi, r.x0 - 1); t < hi && runs[t].x0 <= r.x1 + 1; ++t) uf_union(parent, t, i);
        }
        if (!has_gap(i)) continue;
        const int g = i + 1, gx0 = r.x1 + 1, gx1 = runs[i + 1].x0 - 1;
        if (r.y == 0 || r.y == w.H - 1) uf_union(gparent, 0, g);          // on the image frame
        for (int yy = r.y - 1; yy <= r.y + 1; yy += 2) {
            if (yy < 0 || yy >= w.H) continue;
            const int lo = row_off[yy], hi = row_off[yy + 1];
            if (lo == hi || gx0 < runs[lo].x0 || gx1 > runs[hi - 1].x1) uf_union(gparent, 0, g);    // empty row, or its leading / trailing background
            if (yy > r.y) continue;
            // gap t + 1 = (runs[t].x1, runs[t + 1].x0) meets [gx0, gx1] when runs[t + 1].x0 > gx0 and runs[t].x1 < gx1
            for (int t = max(lo, first_run_reaching(runs, lo, hi, gx0) - 1); t + 1 < hi && runs[t].x1 < gx1; ++t)
                if (runs[t + 1].x0 > gx0) uf_union(gparent, t + 1, g);
        }
    }
    __syncthreads();
    // 2. flatten
    for (int i = tid; i < n; i += 1024) parent[i] = uf_find(parent, i);
    for (int i = tid; i <= n; i += 1024) gparent[i] = uf_find(gparent, i);
    __syncthreads();
    // 3. contours ranked by their start pixel: the region whose first run is i starts at (y, x0_i), the hole whose first gap lies
    //    behind run i at (y, x1_i) (cv2 starts a hole border on the region pixel left of the hole's first pixel); a run cannot
    //    start both, so raster order = (run index, region before hole)
    if (tid == 0) s_ncomp = 0;
    __syncthreads();
    {
        int carry = 0;
        for (int i0 = 0; i0 < n; i0 += 1024) {
            const int i = i0 + tid;
            const int is_root = i < n && lda(&parent[i]) == i;
            const int is_hole = i < n && has_gap(i) && lda(&gparent[i + 1]) == i + 1;     // (a root other than node 0)
            int total;
            const int ex = block_exclusive_scan<1024>(is_root + is_hole, smem, &total);
            if (is_root) {
                const int c = carry + ex;
                comp_of[i] = c;
                if (c < w.max_cand) { cy0[c] = runs[i].y; cy1[c] = runs[i].y; }
            }
            if (is_hole) {
                const int c = carry + ex + is_root;
                gcomp_of[i + 1] = c;
                if (c < w.max_cand) { cy0[c] = runs[i].y - 1; cy1[c] = runs[i].y + 1; }
            }
            carry += total;
        }
        if (tid == 0) s_ncomp = carry;
    }
    __syncthreads();
    const int ncomp = min(s_ncomp, w.max_cand);      // the reference looks at the first max_candidates contours only
    // 4. last row of every contour
    for (int i = tid; i < n; i += 1024) {
        const int c = comp_of[lda(&parent[i])];
        if (c < ncomp) atomicMax(&cy1[c], (int)runs[i].y);
        if (has_gap(i)) {
            const int hc = gcomp_of[lda(&gparent[i + 1])];
            if (hc >= 0 && hc < ncomp) atomicMax(&cy1[hc], (int)runs[i].y + 1);
        }
    }
    __syncthreads();
    // 5. storage offsets of the per-(contour, row) extremes: exclusive scan of the contour heights
    {
        int carry = 0;
        for (int c0 = 0; c0 < ncomp; c0 += 1024) {
            const int c = c0 + tid;
            const int hgt = c < ncomp ? lda(&cy1[c]) - cy0[c] + 1 : 0;
            int total;
            const int ex = block_exclusive_scan<1024>(hgt, smem, &total);
            if (c < ncomp) coff[c] = carry + ex;
            carry += total;
        }
        if (tid == 0) coff[ncomp] = carry;       // <= kExt * n
    }
    __syncthreads();
    const int n_ext = coff[ncomp];
    for (int i = tid; i < n_ext; i += 1024) { ext_l[i] = 0x7fffffff; ext_r[i] = -1; }
    __syncthreads();
    for (int i = tid; i < n; i += 1024) {
        const DbRun r = runs[i];
        const int c = comp_of[lda(&parent[i])];
        if (c < ncomp) {
            const int k = coff[c] + (r.y - cy0[c]);
            atomicMin(&ext_l[k], (int)r.x0);
            atomicMax(&ext_r[k], (int)r.x1);
        }
        if (!has_gap(i)) continue;
        const int hc = gcomp_of[lda(&gparent[i + 1])];
        if (hc < 0 || hc >= ncomp) continue;
        // the hole's border = the region pixels 4-adjacent to it: the two pixels that close this gap, and the region pixels
        // right above / below it
        const int gx0 = r.x1 + 1, gx1 = runs[i + 1].x0 - 1;
        const int k = coff[hc] + (r.y - cy0[hc]);
        atomicMin(&ext_l[k], gx0 - 1);
        atomicMax(&ext_r[k], gx1 + 1);
        for (int yy = r.y - 1; yy <= r.y + 1; yy += 2) {
            const int lo = row_off[yy], hi = row_off[yy + 1];
            const int t = first_run_reaching(runs, lo, hi, gx0);
            if (t >= hi || runs[t].x0 > gx1) continue;
            int u = t;
            while (u + 1 < hi && runs[u + 1].x0 <= gx1) ++u;
            const int kk = coff[hc] + (yy - cy0[hc]);
            atomicMin(&ext_l[kk], max(gx0, (int)runs[t].x0));
            atomicMax(&ext_r[kk], min(gx1, (int)runs[u].x1));
        }
    }
    __syncthreads();
    // 6. one thread per contour: points sorted by (y, x) -> hull -> min-area rectangle -> candidate (or not)
    int2* pts_all = w.pts + (size_t)b * 2 * kExt * w.max_runs;
    P2* hull_all = w.hullbuf + (size_t)b * (2 * (size_t)kExt * w.max_runs + w.max_cand);
    int carry = 0;
    for (int c0 = 0; c0 < ncomp; c0 += 1024) {
        const int c = c0 + tid;
        Cand cd;
        int ok = 0;
        if (c < ncomp) {
            const int base = coff[c], hgt = lda(&cy1[c]) - cy0[c] + 1;
            // this region's scratch: 2 * hgt points (int2) at pts[2 * base], the hull (at most 2 * hgt + 1 vertices while the chain
            // runs) at hullbuf[2 * base + c]; the slices of different regions do not overlap
            int2* pl = pts_all + 2 * (size_t)base;
            P2* hin = hull_all + 2 * (size_t)base + c;
            int m = 0;
            for (int ry = 0; ry < hgt; ++ry) {           // row extremes in raster order = sorted by (y, x), no duplicates
                const int l = lda(&ext_l[base + ry]), r = lda(&ext_r[base + ry]);
                if (r < 0) continue;                      // (cannot happen: every row of a contour holds a border pixel)
                pl[m++] = make_int2(l, cy0[c] + ry);
                if (r != l) pl[m++] = make_int2(r, cy0[c] + ry);
            }
            // hull (output: m + 1 points at most; the slice [2 * base + c, 2 * (base + hgt) + c + 1) of hullbuf is this region's)
            int k = 0;
            if (m < 3) {
                for (int i = 0; i < m; ++i) hin[i] = {(double)pl[i].x, (double)pl[i].y};
                if (m == 2 && (hin[1].x < hin[0].x || (hin[1].x == hin[0].x && hin[1].y < hin[0].y))) { const P2 t = hin[0]; hin[0] = hin[1]; hin[1] = t; }
                k = m;
            } else {
                for (int i = 0; i < m; ++i) {
                    const P2 q = {(double)pl[i].x, (double)pl[i].y};
                    while (k >= 2 && rd_db::cross(hin[k - 2], hin[k - 1], q) >= 0) --k;
                    hin[k++] = q;
                }
                for (int i = m - 2, t = k + 1; i >= 0; --i) {
                    const P2 q = {(double)pl[i].x, (double)pl[i].y};
                    while (k >= t && rd_db::cross(hin[k - 2], hin[k - 1], q) >= 0) --k;
                    hin[k++] = q;
                }
                k -= 1;
                if (k >= 3) rd_db::canonical_cycle(hin, k);
                else if (k == 2 && (hin[1].x < hin[0].x || (hin[1].x == hin[0].x && hin[1].y < hin[0].y))) { const P2 t = hin[0]; hin[0] = hin[1]; hin[1] = t; }
            }
            ok = rd_db::make_candidate_hull(hin, k, w.min_size, cd) ? 1 : 0;
        }
        int total;
        const int ex = block_exclusive_scan<1024>(ok, smem, &total);
        if (ok) {
            DbCand& o = w.cand[(size_t)b * w.max_cand + carry + ex];
            for (int k = 0; k < 4; ++k) {
                o.box[2 * k] = cd.box[k].x; o.box[2 * k + 1] = cd.box[k].y;
                o.rect[2 * k] = cd.r.c[k].x; o.rect[2 * k + 1] = cd.r.c[k].y;
            }
            o.w = cd.r.w;
            o.h = cd.r.h;
        }
        carry += total;
    }
    if (tid == 0) w.n_cand[b] = carry;
}

// ---- candidates + scores -> finished boxes (order preserved)
__global__ void __launch_bounds__(1024) db_finish_kernel(const DbCand* __restrict__ cand, const double* __restrict__ scores,
                                                         const int32_t* __restrict__ n_cand, int max_cand, int H, int W,
                                                         const int32_t* __restrict__ src_hw, float box_thresh, float unclip_ratio, int min_size,
                                                         rd_db::TextBox* __restrict__ out, int max_out, int32_t* __restrict__ n_out) {
    __shared__ int smem[1024];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int n = min(n_cand[b], max_cand);
    const int sh = src_hw[2 * b], sw = src_hw[2 * b + 1];
    int carry = 0;
    for (int i0 = 0; i0 < n; i0 += 1024) {
        const int i = i0 + tid;
        rd_db::TextBox tb;
        int ok = 0;
        if (i < n) {
            const DbCand& q = cand[(size_t)b * max_cand + i];
            Cand c;
            for (int k = 0; k < 4; ++k) {
                c.box[k] = {q.box[2 * k], q.box[2 * k + 1]};
                c.r.c[k] = {q.rect[2 * k], q.rect[2 * k + 1]};
            }
            c.r.w = q.w;
            c.r.h = q.h;
            ok = rd_db::finish_candidate(c, scores[(size_t)b * max_cand + i], H, W, sh, sw, box_thresh, unclip_ratio, min_size, &tb);
        }
        int total;
        const int ex = block_exclusive_scan<1024>(ok, smem, &total);
        if (ok && carry + ex < max_out) out[(size_t)b * max_out + carry + ex] = tb;
        carry += total;
    }
    if (tid == 0) n_out[b] = min(carry, max_out);
}

}  // namespace

namespace rd {

// workspace layout (all offsets 256-byte aligned); see rd_db_boxes_workspace
struct DbWsLayout {
    size_t row_cnt, row_off, n_runs, runs, parent, comp_of, gparent, gcomp_of, ext_l, ext_r, cy0, cy1, coff, pts, hull, cand, n_cand, scores, total;
};
static DbWsLayout db_ws_layout(int B, int H, int max_runs, int max_cand) {
    DbWsLayout L{};
    size_t off = 0;
    auto take = [&](size_t bytes) { const size_t o = off; off += (bytes + 255) & ~size_t(255); return o; };
    L.row_cnt = take((size_t)B * H * 4);
    L.row_off = take((size_t)B * (H + 1) * 4);
    L.n_runs = take((size_t)(B + 1) * 4);                    // [B] counts + the overflow flag
    L.runs = take((size_t)B * max_runs * sizeof(DbRun));
    L.parent = take((size_t)B * max_runs * 4);
    L.comp_of = take((size_t)B * max_runs * 4);
    L.gparent = take((size_t)B * (max_runs + 1) * 4);
    L.gcomp_of = take((size_t)B * (max_runs + 1) * 4);
    L.ext_l = take((size_t)B * kExt * max_runs * 4);
    L.ext_r = take((size_t)B * kExt * max_runs * 4);
    L.cy0 = take((size_t)B * (max_cand + 1) * 4);
    L.cy1 = take((size_t)B * (max_cand + 1) * 4);
    L.coff = take((size_t)B * (max_cand + 1) * 4);
    L.pts = take((size_t)B * 2 * kExt * max_runs * sizeof(int2));
    L.hull = take((size_t)B * (2 * (size_t)kExt * max_runs + max_cand) * sizeof(rd_db::P2));
    L.cand = take((size_t)B * max_cand * sizeof(DbCand));
    L.n_cand = take((size_t)B * 4);
    L.scores = take((size_t)B * max_cand * 8);
    L.total = off;
    return L;
}
size_t db_boxes_workspace_bytes(int B, int H, int max_runs, int max_cand) { return db_ws_layout(B, H, max_runs, max_cand).total; }

int launch_db_boxes(const float* prob, int B, int H, int W, const int32_t* src_hw_dev, float thresh, float box_thresh, float unclip_ratio,
                    int dilate, int max_cand, int max_runs, void* ws, size_t ws_bytes, void* out_boxes, int max_out, int32_t* n_out_dev,
                    hipStream_t s) {
    if (B <= 0) return 0;
    if (W > 32767 || H > 32767 || max_cand <= 0 || max_runs <= 0 || max_out <= 0) return 1;
    const DbWsLayout L = db_ws_layout(B, H, max_runs, max_cand);
    if (!ws || ws_bytes < L.total) return 1;
    uint8_t* base = static_cast<uint8_t*>(ws);
    int32_t* row_cnt = reinterpret_cast<int32_t*>(base + L.row_cnt);
    int32_t* row_off = reinterpret_cast<int32_t*>(base + L.row_off);
    int32_t* n_runs = reinterpret_cast<int32_t*>(base + L.n_runs);
    int32_t* overflow = n_runs + B;
    DbRun* runs = reinterpret_cast<DbRun*>(base + L.runs);
    (void)hipMemsetAsync(overflow, 0, sizeof(int32_t), s);
    const size_t sh = (size_t)((W + 3) & ~3) + 256 * sizeof(int);
    hipLaunchKernelGGL(db_row_runs_kernel<false>, dim3(H, B), dim3(256), sh, s, prob, H, W, thresh, dilate, row_cnt, row_off, runs, max_runs);
    hipLaunchKernelGGL(db_row_scan_kernel, dim3(B), dim3(1024), 0, s, row_cnt, H, row_off, n_runs, max_runs, overflow);
    hipLaunchKernelGGL(db_row_runs_kernel<true>, dim3(H, B), dim3(256), sh, s, prob, H, W, thresh, dilate, row_cnt, row_off, runs, max_runs);
    RegionsWs w{};
    w.runs = runs; w.row_off = row_off; w.n_runs = n_runs;
    w.parent = reinterpret_cast<int32_t*>(base + L.parent);
    w.comp_of = reinterpret_cast<int32_t*>(base + L.comp_of);
    w.gparent = reinterpret_cast<int32_t*>(base + L.gparent);
    w.gcomp_of = reinterpret_cast<int32_t*>(base + L.gcomp_of);
    w.ext_l = reinterpret_cast<int32_t*>(base + L.ext_l);
    w.ext_r = reinterpret_cast<int32_t*>(base + L.ext_r);
    w.comp_y0 = reinterpret_cast<int32_t*>(base + L.cy0);
    w.comp_y1 = reinterpret_cast<int32_t*>(base + L.cy1);
    w.comp_off = reinterpret_cast<int32_t*>(base + L.coff);
    w.pts = reinterpret_cast<int2*>(base + L.pts);
    w.hullbuf = reinterpret_cast<rd_db::P2*>(base + L.hull);
    w.cand = reinterpret_cast<DbCand*>(base + L.cand);
    w.n_cand = reinterpret_cast<int32_t*>(base + L.n_cand);
    w.max_runs = max_runs; w.max_cand = max_cand; w.H = H; w.min_size = 3;
    hipLaunchKernelGGL(db_regions_kernel, dim3(B), dim3(1024), 0, s, w);
    double* scores = reinterpret_cast<double*>(base + L.scores);
    if (launch_db_scores(prob, B, H, W, w.cand, w.n_cand, max_cand, scores, s) != 0) return 1;
    hipLaunchKernelGGL(db_finish_kernel, dim3(B), dim3(1024), 0, s, w.cand, scores, w.n_cand, max_cand, H, W, src_hw_dev, box_thresh,
                       unclip_ratio, 3, reinterpret_cast<rd_db::TextBox*>(out_boxes), max_out, n_out_dev);
    // the overflow flag rides behind the box counts: n_out_dev [B + 1]
    (void)hipMemcpyAsync(n_out_dev + B, overflow, sizeof(int32_t), hipMemcpyDeviceToDevice, s);
    return 0;
}

}  // namespace rd
