// Host: chunk sizes for the recogniser's throughput mode (no reference counterpart: rapidocr chunks by a fixed rec_batch_num,
// rapid_ocr.py:430-440 - that rule is `strict` mode; this one is the engine's own scheduling decision).
//
// The persistent kernels of the recogniser backbone run whole ROUNDS of workgroup tiles on n_cu compute units: 64 lines of width 1056
// are 792 mixer tiles = 3.09 rounds on 256 CUs and cost four.  Given the padded widths of the aspect-sorted lines, the planner cuts the
// list into consecutive chunks (a chunk is padded to its last = widest line) minimising the summed cost model below - a dynamic
// programme over the cut positions, O(lines x candidate sizes).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

#include "../../include/rapiddoc_mi355.h"

namespace {

// estimated GPU time (us) of one backbone forward on n lines padded to wpad (measured per-round times at 64 x 48 x 1056,
// tools/op_profile.py); mirrored by rapiddoc_amd.ocr_host.rec_chunk_cost (tests/test_ocr_host.py keeps the two equal)
double chunk_cost(double n, double wpad, double n_cu) {
    const double t = std::floor(wpad / 8.0);
    const double px96 = n * 24 * t, px192 = n * 12 * t, px384 = n * 6 * t;
    auto rnd = [&](double x) { return std::ceil(x / n_cu - 1e-9); };
    const double mt384 = std::ceil(px384 / 256), mt192 = std::ceil(px192 / 256);
    double c = 6 * 37.0 * rnd(px192 / 128);
    c += 3 * 21.0 * std::ceil(px96 / 16 / (16 * n_cu) - 1e-9);
    c += (2 * 51.0 + 28.0 + 19.0) * rnd(mt384 * 3);
    c += 2 * 27.0 * rnd(mt384 * 6);
    c += (17.0 + 13.0) * rnd(mt192 * 2);
    return c + 0.14 * n * t + 150.0;
}

}  // namespace

extern "C" double rd_rec_chunk_cost(int n, int wpad, int n_cu) { return chunk_cost(n, wpad, n_cu > 0 ? n_cu : 256); }

extern "C" int rd_rec_plan_chunks(const int32_t* wpad_sorted, int n, int n_min, int n_max, int n_step, int n_cu, int32_t* sizes_out,
                                  int max_out, int32_t* n_out) {
    if (!n_out || n < 0 || (n > 0 && (!wpad_sorted || !sizes_out)) || n_min < 1 || n_max < n_min || n_step < 1) return 1;
    *n_out = 0;
    if (n == 0) return 0;
    const double cus = n_cu > 0 ? n_cu : 256;
    const double inf = std::numeric_limits<double>::infinity();
    std::vector<double> best((size_t)n + 1, inf);
    std::vector<int32_t> prev((size_t)n + 1, -1);
    best[0] = 0.0;
    for (int j = 1; j <= n; ++j) {
        const double w = wpad_sorted[j - 1];
        auto relax = [&](int sz) {
            const int i = j - sz;
            if (i < 0 || best[i] == inf) return;
            const double v = best[i] + chunk_cost(sz, w, cus);
            if (v < best[j]) { best[j] = v; prev[j] = i; }
        };
        for (int sz = n_min; sz <= n_max; sz += n_step) relax(sz);
        // the LAST chunk (and a list shorter than n_min) may have any size up to n_max: every line must land in a chunk
        if (j == n)
            for (int sz = 1; sz <= std::min(n_max, n); ++sz) relax(sz);
    }
    if (best[n] == inf) return 1;       // (cannot happen: sizes 1 .. n_max reach every n through the candidate sizes or the tail rule)
    std::vector<int32_t> sizes;
    for (int j = n; j > 0; j = prev[j]) sizes.push_back(j - prev[j]);
    std::reverse(sizes.begin(), sizes.end());
    if ((int)sizes.size() > max_out) return 1;
    std::copy(sizes.begin(), sizes.end(), sizes_out);
    *n_out = (int32_t)sizes.size();
    return 0;
}
