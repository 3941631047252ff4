"""Known-answer tests of the raster / polygon primitives behind the layout polygon branch (rapiddoc_amd/csrc/polygon_ops.cpp).
They stand where cv2 / shapely stand in the reference; neither library is installed here, so these are NOT comparisons with
OpenCV or GEOS (parity unpinned) but with answers that follow from the published algorithms: the border-following order and
corner steps of Suzuki-Abe tracing, shoelace areas, exact intersection areas of hand-made polygons, inclusive polygon fill."""
import numpy as np
import pytest

from rapiddoc_amd import _lib

pytestmark = pytest.mark.skipif(not _lib.LIB_PATH.exists(), reason="librapiddoc_mi355.so not built (host C++ primitives)")


@pytest.fixture(scope="module")
def LP():
    from rapiddoc_amd import layout_polygon
    return layout_polygon


def test_external_contour_of_a_rectangle_runs_down_the_left_side_first(LP):
    m = np.zeros((20, 30), np.uint8)
    m[5:15, 4:24] = 1
    (c,) = LP.find_external_contours(m)
    assert c.tolist() == [[4, 5], [4, 14], [23, 14], [23, 5]]          # start = first pixel in raster order, then counter-clockwise on screen
    assert LP.contour_area(c) == 19 * 9                                # the contour joins pixel CENTRES
    assert LP.arc_length(c, True) == 2 * (19 + 9) and LP.arc_length(c, False) == 9 + 19 + 9


def test_concave_corner_is_cut_diagonally_by_8_connected_tracing(LP):
    m = np.zeros((40, 40), np.uint8)
    m[5:35, 5:15] = 1
    m[25:35, 5:35] = 1                                                  # an L
    (c,) = LP.find_external_contours(m)
    assert c.tolist() == [[5, 5], [5, 34], [34, 34], [34, 25], [15, 25], [14, 24], [14, 5]]
    assert LP.contour_area(c) == 29 * 29 - 20 * 20 + 0.5 + 0                # shoelace of that ring: the cut corner adds half a pixel


def test_single_pixels_frame_contact_and_ordering(LP):
    m = np.zeros((10, 12), np.uint8)
    m[0, 0] = 1                                                         # touches the frame
    m[4, 6] = 1
    m[8:10, 9:12] = 1                                                   # bottom-right corner block
    cs = LP.find_external_contours(m)
    assert [c.tolist() for c in cs] == [[[9, 8], [9, 9], [11, 9], [11, 8]], [[6, 4]], [[0, 0]]]     # last found first
    assert LP.contour_area(cs[1]) == 0 and LP.arc_length(cs[1], True) == 0
    assert LP.find_external_contours(np.zeros((5, 5), np.uint8)) == []


def test_component_inside_a_hole_is_not_external(LP):
    m = np.zeros((30, 30), np.uint8)
    m[2:28, 2:28] = 1
    m[6:24, 6:24] = 0                                                   # a ring ...
    m[12:18, 12:18] = 1                                                 # ... with an island in its hole
    cs = LP.find_external_contours(m)
    assert len(cs) == 1 and cs[0].tolist() == [[2, 2], [2, 27], [27, 27], [27, 2]]
    m[15, 4:12] = 0                                                     # a notch in the ring's left wall that does not cut through it
    assert len(LP.find_external_contours(m)) == 1
    m[15, 2:4] = 0                                                      # cut through: the hole's background is the outside now
    cs = LP.find_external_contours(m)
    assert len(cs) == 2 and cs[0].tolist() == [[12, 12], [12, 17], [17, 17], [17, 12]]


def test_diagonal_line_is_walked_out_and_back(LP):
    m = np.eye(6, dtype=np.uint8)
    (c,) = LP.find_external_contours(m)
    assert c.tolist() == [[0, 0], [5, 5]]                                # SIMPLE keeps the two turning points
    assert LP.contour_area(c) == 0


def test_approx_poly_dp_on_known_shapes(LP):
    # every pixel centre of a square's border (what CHAIN_APPROX_NONE would give): the four corners remain
    ring = [[x, 0] for x in range(0, 10)] + [[10, y] for y in range(0, 10)] + [[x, 10] for x in range(10, 0, -1)] + [[0, y] for y in range(10, 0, -1)]
    out = LP.approx_poly_dp(np.int32(ring), 1.0, True)
    assert sorted(map(tuple, out.tolist())) == [(0, 0), (0, 10), (10, 0), (10, 10)]
    # a bump lower than epsilon disappears, a higher one stays
    bump = lambda h: np.int32([[0, 0], [50, 0], [100, 0], [100, 60], [50, 60 + h], [0, 60]])     # noqa: E731
    assert len(LP.approx_poly_dp(bump(2), 3.0, True)) == 4
    assert len(LP.approx_poly_dp(bump(8), 3.0, True)) == 5
    # epsilon 0 keeps every turning point; a huge epsilon collapses the ring to a point
    assert len(LP.approx_poly_dp(bump(8), 0.0, True)) == 5
    assert len(LP.approx_poly_dp(bump(8), 1e6, True)) == 1
    # open curve: the ends always stay
    zig = np.int32([[0, 0], [10, 1], [20, 0], [30, 1], [40, 0]])
    assert LP.approx_poly_dp(zig, 2.0, False).tolist() == [[0, 0], [40, 0]]
    assert LP.approx_poly_dp(zig, 0.5, False).tolist() == zig.tolist()


def test_approx_poly_dp_of_a_disc_stays_within_epsilon(LP):
    yy, xx = np.mgrid[:120, :120]
    m = ((yy - 60) ** 2 + (xx - 60) ** 2 <= 50 ** 2).astype(np.uint8)
    (c,) = LP.find_external_contours(m)
    eps = 0.004 * LP.arc_length(c, True)
    a = LP.approx_poly_dp(c, eps, True)
    assert 12 <= len(a) <= 40
    assert {tuple(p) for p in a.tolist()} <= {tuple(p) for p in c.tolist()}          # a subset of the contour's points
    # every contour point lies within eps of the approximating polygon
    def dist(p, s, e):
        s, e, p = map(np.float64, (s, e, p))
        t = np.clip(np.dot(p - s, e - s) / max(np.dot(e - s, e - s), 1e-12), 0, 1)
        return np.linalg.norm(p - (s + t * (e - s)))
    worst = max(min(dist(p, a[i], a[(i + 1) % len(a)]) for i in range(len(a))) for p in c)
    assert worst <= eps + 1e-9


def test_min_area_rectangle_of_a_rotated_rectangle(LP):
    th = np.deg2rad(27.0)
    R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
    corners = (np.array([[-40, -10], [40, -10], [40, 10], [-40, 10]]) @ R.T + [100, 80]).astype(np.float32)
    inner = (np.random.default_rng(0).uniform(-1, 1, (50, 2)) * [40, 10]) @ R.T + [100, 80]
    box = LP.min_area_rect_points(np.concatenate([corners, inner.astype(np.float32)]))
    assert box.dtype == np.float32 and box.shape == (4, 2)
    d = np.abs(box[:, None, :] - corners[None, :, :]).sum(-1)
    assert (d.min(axis=1) < 1e-3).all() and sorted(d.argmin(axis=1).tolist()) == [0, 1, 2, 3]


def test_polygon_areas_and_intersections(LP):
    sq = [[0, 0], [10, 0], [10, 10], [0, 10]]
    assert LP.polygon_area(sq) == 100 and LP.polygon_area(sq[::-1]) == 100
    assert LP.polygon_intersection_area(sq, [[5, 5], [15, 5], [15, 15], [5, 15]]) == 25
    assert LP.polygon_intersection_area(sq, [[20, 0], [30, 0], [30, 10], [20, 10]]) == 0
    assert LP.polygon_intersection_area(sq, [[2, 2], [4, 2], [4, 4], [2, 4]]) == 4                 # containment
    assert LP.polygon_intersection_area(sq, [[10, 0], [20, 0], [20, 10], [10, 10]]) == 0           # a shared edge has no area
    ell = [[0, 0], [10, 0], [10, 4], [4, 4], [4, 10], [0, 10]]                                    # concave
    assert LP.polygon_area(ell) == 64
    assert LP.polygon_intersection_area(ell, [[5, 5], [15, 5], [15, 15], [5, 15]]) == 0
    assert LP.polygon_intersection_area(ell, [[2, 2], [8, 2], [8, 8], [2, 8]]) == 6 * 6 - 4 * 4
    comb_a = [[0, 0], [12, 0], [12, 10], [8, 10], [8, 2], [4, 2], [4, 10], [0, 10]]               # a "U" upside down ...
    comb_b = [[0, 4], [12, 4], [12, 6], [0, 6]]                                                  # ... crossed by a bar: two pieces
    assert LP.polygon_intersection_area(comb_a, comb_b) == 2 * (4 * 2)
    diamond = [[5, -2], [12, 5], [5, 12], [-2, 5]]                                               # |x - 5| + |y - 5| <= 7, area 98
    assert abs(LP.polygon_intersection_area(sq, diamond) - (98 - 4 * 4)) < 1e-9                  # minus the four tips outside the square
    assert LP.polygon_intersection_area(sq, [[5, -5], [15, 5], [5, 15], [-5, 5]]) == 100          # the square inside a larger diamond
    assert abs(LP.polygon_overlap_ratio(sq, [[5, 5], [15, 5], [15, 15], [5, 15]], "union") - 25 / 175) < 1e-12
    assert LP.polygon_overlap_ratio(sq, [[2, 2], [4, 2], [4, 4], [2, 4]], "small") == 1.0
    assert LP.polygon_overlap_ratio(sq, [[2, 2], [4, 2], [4, 4], [2, 4]], "large") == 0.04


def test_intersection_area_against_a_dense_grid_estimate(LP):
    rng = np.random.default_rng(3)

    def star(cx, cy):
        ang = np.sort(rng.uniform(0, 2 * np.pi, 11))
        rad = rng.uniform(15, 45, 11)
        return np.stack([cx + rad * np.cos(ang), cy + rad * np.sin(ang)], 1)

    def inside(poly, x, y):                                            # even-odd crossing test on pixel centres
        n, res = len(poly), np.zeros(x.shape, bool)
        for i in range(n):
            (x0, y0), (x1, y1) = poly[i], poly[(i + 1) % n]
            cond = (y0 > y) != (y1 > y)
            xi = x0 + (y - y0) * (x1 - x0) / np.where(y1 == y0, 1, y1 - y0)
            res ^= cond & (x < xi)
        return res

    yy, xx = np.mgrid[0:100:0.25, 0:100:0.25]
    for _ in range(5):
        a, b = star(45, 50), star(55, 50)
        est = (inside(a, xx, yy) & inside(b, xx, yy)).sum() * 0.25 ** 2
        got = LP.polygon_intersection_area(a, b)
        assert abs(got - est) < 0.02 * est + 2.0
        assert abs(LP.polygon_area(a) - inside(a, xx, yy).sum() * 0.25 ** 2) < 0.02 * LP.polygon_area(a) + 2.0
        assert abs(LP.polygon_intersection_area(a, b) - LP.polygon_intersection_area(b, a)) < 1e-9


def test_fill_poly_is_inclusive_and_clipped(LP):
    m = np.zeros((12, 12), np.uint8)
    LP.fill_poly(m, [[1, 1], [9, 1], [9, 9], [1, 9]], 1)
    assert m.sum() == 81 and m[1:10, 1:10].all()                       # both borders belong to the polygon
    t = np.zeros((12, 12), np.uint8)
    LP.fill_poly(t, [[1, 1], [10, 5], [1, 9]], 1)
    assert (t == t[::-1][np.r_[1:12, 0]]).all() or t[1:10].tolist() == t[1:10][::-1].tolist()       # symmetric about row 5
    assert t[5, 1:11].all() and t[1, 1] and t[9, 1] and not t[0].any() and not t[:, 11].any()
    big = np.zeros((20, 20), np.uint8)
    LP.fill_poly(big, [[-10, -10], [30, -10], [30, 30], [-10, 30]], 7)
    assert (big == 7).all()                                             # vertices far outside: clipped, no fault
    e = np.zeros((5, 5), np.uint8)
    LP.fill_poly(e, np.zeros((0, 2), np.int32), 1)
    assert e.sum() == 0


def test_fill_poly_against_point_in_polygon_on_random_convex_shapes(LP):
    rng = np.random.default_rng(5)
    for _ in range(8):
        ang = 2 * np.pi * (np.arange(7) + rng.uniform(-0.3, 0.3, 7)) / 7                # well separated: rounding keeps it convex
        pts = np.round(np.stack([40 + rng.uniform(20, 35) * np.cos(ang), 40 + rng.uniform(20, 35) * np.sin(ang)], 1)).astype(np.int32)
        m = np.zeros((80, 80), np.uint8)
        LP.fill_poly(m, pts, 1)
        yy, xx = np.mgrid[:80, :80]
        # signed distance to every edge line (positive inside for this orientation)
        d = np.full((80, 80), np.inf)
        for i in range(len(pts)):
            (x0, y0), (x1, y1) = pts[i], pts[(i + 1) % len(pts)]
            nrm = np.hypot(x1 - x0, y1 - y0)
            d = np.minimum(d, ((x1 - x0) * (yy - y0) - (y1 - y0) * (xx - x0)) / nrm)
        assert m[d > 0.75].all()                                        # clearly inside: filled
        assert not m[d < -0.75].any()                                   # clearly outside: untouched
        assert m[pts[:, 1], pts[:, 0]].all()                            # the vertices themselves


def test_resize_nearest_takes_floor_of_the_scaled_index(LP):
    a = np.arange(5, dtype=np.uint8)[None, :].repeat(2, 0)
    assert LP.resize_nearest(a, 3, 2)[0].tolist() == [0, 1, 3]          # floor(i * 5 / 3)
    assert LP.resize_nearest(a, 10, 4).shape == (4, 10) and LP.resize_nearest(a, 10, 4)[3].tolist() == [0, 0, 1, 1, 2, 2, 3, 3, 4, 4]


def test_c_abi_return_codes_of_the_primitives():
    """0 = ok, 1 = bad arguments, 2 = output capacity too small with the needed counts written back (include/rapiddoc_mi355.h)."""
    import ctypes as C
    lib = _lib.load()
    m = np.zeros((8, 8), np.uint8)
    m[1:3, 1:3] = 1
    m[5:7, 4:8] = 1
    n_c, n_p = C.c_int32(0), C.c_int32(0)
    pts, counts = np.zeros((2, 2), np.int32), np.zeros(1, np.int32)
    assert lib.rd_find_external_contours(m.ctypes.data, 8, 8, pts.ctypes.data, 2, counts.ctypes.data, 1, C.byref(n_c), C.byref(n_p)) == 2
    assert (n_c.value, n_p.value) == (2, 8)
    pts, counts = np.zeros((8, 2), np.int32), np.zeros(2, np.int32)
    assert lib.rd_find_external_contours(m.ctypes.data, 8, 8, pts.ctypes.data, 8, counts.ctypes.data, 2, C.byref(n_c), C.byref(n_p)) == 0
    assert counts.tolist() == [4, 4] and pts[:4].tolist() == [[4, 5], [4, 6], [7, 6], [7, 5]]
    assert lib.rd_find_external_contours(None, 8, 8, pts.ctypes.data, 8, counts.ctypes.data, 2, C.byref(n_c), C.byref(n_p)) == 1
    assert lib.rd_find_external_contours(m.ctypes.data, 0, 8, pts.ctypes.data, 8, counts.ctypes.data, 2, C.byref(n_c), C.byref(n_p)) == 1
    out, n = np.zeros((4, 2), np.int32), C.c_int32(7)
    assert lib.rd_approx_poly_dp(None, 0, 1.0, 1, out.ctypes.data, C.byref(n)) == 0 and n.value == 0          # an empty curve is not an error
    assert lib.rd_approx_poly_dp(None, 3, 1.0, 1, out.ctypes.data, C.byref(n)) == 1
    assert lib.rd_approx_poly_dp(pts.ctypes.data, 4, -1.0, 1, out.ctypes.data, C.byref(n)) == 1
    assert lib.rd_min_area_rect_points(None, 4, out.ctypes.data) == 1
    assert lib.rd_fill_poly(None, 8, 8, pts.ctypes.data, 4, 1) == 1
    assert lib.rd_polygon_area(None, 4) == 0.0 and lib.rd_contour_area(None, 4) == 0.0 and lib.rd_arc_length(pts.ctypes.data, 1, 1) == 0.0
    sel, src, k = np.zeros((4, 6), np.float32), np.zeros(4, np.int32), C.c_int32(0)
    assert lib.rd_layout_postprocess_select(sel.ctypes.data, 4, 5, 100, 100, None, sel.ctypes.data, src.ctypes.data, C.byref(k)) == 1
