"""GPU: the three RT-DETR-family head operators of csrc/kernels_rtdetr.hip / api.cpp (preparation for the layout neck /
decoder, whose graph is ONNX-only and absent offline - PARITY UNPINNED, wired into nothing, not part of the pages/s line) against float64
restatements of their published definitions:

  * rd_msdeform_attn   multi-scale deformable attention = sum over levels / points of attention weight x bilinear sample
                        (F.grid_sample(align_corners=False, padding_mode="zeros") on every level, Deformable DETR's reference form);
  * rd_topk_rows       torch.topk values, plus the tie rule the kernel defines (equal values in ascending index order) and NaN above +inf;
  * rd_encoder_layer   post-norm transformer encoder layer with q = k = x + pos, v = x (AIFI), written out in float64.
"""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _lib():
    from rapiddoc_amd import _lib
    return _lib.load()


def _msdeform_ref(value, shapes, loc, attn):
    """value [B,S,H,D], shapes [(h,w)], loc [B,Q,H,L,P,2] in [0,1], attn [B,Q,H,L,P] -> [B,Q,H*D] in float64."""
    B, S, H, D = value.shape
    Q, L, P = loc.shape[1], loc.shape[3], loc.shape[4]
    out = torch.zeros((B, H, D, Q), dtype=torch.float64, device=value.device)
    start = 0
    for l, (h, w) in enumerate(shapes):
        v = value[:, start:start + h * w].double().permute(0, 2, 3, 1).reshape(B * H, D, h, w)
        grid = (2 * loc[:, :, :, l].double() - 1).permute(0, 2, 1, 3, 4).reshape(B * H, Q, P, 2)
        s = torch.nn.functional.grid_sample(v, grid, mode="bilinear", padding_mode="zeros", align_corners=False)     # [B*H, D, Q, P]
        a = attn[:, :, :, l].double().permute(0, 2, 1, 3).reshape(B * H, 1, Q, P)
        out += (s * a).sum(-1).reshape(B, H, D, Q)
        start += h * w
    return out.permute(0, 3, 1, 2).reshape(B, Q, H * D)


@pytest.mark.parametrize("B,H,D,Q,P,shapes", [
    (2, 8, 32, 300, 4, [(80, 80), (40, 40), (20, 20)]),        # the shape class of a 640-pixel RT-DETR decoder
    (1, 8, 32, 37, 4, [(25, 25), (13, 13), (7, 7), (4, 4)]),   # four levels, odd sizes
    (3, 4, 16, 5, 2, [(3, 5), (1, 1)]),                        # tiny maps: most corners fall outside
])
def test_msdeform_attn_matches_fp64(B, H, D, Q, P, shapes):
    g = torch.Generator(device="cuda").manual_seed(B * 100 + Q)
    L = len(shapes)
    S = sum(h * w for h, w in shapes)
    value = torch.rand((B, S, H, D), device="cuda", generator=g) * 2 - 1
    loc = torch.rand((B, Q, H, L, P, 2), device="cuda", generator=g) * 1.3 - 0.15          # some samples outside [0, 1]: zero padding
    loc[0, 0, 0, 0, 0] = torch.tensor([0.0, 1.0], device="cuda")                            # exactly on the border
    attn = torch.softmax(torch.rand((B, Q, H, L * P), device="cuda", generator=g) * 3, -1).reshape(B, Q, H, L, P).contiguous()
    sh = torch.tensor(shapes, dtype=torch.int32, device="cuda")
    st = torch.tensor(np.concatenate([[0], np.cumsum([h * w for h, w in shapes])[:-1]]), dtype=torch.int32, device="cuda")
    out = torch.full((B, Q, H * D), 555.0, device="cuda")
    rc = _lib().rd_msdeform_attn(0, value.data_ptr(), sh.data_ptr(), st.data_ptr(), loc.data_ptr(), attn.data_ptr(), out.data_ptr(), B, S, H, D, Q,
                                 L, P, None)
    torch.cuda.synchronize()
    assert rc == 0
    ref = _msdeform_ref(value, shapes, loc, attn)
    assert float((out.double() - ref).abs().max()) < 2e-6


def _topk(scores, k):
    rows, n = scores.shape
    vals = torch.empty((rows, k), device="cuda")
    idx = torch.empty((rows, k), dtype=torch.int32, device="cuda")
    rc = _lib().rd_topk_rows(0, scores.data_ptr(), rows, n, k, vals.data_ptr(), idx.data_ptr(), None)
    torch.cuda.synchronize()
    assert rc == 0
    return vals, idx


@pytest.mark.parametrize("rows,n,k", [(4, 8400 * 25, 300), (2, 8400 * 11, 300), (3, 1000, 1000), (5, 77, 1), (1, 2048, 1024)])
def test_topk_matches_torch(rows, n, k):
    g = torch.Generator(device="cuda").manual_seed(n + k)
    scores = torch.sigmoid(torch.randn((rows, n), device="cuda", generator=g) * 3)
    vals, idx = _topk(scores, k)
    tv, _ = torch.topk(scores, k, dim=1)
    assert torch.equal(vals, tv)                                                     # the same values, in the same (descending) order
    assert torch.equal(torch.gather(scores, 1, idx.long()), vals)                    # and the indices point at them
    assert all(len(set(r.tolist())) == k for r in idx.cpu())                         # no index twice


def test_topk_tie_rule_and_specials():
    """Equal values come out in ascending index order; NaN sorts above +inf, -inf last; all-equal rows return indices 0 .. k-1."""
    s = torch.zeros((3, 5000), device="cuda")
    s[0, [4000, 17, 2500]] = 1.0                                  # three equal maxima, then zeros: 17, 2500, 4000, 0, 1, 2 ...
    s[1] = 0.25                                                    # all equal
    s[2, 10] = float("nan"); s[2, 20] = float("inf"); s[2, 30] = float("-inf"); s[2, 40] = 3.0
    vals, idx = _topk(s, 8)
    assert idx[0].tolist() == [17, 2500, 4000, 0, 1, 2, 3, 4]
    assert idx[1].tolist() == list(range(8))
    assert idx[2].tolist()[:3] == [10, 20, 40] and torch.isnan(vals[2, 0]) and float(vals[2, 1]) == float("inf")
    full, fidx = _topk(s[2:3, :64].contiguous(), 64)
    assert fidx[0, -1].item() == 30 and float(full[0, -1]) == float("-inf")


@pytest.mark.parametrize("B,T,Dm,heads,F,act", [(2, 625, 256, 8, 1024, 2), (1, 100, 128, 8, 256, 1), (3, 49, 64, 2, 128, 2)])
def test_encoder_layer_matches_fp64(B, T, Dm, heads, F, act):
    g = torch.Generator(device="cuda").manual_seed(T + Dm)
    r = lambda *s, sc=1.0: ((torch.rand(s, device="cuda", generator=g) * 2 - 1) * sc).contiguous()
    x, pos = r(B, T, Dm), r(B, T, Dm, sc=0.5)
    in_w, in_b = r(3 * Dm, Dm, sc=Dm ** -0.5), r(3 * Dm, sc=0.1)
    out_w, out_b = r(Dm, Dm, sc=Dm ** -0.5), r(Dm, sc=0.1)
    w1, b1, w2, b2 = r(F, Dm, sc=Dm ** -0.5), r(F, sc=0.1), r(Dm, F, sc=F ** -0.5), r(Dm, sc=0.1)
    g1, be1, g2, be2 = r(Dm) + 1.5, r(Dm, sc=0.2), r(Dm) + 1.5, r(Dm, sc=0.2)
    lib = _lib()
    nbytes = lib.rd_encoder_layer_workspace(B * T, Dm, F)
    ws = torch.empty(nbytes // 4 + 16, device="cuda")
    out = torch.empty((B, T, Dm), device="cuda")
    rc = lib.rd_encoder_layer(0, x.data_ptr(), pos.data_ptr(), B, T, Dm, heads, F, act, in_w.data_ptr(), in_b.data_ptr(), out_w.data_ptr(),
                              out_b.data_ptr(), g1.data_ptr(), be1.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(),
                              g2.data_ptr(), be2.data_ptr(), 1e-5, out.data_ptr(), ws.data_ptr(), nbytes, None)
    torch.cuda.synchronize()
    assert rc == 0
    d = lambda t: t.double()
    hd = Dm // heads
    qk = d(x) + d(pos)
    q = (qk @ d(in_w[:Dm]).T + d(in_b[:Dm])).reshape(B, T, heads, hd).transpose(1, 2)
    k = (qk @ d(in_w[Dm:2 * Dm]).T + d(in_b[Dm:2 * Dm])).reshape(B, T, heads, hd).transpose(1, 2)
    v = (d(x) @ d(in_w[2 * Dm:]).T + d(in_b[2 * Dm:])).reshape(B, T, heads, hd).transpose(1, 2)
    a = torch.softmax(q @ k.transpose(-1, -2) / hd ** 0.5, -1) @ v
    a = a.transpose(1, 2).reshape(B, T, Dm)
    y1 = torch.nn.functional.layer_norm(d(x) + a @ d(out_w).T + d(out_b), (Dm,), d(g1), d(be1), 1e-5)
    h = y1 @ d(w1).T + d(b1)
    h = torch.nn.functional.gelu(h) if act == 2 else torch.relu(h)
    ref = torch.nn.functional.layer_norm(y1 + h @ d(w2).T + d(b2), (Dm,), d(g2), d(be2), 1e-5)
    err = float((out.double() - ref).abs().max())
    assert err < 2e-5, err                 # fp32 MFMA products + fp32 softmax / LayerNorm, output of order 1-3
