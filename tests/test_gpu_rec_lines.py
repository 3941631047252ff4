"""GPU: the recogniser over text lines of DIFFERENT reference padded widths in one launch (rd_rec_backbone_forward_lines + the ragged
tail + rd_ctc_collapse_lines) - the strict rec mode of PagePipeline.  The reference pads a line to the width of its own chunk of six
(rapid_ocr.py:404-449) and the network's output depends on that width, so "the reference's result at GPU launch sizes" means: line b of a
launch == the network run on x[b:b+1, :, :, :w_b] alone.  Checked against the oracle (oracle/nets.py, pinned to the reference's BaseModel)."""
import numpy as np
import pytest
import torch

from oracle import nets as O
from rapiddoc_amd import weights as W

pytestmark = pytest.mark.gpu
TOL = 1e-3


def _state(golden_dir, kind="ppocrv6_rec"):
    return W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{kind}.json"), 0)


def _lines_input(widths, W_launch, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros((len(widths), 3, 48, W_launch))
    for b, w in enumerate(widths):
        x[b, :, :, :w] = torch.rand((3, 48, w), generator=g) * 2 - 1
    return x


@pytest.mark.parametrize("precision", ["auto", "fp32"])
@pytest.mark.parametrize("widths,W_launch", [([320, 320, 323, 401, 517, 640, 638, 77, 16], 672), ([321], 352), ([1000, 1003, 1056], 1056)])
def test_backbone_lines_equal_the_per_line_forward_and_the_oracle(golden_dir, precision, widths, W_launch):
    from rapiddoc_amd.engine import RdEngine, rec_line_table
    st_np = _state(golden_dir)
    eng = RdEngine("ppocrv6_rec").load_weights(st_np)
    eng.set_precision(precision)
    st = O.as_torch_state(st_np)
    x = _lines_input(widths, W_launch, seed=len(widths))
    w = np.asarray(widths)
    T = (((w - 1) // 2 + 1 - 1) // 2 + 1) // 2
    first = np.cumsum(T) - T + 3                               # (+3: offsets are honoured, not assumed to start at 0)
    tab = torch.from_numpy(rec_line_table(w, first)).cuda()
    dim = eng.rec_token_dim
    tokens = torch.full((int(first[-1] + T[-1]) + 2, dim), 777.0, device="cuda")
    eng.rec_backbone_forward_lines(x.cuda(), tab, tokens)
    torch.cuda.synchronize()
    got = tokens.cpu()
    assert float((got[:3] - 777.0).abs().max()) == 0.0 and float((got[-2:] - 777.0).abs().max()) == 0.0     # nothing outside the lines' slots
    for b, wb in enumerate(widths):
        xb = x[b:b + 1, :, :, :wb].contiguous()
        with torch.no_grad():
            ref = O.rec_forward(st, xb, return_all=True)["backbone"]          # [1, 384, 1, T]
        ref = ref[0, :, 0, :].t()
        assert ref.shape[0] == T[b] == eng._l.rd_rec_seq_len(int(wb))
        mine = got[first[b]: first[b] + T[b]]
        scale = max(1.0, float(ref.abs().max()))
        assert float((mine - ref).abs().max()) < TOL * scale, (b, wb)
        alone = eng.rec_backbone_forward(xb.cuda()).cpu()[0]                  # the same line as a launch of its own
        assert float((mine - alone).abs().max()) < 2e-4 * scale, (b, wb)
    assert not eng.range_overflow()


def test_collapse_lines_equals_the_host_decode(golden_dir):
    """rd_ctc_collapse_lines over ragged lines == ocr_host.ctc_decode line by line (strings, confidences bit for bit) and its kept
    columns are the time steps of the kept characters."""
    import ctypes as C
    from rapiddoc_amd import _lib, ocr_host
    lib = _lib.load()
    rng = np.random.default_rng(3)
    chars = ["blank"] + [chr(0x4E00 + i) for i in range(200)] + [" "]
    tab, max_len = ocr_host.char_table(chars)
    lens = [40, 1, 133, 7, 300]
    seg = np.zeros((len(lens), 2), np.int32)
    seg[:, 1] = lens
    seg[:, 0] = np.cumsum(lens) - lens + 5
    n_tok = int(seg[-1].sum())
    idx = rng.integers(0, 6, n_tok).astype(np.int32) * rng.integers(0, 2, n_tok).astype(np.int32) * 37 % len(chars)
    prob = rng.uniform(0.1, 1.0, n_tok).astype(np.float32)
    Tmax = 304
    row_bytes = (16 + Tmax * max_len + 15) // 16 * 16
    d = lambda a: torch.from_numpy(a).cuda()
    idx_d, prob_d, seg_d, tab_d = d(idx), d(prob), d(seg.reshape(-1)), d(tab)
    rows = torch.zeros((len(lens), row_bytes), dtype=torch.uint8, device="cuda")
    cols = torch.full((len(lens), Tmax), -1, dtype=torch.int16, device="cuda")
    confs = torch.full((len(lens), Tmax), -1.0, dtype=torch.float32, device="cuda")
    rc = lib.rd_ctc_collapse_lines(0, idx_d.data_ptr(), prob_d.data_ptr(), len(lens), seg_d.data_ptr(), Tmax, tab_d.data_ptr(), max_len, len(chars),
                                   rows.data_ptr(), row_bytes, cols.data_ptr(), confs.data_ptr(), None)
    assert rc == 0
    torch.cuda.synchronize()
    dec = ocr_host.parse_ctc_rows(rows.cpu().numpy())
    n_kept = rows.cpu().numpy()[:, 8:12].copy().view("<i4")[:, 0]
    for b, (f, t) in enumerate(seg.tolist()):
        want = ocr_host.ctc_decode(idx[None, f:f + t], prob[None, f:f + t], chars)[0]
        assert dec[b][0] == want[0] and np.float32(dec[b][1]) == np.float32(want[1])
        row = idx[f:f + t]
        keep = [i for i in range(t) if row[i] != 0 and (i == 0 or row[i] != row[i - 1])]
        assert n_kept[b] == len(keep) and cols.cpu().numpy()[b, :len(keep)].tolist() == keep
        assert np.array_equal(confs.cpu().numpy()[b, :len(keep)], prob[f:f + t][keep])


def _reference_rec_chunks(crop_hw, rec_batch_num=6):
    """rapid_ocr.py:404-449 restated independently of rapiddoc_amd: [(indices, imgW)] for crops of the given (h, w)."""
    width_list = [w / float(h) for h, w in crop_hw]
    indices = np.argsort(np.array(width_list))
    out = []
    for beg in range(0, len(crop_hw), rec_batch_num):
        end = min(len(crop_hw), beg + rec_batch_num)
        max_wh_ratio = 320 / 48
        for ino in range(beg, end):
            h, w = crop_hw[indices[ino]]
            max_wh_ratio = max(max_wh_ratio, w * 1.0 / h)
        out.append(([int(indices[i]) for i in range(beg, end)], int(48 * max_wh_ratio)))
    return out


def check_lines_against_oracle(pipe, st_rec, flat_lines, batch_ids=None, per_batch=None):
    """Every kept rec batch of a strict-mode call: each line's (idx, prob) == the oracle run on that line's tensor cut at the line's
    reference width, and the strings / scores the call returned are the decode of exactly those.  Returns {pooled line: width}."""
    from rapiddoc_amd import ocr_host
    got_w = {}
    ids = range(len(pipe.last_rec_batches)) if batch_ids is None else batch_ids
    for bi in ids:
        chunk, x, line_w, idxs, probs = pipe.last_rec_batches[bi]
        sel = range(len(chunk)) if per_batch is None else sorted(set(np.linspace(0, len(chunk) - 1, per_batch).astype(int).tolist()))
        for j in sel:
            wj = int(line_w[j])
            assert wj <= x.shape[3] and (wj == x.shape[3] or float(x[j, :, :, wj:].abs().max()) == 0.0)
            with torch.no_grad():
                lg = O.rec_forward(st_rec, x[j:j + 1, :, :, :wj].cpu().contiguous())
            ridx, rprob = O.ctc_greedy_stats(lg)
            top2 = torch.topk(lg, 2, dim=2).values
            safe = ((top2[..., 0] - top2[..., 1]) > 1e-2).numpy()[0]
            idx, prob = idxs[j].cpu().numpy(), probs[j].cpu().numpy()
            assert idx.shape == (lg.shape[1],)
            assert (idx == ridx.numpy()[0])[safe].all()
            assert np.abs(prob - rprob.numpy()[0])[safe].max() < TOL
            t, s = ocr_host.ctc_decode(idx[None], prob[None], pipe.characters)[0]
            i = int(chunk[j])
            assert flat_lines[i][1] == t and flat_lines[i][2] == ocr_host.format_score(s)
        for j, i in enumerate(chunk.tolist()):
            got_w[int(i)] = int(line_w[j])
    return got_w


def test_strict_mode_gives_every_line_its_reference_width_in_gpu_sized_launches(golden_dir):
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = {k: _state(golden_dir, k) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, n_rec_streams=4)
    assert pipe.rec_mode == "strict"                          # the default
    pipe.keep_rec_inputs = True
    pages_np, boxes = synth_batch(7, 3)
    pages = torch.from_numpy(pages_np).cuda()
    maps = render_text_maps(boxes, pages_np.shape[1:3], pipe.det_preprocess(pages[:1])[1], pages.device)
    res = pipe.run_batch(pages, None, det_maps_override=maps)
    flat = [ln for r in res for ln in r.lines]
    n = len(flat)
    assert n == 135
    cw, ch, rot, keep = pipe.last_rec_crop_sizes
    crop_hw = [(int(cw[i]), int(ch[i])) if rot[i] else (int(ch[i]), int(cw[i])) for i in range(n)]   # (h, w) of the image rec sees
    expected = _reference_rec_chunks(crop_hw)
    assert len(expected) == 23
    want_w = {i: w for idxs, w in expected for i in idxs}
    got_w = check_lines_against_oracle(pipe, O.as_torch_state(states["ppocrv6_rec"]), flat)
    assert got_w == want_w                                    # every line sees exactly the padded width the reference gives it
    assert len(pipe.last_rec_batches) < 12                    # ... in launches of GPU size, not 23 chunks of six
    order = [int(i) for chunk, *_ in pipe.last_rec_batches for i in chunk.tolist()]
    assert order == [i for idxs, _w in expected for i in idxs]    # the launches are runs of the reference's sorted order
    # zero right-padding starts at min(imgW, ceil(48 * w / h)) (resize_norm_img)
    chunk, x, line_w, _i, _p = pipe.last_rec_batches[-1]
    for j, i in enumerate(chunk.tolist()):
        h, w = crop_hw[i]
        rw = min(int(line_w[j]), int(np.ceil(48 * (w / float(h)))))
        assert rw == x.shape[3] or float(x[j, :, :, rw:].abs().max()) == 0.0
    # the same strings and scores as the one-launch-per-width form of the strict mode (RD_REC_TWO_STAGE=0 path)
    pipe.rec_two_stage = False
    pipe.keep_rec_inputs = False
    res1 = pipe.run_batch(pages, None, det_maps_override=maps)
    for a, b in zip(res, res1):
        assert [t for _q, t, _s in a.lines] == [t for _q, t, _s in b.lines]
        # scores are printed to three places (analyze_utils.py:280): the two launch forms may sit on either side of a rounding boundary,
        # i.e. differ by ONE unit of the third place (0.001 up to its own float rounding), never more
        assert max(abs(sa - sb) for (_q, _t, sa), (_q2, _t2, sb) in zip(a.lines, b.lines)) <= 1e-3 + 1e-9


def test_strict_mode_falls_back_to_fp32_with_the_line_table(golden_dir):
    """The split-fp16 range guard under the line table: an engine forced to fp32 (the separate stem kernels + mask_cols) returns the
    same strings."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = {k: _state(golden_dir, k) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, n_rec_streams=2)
    pages_np, boxes = synth_batch(3, 1)
    pages = torch.from_numpy(pages_np).cuda()
    maps = render_text_maps(boxes, pages_np.shape[1:3], pipe.det_preprocess(pages[:1])[1], pages.device)
    a = pipe.run_batch(pages, None, det_maps_override=maps)
    for e in pipe.rec_engines + [pipe.rec_tail]:
        e.set_precision("fp32")
    pipe.keep_rec_inputs = True
    b = pipe.run_batch(pages, None, det_maps_override=maps)
    flat = [ln for r in b for ln in r.lines]
    check_lines_against_oracle(pipe, O.as_torch_state(states["ppocrv6_rec"]), flat, per_batch=4)
    assert [t for _q, t, _s in a[0].lines] == [t for _q, t, _s in b[0].lines]


def test_page_uploader_streams_host_batches_and_run_batch_takes_host_pages(golden_dir):
    """The reference hands host arrays to every batch (batch_analyze.py:108-111): `run_batch` on numpy pages == on the same pages already in
    HBM; `PageUploader` (double-buffered, own copy stream) over a stream of four batches - pinned and pageable sources, buffers recycled -
    gives every batch its own pages and the same results."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, PageUploader, render_text_maps
    states = {k: _state(golden_dir, k) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, n_rec_streams=2)
    sets = [synth_batch(20 + k, 2) for k in range(4)]
    det_hw = pipe.det_preprocess(torch.from_numpy(sets[0][0]).cuda()[:1])[1]
    maps = [render_text_maps(b, p.shape[1:3], det_hw, torch.device("cuda", 0)) for p, b in sets]
    strings = lambda res: [[(t, s) for _q, t, s in r.lines] for r in res]
    want = [strings(pipe.run_batch(torch.from_numpy(p).cuda(), None, det_maps_override=m)) for (p, _b), m in zip(sets, maps)]
    assert all(len(page) == 45 for w in want for page in w) and want[0] != want[1]
    assert strings(pipe.run_batch(sets[1][0], None, det_maps_override=maps[1])) == want[1]                       # numpy pages
    assert strings(pipe.run_batch(torch.from_numpy(sets[2][0]), None, det_maps_override=maps[2])) == want[2]     # CPU tensor
    up = PageUploader(0, n_buffers=2)
    host = []
    for k, (p, _b) in enumerate(sets):
        if k % 2 == 0:
            t = PageUploader.pinned_like(p.shape)
            t.copy_(torch.from_numpy(p))
            host.append(t)
        else:
            host.append(p)                                    # pageable numpy: staged through the uploader's pinned buffer
    got = []
    nxt = up.submit(host[0])
    for k in range(4):
        cur, nxt = nxt, (up.submit(host[k + 1]) if k + 1 < 4 else None)
        dev = up.wait(cur)
        assert dev.is_cuda and tuple(dev.shape) == sets[k][0].shape
        res = pipe.run_batch(dev, None, det_maps_override=maps[k])
        up.release(cur)
        got.append(strings(res))
    torch.cuda.synchronize()
    assert got == want and up.bytes_uploaded == sum(p.nbytes for p, _b in sets)


def test_pipeline_pool_equals_single_pipeline_with_device_db_postprocess(golden_dir):
    """PagePipelinePool(workers = 2): two pipelines on two host threads and streams, each with its OWN DB post-process workspaces
    (a process-wide cache handed both threads the same union-find arrays).  Boxes come from the device post-process
    (quads_per_page=None); both shards have the same (B, H, W) so they used to share a cache key.  Repeated: a race is not every run."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, PagePipelinePool, render_text_maps
    states = {k: _state(golden_dir, k) for k in ("ppocrv6_det", "ppocrv6_rec")}
    kw = dict(rec_mode="throughput", rec_batch_num=1, n_rec_streams=2)      # one line per rec batch: a line's result does not depend on its shard
    single = PagePipeline(states, **kw)
    pool = PagePipelinePool(states, workers=2, **kw)
    assert pool.pipes[0].db_ws is not pool.pipes[1].db_ws
    pages_np, boxes = synth_batch(31, 4)
    pages = torch.from_numpy(pages_np).cuda()
    maps = render_text_maps(boxes, pages_np.shape[1:3], single.det_preprocess(pages[:1])[1], pages.device)
    ref = single.run_batch(pages, None, det_maps_override=maps)
    want = [[(q.round(2).tolist(), t, s) for q, t, s in r.lines] for r in ref]
    assert [len(w) for w in want] == [45] * 4
    for _ in range(6):
        got = [[(q.round(2).tolist(), t, s) for q, t, s in r.lines] for r in pool.run_batch(pages, None, det_maps_override=maps)]
        assert got == want


def test_bench_step_in_the_default_strict_mode_matches_the_oracle(golden_dir):
    """One whole benchmark step as bench.py runs it by default (32 pages, 1440 lines, 8 rec streams, strict rec mode): every launch keeps the
    reference's chunk widths (all 240 chunks of six), sampled lines of every launch against the oracle at their own width, the strings the
    step returned are the decode of those, and two steps give the same result (what bench.py's result_crc32 hashes)."""
    from rapiddoc_amd.pages import synth_pages
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    states = {k: _state(golden_dir, k) for k in ("ppocrv6_det", "ppocrv6_rec", "pphgnetv2_b4")}
    pipe = PagePipeline(states, rec_width_multiple=32, n_rec_streams=8)
    assert pipe.rec_mode == "strict"
    pages_np, boxes = synth_pages(list(range(32)))
    pages = torch.from_numpy(pages_np).cuda()
    maps = render_text_maps(boxes, pages_np.shape[1:3], pipe.det_preprocess(pages[:1])[1], pages.device)
    pipe.keep_rec_inputs = True
    res = pipe.run_batch(pages, None, det_maps_override=maps)
    assert [len(r.lines) for r in res] == [45] * 32
    flat = [ln for r in res for ln in r.lines]
    cw, ch, rot, keep = pipe.last_rec_crop_sizes
    crop_hw = [(int(cw[i]), int(ch[i])) if rot[i] else (int(ch[i]), int(cw[i])) for i in range(len(flat))]
    want_w = {i: w for idxs, w in _reference_rec_chunks(crop_hw) for i in idxs}
    nb = len(pipe.last_rec_batches)
    assert 8 <= nb <= 24 and sum(len(c) for c, *_ in pipe.last_rec_batches) == 1440
    got_w = check_lines_against_oracle(pipe, O.as_torch_state(states["ppocrv6_rec"]), flat, per_batch=3)
    assert got_w == want_w
    pipe.keep_rec_inputs = False
    pipe.last_rec_batches = []
    again = pipe.run_batch(pages, None, det_maps_override=maps)
    assert [[(t, s) for _q, t, s in r.lines] for r in again] == [[(t, s) for _q, t, s in r.lines] for r in res]


def test_prefetching_the_next_batch_changes_no_result(golden_dir):
    """`run_batch(..., prefetch=next_pages)`: the next batch's det + layout forwards run under this batch's recognition and the next call
    picks them up.  Over a stream of five batches (uploader buffers recycled, boxes from the det network's OWN maps for two of them, the
    layout features kept) every page's lines and features equal the unprefetched run's; a call on a different tensor than the one that
    was announced computes its own front; a pool of two shards prefetches per shard."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, PagePipelinePool, PageUploader, render_text_maps
    states = {k: _state(golden_dir, k) for k in ("ppocrv6_det", "ppocrv6_rec", "pphgnetv2_b4")}
    pipe = PagePipeline(states, n_rec_streams=2, keep_feats=True)
    sets = [synth_batch(50 + k, 2) for k in range(5)]
    dev_sets = [torch.from_numpy(p).cuda() for p, _b in sets]
    det_hw = pipe.det_preprocess(dev_sets[0][:1])[1]
    maps = [render_text_maps(b, p.shape[1:3], det_hw, torch.device("cuda", 0)) if k % 2 == 0 else None for k, (p, b) in enumerate(sets)]

    def digest(res):
        torch.cuda.synchronize()
        return [([(q.round(3).tolist(), t, s) for q, t, s in r.lines], [f.clone() for f in r.layout_feats]) for r in res]

    def same(a, b):
        return len(a) == len(b) and all(la == lb and len(fa) == len(fb) and all(torch.equal(x, y) for x, y in zip(fa, fb))
                                        for (la, fa), (lb, fb) in zip(a, b))
    want = [digest(pipe.run_batch(d, None, det_maps_override=m)) for d, m in zip(dev_sets, maps)]
    assert pipe.stats["front_prefetched"] == 0.0 and any(len(page[0]) for page in want[0])
    up = PageUploader(0, n_buffers=2)
    nxt = up.submit(sets[0][0])
    for k in range(5):
        cur, nxt = nxt, (up.submit(sets[k + 1][0]) if k + 1 < 5 else None)
        got = digest(pipe.run_batch(up.wait(cur), None, det_maps_override=maps[k], prefetch=up.wait(nxt) if nxt is not None else None))
        up.release(cur)
        assert pipe.stats["front_prefetched"] == (1.0 if k > 0 else 0.0)
        assert same(got, want[k]), k
    assert pipe._prefetched is None
    # announced one batch, then asked for another: the stale front is dropped
    pipe.run_batch(dev_sets[0], None, det_maps_override=maps[0], prefetch=dev_sets[1])
    assert pipe._prefetched is not None
    assert same(digest(pipe.run_batch(dev_sets[3], None, det_maps_override=maps[3])), want[3]) and pipe.stats["front_prefetched"] == 0.0
    # pool of two shards (one page each)
    pool = PagePipelinePool(states, workers=2, n_rec_streams=2, keep_feats=True)
    ref = [digest(pool.run_batch(d, None, det_maps_override=m)) for d, m in zip(dev_sets[:3], maps[:3])]
    for k in range(3):
        got = digest(pool.run_batch(dev_sets[k], None, det_maps_override=maps[k], prefetch=dev_sets[k + 1] if k + 1 < 3 else None))
        assert same(got, ref[k]), k
        assert all(p.stats["front_prefetched"] == (1.0 if k > 0 else 0.0) for p in pool.pipes)


@pytest.mark.parametrize("precision", ["fp32", "auto"])
def test_region_ocr_of_a_page_shard_reads_what_the_whole_batch_reads(golden_dir, monkeypatch, precision):
    """analyze.RegionOcr under page sharding: with `rec_width_sync` set and the pages' GLOBAL positions handed over as `page_keys`, the
    shard's lines are recognised at the widths the reference gives them inside the pooled list of the WHOLE batch
    (analyze_utils.py:216-252 -> rapid_ocr.py:404-449).  The exchange is played by a two-round stand-in (round one records every
    shard's (key, ratio) contribution and answers with the LOCAL rule, round two answers from the union: what dist.GlobalLineWidths
    computes from its all-gather, covered by tests/test_rec_width_sync_gloo.py).  Checked: every line's padded width equals the
    unsharded call's (exact; the local rule of round one does NOT give them - the control); the strings and scores are the unsharded
    call's - bit for bit in BOTH precision modes (round 6: a layer's kernel, product order and accumulator scheme follow from the layer,
    never from the launch size - tests/test_gpu_launch_invariance.py; rounds 1-5 picked the one-accumulator GEMM from 2048 rows on and
    the default mode's strings agreed for 70 % of these random-weight lines only)."""
    from rapiddoc_amd import ocr_host
    from rapiddoc_amd.analyze import RegionOcr
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline, render_text_maps
    from rapiddoc_amd import weights as W
    monkeypatch.setenv("RD_PRECISION", precision)
    states = {k: W.synth_state_dict(W.load_manifest(golden_dir / f"manifest_{k}.json"), 0) for k in ("ppocrv6_det", "ppocrv6_rec")}
    pipe = PagePipeline(states, rec_mode="strict", n_rec_streams=2)
    pipe.keep_rec_inputs = True
    pages_np, boxes = synth_batch(3, 3)
    pages = torch.from_numpy(pages_np).cuda()

    def region(x0, y0, x1, y1, order):
        return {"category_id": 1, "original_label": "text", "original_order": order,
                "poly": [x0, y0, x1, y0, x1, y1, x0, y1], "score": 0.9}
    dets = [[region(60, 50, 1140, 985, 0), region(60, 990, 645, 1420, 1)] for _ in range(3)]

    def maps_fn_for(first_page):
        def maps_fn(regs, ghw, dhw):
            per_img = []
            for p, r, useful in regs:
                px, py, x0, y0 = useful[0], useful[1], useful[2], useful[3]
                bb = np.asarray(boxes[first_page + p], dtype=np.float64).reshape(-1, 4)
                x1, y1 = r["poly"][4], r["poly"][5]
                inside = bb[(bb[:, 0] >= x0) & (bb[:, 2] <= x1) & (bb[:, 1] >= y0) & (bb[:, 3] <= y1)]
                per_img.append(inside - [x0 - px, y0 - py, x0 - px, y0 - py])
            return render_text_maps(per_img, ghw, dhw, pages.device)
        return maps_fn
    ocr = RegionOcr(pipe)

    def run(a, b, **kw):
        """-> (spans per page, padded width of every line in pooled order)"""
        out = ocr(pages[a:b].contiguous(), [list(d) for d in dets[a:b]], det_maps_fn=maps_fn_for(a), **kw)
        widths = {}
        for chunk, _x, lw, _i, _p in pipe.last_rec_batches:
            for j, i in enumerate(chunk.tolist()):
                widths[int(i)] = int(lw[j])
        spans = [[(d["text"], d["score"], d["category_id"]) for d in page if d["category_id"] in (15, 16)] for page in out]
        return spans, [widths[i] for i in range(len(widths))]
    whole, whole_w = run(0, 3)
    assert all(len(s) > 5 for s in whole) and len(whole_w) == sum(len(s) for s in whole)
    shards = [(0, 2), (2, 3)]                          # rank 0: pages 0-1, rank 1: page 2

    class Exchange:
        def __init__(self):
            self.contrib, self.rank, self.answer = {}, 0, False

        def __call__(self, keys, ratios):
            if not self.answer:                        # round one: record, answer with the local rule
                self.contrib[self.rank] = (np.array(keys), np.array(ratios))
                return ocr_host.rec_reference_widths(list(ratios))
            ks = np.concatenate([self.contrib[r][0] for r in sorted(self.contrib)])
            rs = np.concatenate([self.contrib[r][1] for r in sorted(self.contrib)])
            rk = np.concatenate([np.full(len(self.contrib[r][0]), r) for r in sorted(self.contrib)])
            assert np.array_equal(self.contrib[self.rank][0], keys) and np.array_equal(self.contrib[self.rank][1], ratios)
            order = np.argsort(ks, kind="stable")      # pooled page by page (lines of one page live on one rank, in its order)
            w, r = ocr_host.rec_reference_widths(rs[order].tolist())
            mine = rk[order] == self.rank
            return w[mine], r[mine]
    ex = Exchange()
    pipe.rec_width_sync = ex
    with pytest.raises(ValueError, match="page_keys"):
        ocr(pages[:1], [list(dets[0])], det_maps_fn=maps_fn_for(0))
    local, local_w = {}, {}
    for ex.answer in (False, True):
        for ex.rank, (a, b) in enumerate(shards):
            local[ex.rank], local_w[ex.rank, ex.answer] = run(a, b, page_keys=list(range(a, b)))
    assert sorted(set(ex.contrib[0][0].tolist())) == [0, 1] and sorted(set(ex.contrib[1][0].tolist())) == [2]      # GLOBAL page keys
    assert local_w[0, True] + local_w[1, True] == whole_w                   # the whole batch's widths, line by line
    assert local_w[0, False] + local_w[1, False] != whole_w                 # control: the shard's own pool gives other widths
    got = [ln for page in local[0] + local[1] for ln in page]
    want = [ln for page in whole for ln in page]
    assert len(got) == len(want) and [g[2] for g in got] == [w[2] for w in want]
    assert got == want, (precision, sum(g == w for g, w in zip(got, want)), len(want))
    # a rank without a single region still takes part in the exchange (its peers' collective would not pair up otherwise)
    n_before = len(ex.contrib)
    ex.rank, ex.answer = 7, False
    assert ocr(pages[:1], [[]], page_keys=[9]) == [[]]
    assert len(ex.contrib) == n_before + 1 and len(ex.contrib[7][0]) == 0
    pipe.rec_width_sync = None


def test_det_range_trip_under_the_width_exchange_makes_one_collective_call(golden_dir, monkeypatch):
    """with `rec_width_sync` set, a det engine that leaves the fp16 range is caught BEFORE the recogniser stage - the
    det forward is repeated on fp32 MFMA, the (key, ratio) list this rank hands to the exchange comes from the repeated maps, and the
    exchange is called exactly ONCE for the batch (a second call would pair with the peers' next collective)."""
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipeline
    from rapiddoc_amd import ocr_host
    monkeypatch.setenv("RD_PRECISION", "auto")
    states = {k: _state(golden_dir, k) for k in ("ppocrv6_det", "ppocrv6_rec")}
    big = dict(states["ppocrv6_det"])
    stem = [k for k in big if k.endswith("weight") and big[k].ndim == 4 and big[k].shape[1] == 3][0]
    big[stem] = big[stem] * 3.0e5                          # every later det activation ~1e5 x larger: the split kernels flag it
    pipe = PagePipeline({"ppocrv6_det": big, "ppocrv6_rec": states["ppocrv6_rec"]}, rec_mode="strict", n_rec_streams=2)
    pages_np, boxes = synth_batch(5, 2)
    pages = torch.from_numpy(pages_np).cuda()
    quads = [np.asarray([[x0, y0, x1, y0, x1, y1, x0, y1] for x0, y0, x1, y1 in np.asarray(b).reshape(-1, 4)[:7]], np.float32).reshape(-1, 4, 2)
             for b in boxes]
    calls = []

    def exchange(keys, ratios):
        calls.append((np.array(keys), np.array(ratios)))
        return ocr_host.rec_reference_widths(list(ratios))
    pipe.rec_width_sync = exchange
    assert pipe.det.precision == "auto"
    res = pipe.run_batch(pages, quads, page_keys=[4, 9])
    assert pipe.det.precision == "fp32" and pipe.det.range_fallbacks == 1          # tripped, and settled before the recogniser ran
    assert len(calls) == 1 and sorted(set(calls[0][0].tolist())) == [4, 9]
    assert [len(r.lines) for r in res] == [7, 7]
    res2 = pipe.run_batch(pages, quads, page_keys=[4, 9])                           # the next batch: one more call, no new trip
    assert len(calls) == 2 and pipe.det.range_fallbacks == 1
    assert [[t for _q, t, _s in r.lines] for r in res] == [[t for _q, t, _s in r.lines] for r in res2]
    pipe.rec_width_sync = None
