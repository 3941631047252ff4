#!/usr/bin/env python3
"""Headline benchmark: pages/s of the page hot path (layout backbone + OCR det + OCR rec) on synthetic pages.

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" = one pass of the hot path over one batch of 32 synthetic 1684x1191 pages per GPU (BASELINE.json
configs[1]): PPHGNetV2-B4 backbone @800x800 (the PP-DocLayout-L/V2/V3 backbone; neck/decoder are ONNX-only and
absent), PP-OCRv6-small det @960x704, PP-OCRv6-small rec on the 45 text lines of every page, fused
CTC argmax, host CTC decode.  A document STREAM: --vary-pages different page sets (default 8) cycle through the steps, and every
step's pages start in pinned HOST memory and are uploaded inside the timed region (copy stream, batch i + 1 under batch i;
--resident-pages --vary-pages 1 = rounds 1-3: one set, resident in HBM).  Weights are seed-0 synthetic (no checkpoints exist
offline) - throughput is weight independent.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time
from collections import defaultdict
from pathlib import Path

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")      # before the HIP runtime starts: one hardware queue per stream (rapiddoc_amd/__init__.py)

import numpy as np
import torch

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PAGES_PER_GPU = 32
FP32_MFMA_PEAK_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, 256 CUs @ 2.4 GHz
F16_MFMA_PEAK_TFLOPS = 2516.6   # same guide: v_mfma_f32_32x32x16_f16 dense
HBM_PEAK_GBS = 8000.0


def _throughput_rule(args):
    if args.rec_chunking == "adaptive":
        return ("chunks of the aspect-sorted lines whose size follows their width so that the persistent kernels' tile counts fill whole "
                "rounds of the chip (ocr_host.rec_batches_adaptive / rd_rec_plan_chunks, 16-160 lines), width rounded up to x%d" % args.rec_width_multiple)
    return "chunks of %d aspect-sorted lines, width rounded up to x%d" % (args.rec_batch, args.rec_width_multiple)


def _strict_rule(n_lines):
    return ("the reference's batching result: one global np.argsort of all %d lines, chunks of 6, every line padded to int(48 * max ratio of "
            "ITS chunk) (rapid_ocr.py:404-449); launches are runs of that sorted list sized by rd_rec_plan_chunks, every line computed at its "
            "own padded width inside the launch tensor (rd_rec_backbone_forward_lines)" % n_lines)


def load_states():
    from rapiddoc_amd import weights as W
    g = ROOT / "tests" / "golden"
    return {k: W.synth_state_dict(W.load_manifest(g / f"manifest_{k}.json"), 0)
            for k in ("ppocrv6_det", "ppocrv6_rec", "pphgnetv2_b4")}


def cpu_baseline(states, max_threads):
    """The oracle (CPU restatement of the reference networks, oracle/nets.py) on the host cores, bounded sample (~25 s): one page worth of
    layout-backbone + det, and 6 of its 45 rec crops (scaled by 45/6) - at 16 / 32 / 64 / 128 / all threads with one page per call, and with
    four pages per call at all threads (the whole host, not a silent cap at 32).  `value` / `cores` = the best of them;
    every configuration is listed in `thread_scaling`."""
    from oracle import nets as O
    rng = np.random.default_rng(0)
    st = {k: O.as_torch_state(v) for k, v in states.items()}

    def page_time(threads, pages, reps):
        torch.set_num_threads(threads)
        xb = torch.from_numpy(rng.uniform(0, 1, (pages, 3, 800, 800)).astype(np.float32))
        xd = torch.from_numpy(rng.standard_normal((pages, 3, 960, 704)).astype(np.float32))
        xr = torch.from_numpy(rng.uniform(-1, 1, (6 * pages, 3, 48, 1088)).astype(np.float32))
        with torch.no_grad():
            def timed(fn, n):
                fn()
                t0 = time.perf_counter()
                for _ in range(n):
                    fn()
                return (time.perf_counter() - t0) / n
            t_b4 = timed(lambda: O.pphgnetv2_features(st["pphgnetv2_b4"], xb), reps[0])
            t_det = timed(lambda: O.det_forward(st["ppocrv6_det"], xd), reps[1])
            t_rec = timed(lambda: O.ctc_greedy_stats(O.rec_forward(st["ppocrv6_rec"], xr)), reps[2])
        return (t_b4 + t_det + t_rec * 45.0 / 6.0) / pages, (t_b4, t_det, t_rec)
    counts = sorted({t for t in (16, 32, 64, 128, max_threads) if 1 <= t <= max_threads})
    scaling, best = {}, None
    for t in counts:
        tp, parts = page_time(t, 1, (2, 4, 3))
        scaling["%d threads, 1 page per call" % t] = round(1.0 / tp, 4)
        if best is None or tp < best[0]:
            best = (tp, t, "1 page per call", parts)
    tp, parts = page_time(max_threads, 4, (1, 2, 1))
    scaling["%d threads, 4 pages per call" % max_threads] = round(1.0 / tp, 4)
    if tp < best[0]:
        best = (tp, max_threads, "4 pages per call", parts)
    torch.set_num_threads(min(32, max_threads))
    return {"value": round(1.0 / best[0], 4), "unit": "pages/s", "cores": best[1], "kind": "port", "host_cores": max_threads,
            "thread_scaling": scaling,
            "sample": "torch-CPU fp32 oracle (oracle/nets.py), best of the listed thread counts (%s): B4 backbone 3x800x800 (%.3fs) + det "
                      "3x960x704 (%.3fs) + rec 6x3x48x1088 (%.3fs) scaled x45/6, per call" % ((best[2],) + best[3])}


def measure_backbone(pipe, pages, steps, warmup, dist=None, backend="nccl"):
    """PP-DocLayout's PPHGNetV2-B4 backbone alone on this rank's pages (pre-process + forward): wall time of `steps` passes,
    then one profiled pass for TFLOP/s and the fraction of the peak of the arithmetic ACTUALLY ISSUED: every layer's FLOPs are
    priced at the dense fp16 MFMA peak / 3 when it ran on the split-fp16 kernels and at the fp32 MFMA peak otherwise."""
    eng = pipe.layout

    def step():
        feats = pipe.layout_forward(pages)
        if eng.check_range_and_fallback():
            feats = pipe.layout_forward(pages)
        return feats
    for _ in range(max(1, warmup)):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if dist:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    eng.set_profiling(True)
    eng.profile_log = []
    step()
    torch.cuda.synchronize()
    eng.set_profiling(False)
    ops = eng.profile_log
    split = [o for o in ops if o["cfg"].endswith("/h3") or "_h3" in o["kind"] or "_ws" in o["kind"]]
    dense = [o for o in ops if o["flops"] > 0 and o not in split]
    fl_split, fl_dense = sum(o["flops"] for o in split), sum(o["flops"] for o in dense)
    ms_all = sum(o["ms"] for o in ops)
    t_ideal = fl_split / (F16_MFMA_PEAK_TFLOPS / 3.0 * 1e12) + fl_dense / (FP32_MFMA_PEAK_TFLOPS * 1e12)
    P = pages.shape[0]
    return {"dt": dt, "pages": P, "gflop_per_page": round((fl_split + fl_dense) / P / 1e9, 3),
            "roofline": {"bound": "mfma", "kernel": "PPHGNetV2-B4 backbone (all layers)",
                         "achieved": round((fl_split + fl_dense) / (ms_all * 1e-3) / 1e12, 3), "unit": "TFLOP/s",
                         "peak": round((fl_split + fl_dense) / t_ideal / 1e12, 1),
                         "frac": round(t_ideal / (ms_all * 1e-3), 4), "traffic": None,
                         "split_fp16_flop_share": round(fl_split / (fl_split + fl_dense), 3), "kernel_ms": round(ms_all, 3),
                         "note": "peak = FLOP-weighted harmonic mix of 838.9 (split-fp16 layers) and 157.3 TFLOP/s (fp32-MFMA layers)"}}


def measure_s2_dropin(states, pipe, pages, text_maps, device):
    """The S2 drop-in seam as the UNCHANGED RapidDoc pipeline would drive it (VERDICT r4 missing #4): rapidocr's TextDetector /
    TextRecognizer call `session(np.ndarray) -> np.ndarray` (rapid_doc/model/ocr/torch.py:171-192) - det in batches of same-size images
    (rapid_ocr.py:474-528, max_batch_size 8), rec in the reference's chunks of six lines, each call returning the softmax tensor
    [6, T, 18710] to the host (rapid_ocr.py:443).  Timed: exactly those session calls on this step's tensors (H2D of every input, the
    forward, D2H of every output), through rapiddoc_amd.session.Mi355DetSession / Mi355RecSession.  NOT timed: the reference's host-side
    cv2 pre- / post-processing between the calls (absent here), which the unchanged pipeline would add on its CPU."""
    from rapiddoc_amd.session import Mi355DetSession, Mi355RecSession
    P = pages.shape[0]
    det_x = pipe.det_preprocess(pages)[0].cpu().numpy()                     # [P,3,960,704] float32, what DetPreProcess hands over
    det_batches = [np.ascontiguousarray(det_x[i:i + 8]) for i in range(0, P, 8)]
    keep = pipe.keep_rec_inputs
    pipe.keep_rec_inputs = True
    pipe.run_batch(pages, None, det_maps_override=text_maps)
    line_x, ratios = {}, {}
    cw, ch, rot, _k = pipe.last_rec_crop_sizes
    for chunk, x, lw, _i, _p in pipe.last_rec_batches:
        xh = x.cpu().numpy()
        for j, i in enumerate(chunk.tolist()):
            line_x[int(i)] = (xh[j], int(lw[j]))
    pipe.keep_rec_inputs = keep
    pipe.last_rec_batches = []
    n = len(line_x)
    for i in range(n):
        h, w = (int(cw[i]), int(ch[i])) if rot[i] else (int(ch[i]), int(cw[i]))
        ratios[i] = w / float(h)
    order = np.argsort(np.array([ratios[i] for i in range(n)]))
    rec_batches = []
    for beg in range(0, n, 6):
        idxs = [int(i) for i in order[beg:beg + 6]]
        img_w = line_x[idxs[0]][1]
        rec_batches.append(np.ascontiguousarray(np.stack([line_x[i][0][:, :, :img_w] for i in idxs])))
    det_s = Mi355DetSession(states["ppocrv6_det"], device)
    rec_s = Mi355RecSession(states["ppocrv6_rec"], device)
    assert rec_s.lazy_softmax            # the seam's default: the softmax tensor stays in HBM (session.LazySoftmax)

    def step():
        """-> bytes that crossed PCIe device -> host.  Every rec result is consumed the way CTCLabelDecode consumes it
        (`preds.argmax(axis=2)`, `preds.max(axis=2)`, rapid_ocr.py:443-449) before the next call."""
        out_bytes = 0
        for xb in det_batches:
            out_bytes += det_s(xb).nbytes
        for xb in rec_batches:
            preds = rec_s(xb)
            am, mx = preds.argmax(axis=2), preds.max(axis=2)
            out_bytes += preds.nbytes if not rec_s.lazy_softmax or preds.materialized else am.size * 8      # int32 index + float32 probability per time step
        return out_bytes

    def timed():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nbytes = step()
        torch.cuda.synchronize()
        return time.perf_counter() - t0, nbytes
    # warm-up: three passes (a shape's plan is built on its first call, captured as a hipGraph on its second, replayed from the third on;
    # the staging buffers grow to the widest chunk)
    for _ in range(3):
        step()
    rec_s.host_ms = {k_: 0 for k_ in rec_s.host_ms}
    plans0 = rec_s.engine.plan_stats()
    sec, out_bytes = timed()                    # default: lazy softmax
    materialized = rec_s.softmax_materialized
    plans1 = rec_s.engine.plan_stats()
    rec_host = {k_: round(v_, 2) for k_, v_ in rec_s.host_ms.items()}
    # the same calls with plain ndarrays out of the session (rounds 1-5): fresh arrays, then views of the pinned staging buffers
    rec_s.lazy_softmax = False
    step()
    sec_eager, out_bytes_eager = timed()
    det_s.copy_out = rec_s.copy_out = False
    sec_view, _ = timed()
    in_bytes = sum(b.nbytes for b in det_batches) + sum(b.nbytes for b in rec_batches)
    return {"pages_s": round(P / sec, 3), "ms_per_step": round(sec * 1e3, 3), "steps": 1, "warmup": "3 passes",
            "d2h_mb_per_step": round(out_bytes / 1e6, 1), "softmax_tensors_materialized": int(materialized),
            "rec_host_ms": rec_host, "rec_hipgraph": {k_: plans1[k_] - plans0[k_] for k_ in plans1},
            "eager_ndarray": {"pages_s": round(P / sec_eager, 3), "ms_per_step": round(sec_eager * 1e3, 3),
                              "d2h_mb_per_step": round(out_bytes_eager / 1e6, 1),
                              "pages_s_pinned_views": round(P / sec_view, 3), "ms_per_step_pinned_views": round(sec_view * 1e3, 3),
                              "what": "Mi355RecSession(lazy_softmax=False): every call copies softmax [6,T,18710] to the host (rounds 1-5)"},
            "det_session_calls": len(det_batches), "rec_session_calls": len(rec_batches), "lines": n,
            "h2d_mb_per_step": round(in_bytes / 1e6, 1),
            "what": "numpy -> session -> result exactly as rapid_ocr.py:443,528 call it (det batches of <= 8 pages, rec chunks of 6); the rec "
                    "session returns session.LazySoftmax - softmax [6,T,C] left in HBM, argmax(axis=2) / max(axis=2) (all CTCLabelDecode asks) "
                    "answered from the device's reductions of it, any other access materialises the exact ndarray; session calls only - the "
                    "reference's host cv2 pre / post-processing is not in this number"}


def measure_formula():
    """PP-FormulaNet_plus-M (BASELINE config 3's formula stage; not part of `value`): the PPHGNetV2-B6 encoder at B = 32 (98.945 GFLOP per
    384x384 formula) and the greedy MBart decode at B = 8 / 32 (64 new tokens on 144 encoder states), synthetic weights."""
    from rapiddoc_amd import weights as W
    from rapiddoc_amd.engine import RdEngine
    man = W.load_manifest(ROOT / "tests" / "golden" / "manifest_ppformulanet_plus_m_m8.json")
    man = [(n, (2562, 512) if n.endswith("embed_positions.weight") else s_, d) for n, s_, d in man]     # the full 2560-token position table
    st = W.synth_state_dict(man, 0)
    enc_eng = RdEngine("pphgnetv2_b6_formula").load_weights({k: v for k, v in st.items() if k.startswith("backbone.")})
    dec_eng = RdEngine("ppformulanet_head").load_weights({k: v for k, v in st.items() if k.startswith("head.")})
    out = {"what": "PP-FormulaNet_plus-M, synthetic weights; encoder = PPHGNetV2-B6 @384x384 (98.945 GFLOP / formula), decode = greedy MBart, "
                   "64 new tokens over 144 encoder states"}
    x = torch.rand((32, 1, 384, 384), device="cuda") * 2 - 1
    enc_eng.formula_encoder_forward(x)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        enc_eng.formula_encoder_forward(x)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    out["encoder_b32"] = {"ms": round(dt * 1e3, 3), "formulas_s": round(32 / dt, 1), "tflops": round(98.945e9 * 32 / dt / 1e12, 1)}
    for B in (8, 32):
        enc = torch.randn((B, 144, 2048), device="cuda") * 3
        dec_eng.formula_decode(enc, 8)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = dec_eng.formula_decode(enc, 64)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        steps = int(ids.shape[1]) - 1
        out["decode_b%d" % B] = {"ms_per_step": round(dt * 1e3 / steps, 4), "tokens_s": round(B * steps / dt, 0), "steps": steps}
    del enc_eng, dec_eng
    return out


def bench_backbone(args, pool, pages, rank, world, dist, backend):
    """--only backbone: the backbone measurement as the whole job (north_star's >= 40 % MFMA item)."""
    m = measure_backbone(pool.pipes[0], pages, args.steps, args.warmup, dist, backend)
    P = pages.shape[0]
    if rank == 0:
        rec = {"metric": "pages/sec (PP-DocLayout backbone only)", "value": round(P * world * args.steps / m["dt"], 3), "unit": "pages/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(m["dt"] / args.steps * 1e3, 3),
               "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "PPHGNetV2-B4 @800x800 (pre-process + backbone) on %d synthetic pages per GPU" % P,
                          "gflop_per_page": m["gflop_per_page"]},
               "roofline": m["roofline"]}
        print(json.dumps(rec), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


def pin_rank_to_cores(local_rank, local_world):
    """N ranks share one host: give each its own slice of the cores this process may run on (host stages - DB rectangles,
    crop descriptors, launch enqueue, string parsing - are ~15 ms of a ~90 ms step; unpinned ranks migrate across NUMA
    nodes and steal each other's caches).  Returns the cores the rank keeps (also what torch's intra-op pool is sized to)."""
    try:
        cores = sorted(os.sched_getaffinity(0))
        per = max(1, len(cores) // max(1, local_world))
        mine = cores[local_rank * per: (local_rank + 1) * per] or cores
        os.sched_setaffinity(0, mine)
        torch.set_num_threads(max(1, min(len(mine), 8)))
        return len(mine)
    except (AttributeError, OSError):
        return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: ~1.5 s of timed work.  One warm-up step is not enough on a box whose GPU has been idle: the first process measured
    # 125 ms/step over 3 steps after 1 warm-up against 102 in every later run (clocks / first-use code loading), with normal
    # per-kernel durations in the profiling pass that follows
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pages", type=int, default=PAGES_PER_GPU, help="pages per GPU and step (weak scaling)")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak: the global document has --pages x N pages; strong: it has --global-pages whatever N is.  Either way "
                         "rank r takes pages {i : i mod N = r} of ONE global list (rapiddoc_amd.dist.shard_pages)")
    ap.add_argument("--global-pages", type=int, default=256)
    ap.add_argument("--only", choices=("all", "backbone"), default="all",
                    help="backbone: time the PPHGNetV2-B4 layout backbone alone (north_star's >= 40 %% MFMA item)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dump-profile", type=str, default="")
    ap.add_argument("--rec-batch", type=int, default=64)
    ap.add_argument("--rec-chunking", choices=("fixed", "adaptive"), default="adaptive",
                    help="throughput mode: chunks of --rec-batch lines, or chunk sizes chosen per width so that the persistent kernels' "
                         "tile counts fill whole rounds of the chip (ocr_host.rec_batches_adaptive)")
    ap.add_argument("--rec-streams", type=int, default=8)
    ap.add_argument("--rec-width-multiple", type=int, default=32, help="padded rec batch width is rounded up to this (plan-cache granularity)")
    ap.add_argument("--inflight", type=int, default=1, help="page batches (steps) in flight per GPU: each runs a whole batch on its own "
                    "pipeline / host thread, so the GPU has the next batch's det + layout while this one decodes")
    ap.add_argument("--workers", type=int, default=1, help="page-batch shards in flight per GPU (host stages of one overlap GPU stages of the other)")
    ap.add_argument("--rec-mode", choices=("throughput", "strict"), default="strict",
                    help="rec batching of the TIMED steps: strict (default, the product's default) = the reference's own batching result "
                         "(one global argsort, chunks of 6, every line at the padded width int(48 * max ratio) of ITS chunk, "
                         "rapid_ocr.py:404-449) in GPU-sized launches (rd_rec_backbone_forward_lines); throughput = every line at its "
                         "launch's width.  The other mode is measured in a short post-pass and reported next to it")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not enqueue the next batch's det + layout forwards under this batch's recognition (A/B switch; default: "
                         "a two-stage software pipeline over the batch stream - inside the timed region the first step's front is "
                         "exposed and the last step runs nothing ahead, so exactly --steps batches of work are timed)")
    ap.add_argument("--vary-pages", type=int, default=8,
                    help="K different page sets, step i runs set i mod K: a document stream does not repeat a batch, so the steps keep "
                         "meeting new line widths / token counts (plan caches, hipGraph slots and the tail's bucketing are exercised, and "
                         "with K > --warmup the timed region builds the plans of the sets it has not seen: config.plan_cache_misses). "
                         "1 = re-run one set (what rounds 1-3 measured)")
    ap.add_argument("--setup-passes", type=int, default=3,
                    help="untimed passes over ALL page sets before the warm-up: plans built and hipGraphs captured for every launch shape "
                         "the stream contains, so that the timed steps are replays (0 = only what --warmup happens to cover, as in rounds 4-5)")
    ap.add_argument("--setup-steps", type=int, default=8,
                    help="untimed passes over page set 0 BEFORE the --warmup steps, part of the setup like loading the weights: synthesising the "
                         "page sets keeps the host busy and the GPU idle for ~20 s, and a GPU coming out of idle runs its first second of "
                         "work at lower clocks (measured: the first 9 steps of the first process on a fresh box 96-120 ms against 89). They touch "
                         "no page set the warm-up / timed steps have not seen, so the plan-cache misses of the stream stay in the timed region")
    ap.add_argument("--resident-pages", action="store_true",
                    help="keep the page sets in HBM and skip the host -> device upload of every step (rounds 1-3); default: the pages of "
                         "every step start in pinned HOST memory, like the arrays the reference hands to a batch (batch_analyze.py:108-111), "
                         "and are uploaded inside the timed region by PageUploader (copy stream, batch i + 1 under batch i)")
    ap.add_argument("--no-extra-passes", action="store_true", help="skip the post-passes (other rec mode, fp32 precision, backbone alone)")
    args = ap.parse_args()

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (no CPU fallback)")
    backend = os.environ.get("RD_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    # --gpus N is the job size.  One process per GPU: with the RCCL backend N ranks need N devices - fewer is an error, never a
    # silent N = 1 number.  RD_BENCH_BACKEND=gloo (ranks share the devices that exist) exists only to exercise the multi-process
    # code path on a single-GPU box.
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if backend == "nccl" and args.gpus > n_dev:
        raise SystemExit("bench.py: --gpus %d asked for, %d device(s) visible: one rank per GPU over RCCL needs %d devices"
                         % (args.gpus, n_dev, args.gpus))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # a bare `python bench.py --gpus N`: start the N ranks ourselves, exactly as the driver's torchrun line does
        import socket
        with socket.socket() as s_:
            s_.bind(("127.0.0.1", 0))
            port = s_.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
               "--master-addr", "127.0.0.1", "--master-port", str(port), str(Path(__file__).resolve())] + sys.argv[1:]
        sys.stdout.flush()
        os.execv(sys.executable, cmd)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    dev_index = local_rank % n_dev
    torch.cuda.set_device(dev_index)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    if world > 1:
        cores_per_rank = pin_rank_to_cores(local_rank, local_world)
    else:           # one rank: no pinning, it may run on every core the process is allowed
        try:
            cores_per_rank = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            cores_per_rank = os.cpu_count()
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist_mod.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist_mod.init_process_group(backend)
        dist = dist_mod
    # what the process group itself reports: its backend, its size, and the device every rank computes on (uuid all-gathered over the
    # group - over RCCL the eight must be distinct; the gloo test mode on a 1-GPU box shows one uuid N times)
    props = torch.cuda.get_device_properties(dev_index)
    my_dev = "%s/%s" % (getattr(props, "uuid", None) or "index%d" % dev_index, props.name)
    group_info = {"backend": None, "world_size": 1, "rccl_ranks_seen": 1, "devices": [my_dev], "distinct_devices": 1}
    if dist:
        seen = [None] * world
        dist.all_gather_object(seen, (rank, my_dev))
        assert sorted(r for r, _ in seen) == list(range(world)), seen
        devs = [d for _, d in sorted(seen)]
        group_info = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "rccl_ranks_seen": len(seen),
                      "devices": devs, "distinct_devices": len(set(devs))}
        if backend == "nccl" and len(set(devs)) != world:
            raise SystemExit("bench.py: %d ranks over RCCL share %d device(s): %s" % (world, len(set(devs)), devs))

    # rank 0 (re)builds the library if it is missing or stale; EVERY rank then passes the same barrier, so no rank can
    # dlopen a half-linked file or pair its first collective with rank 0's barrier (build() links to a temporary name
    # and renames it into place)
    from rapiddoc_amd import build as rd_build
    if rank == 0:
        rd_build.build(verbose=False)
    if dist:
        dist.barrier()
    from rapiddoc_amd.pages import synth_batch
    from rapiddoc_amd.pipeline import PagePipelinePool, PageUploader, boxes_to_quads, render_text_maps
    from rapiddoc_amd.dist import gather_page_results

    states = load_states()
    pool_kw = dict(device=dev_index, workers=args.workers, rec_batch_num=args.rec_batch, rec_width_multiple=args.rec_width_multiple,
                   n_rec_streams=max(1, args.rec_streams // max(1, args.workers)), rec_chunking=args.rec_chunking)
    pool = PagePipelinePool(states, rec_mode=args.rec_mode, **pool_kw)
    pipe = pool.pipes[0]
    extra_pools = [PagePipelinePool(states, rec_mode=args.rec_mode, **pool_kw) for _ in range(max(1, args.inflight) - 1)]
    pools = [pool] + extra_pools
    from rapiddoc_amd.dist import shard_pages
    from rapiddoc_amd.pages import synth_pages
    n_global = args.global_pages if args.scaling == "strong" else args.pages * world
    my_pages = shard_pages(n_global, rank, world)          # interleaved split of ONE document list
    P = len(my_pages)
    pages_np, boxes = synth_pages(my_pages)
    pages = torch.from_numpy(pages_np).cuda()
    K_sets = max(1, args.vary_pages)
    upload = not args.resident_pages
    # page sets (--vary-pages): the same shard positions of later "documents" (page ids offset by n_global * k).  With the upload in
    # the timed region they live in pinned host memory (a renderer would write them there), else in HBM.
    def place(arr):
        if not upload:
            return torch.from_numpy(arr).cuda()
        t = PageUploader.pinned_like(arr.shape)
        t.copy_(torch.from_numpy(arr))
        return t
    page_sets = [(place(pages_np), boxes)]
    for k in range(1, K_sets):
        pk, bk = synth_pages([i + n_global * k for i in my_pages])
        page_sets.append((place(pk), bk))
    uploader = PageUploader(dev_index, n_buffers=2) if upload else None
    if args.only == "backbone":
        return bench_backbone(args, pool, pages, rank, world, dist, backend)
    # random-weight det maps carry no text, so the DB post-process stage gets maps rendered from the generator's own
    # line boxes (the det network still runs every step); its boxes then drive cropping and recognition
    det_hw = pipe.det_forward(pages[:1])[1]
    text_maps = render_text_maps(boxes, pages_np.shape[1:3], det_hw, pages.device)
    set_maps = [text_maps] + [render_text_maps(b, pages_np.shape[1:3], det_hw, pages.device) for _p, b in page_sets[1:]]
    quads = None
    step_no = [0]
    prefetch_on = not args.no_prefetch and len(pools) == 1
    # strict rec batching over N ranks: the reference pools the lines of the WHOLE page batch before it sorts and chunks them, so every
    # rank takes its lines' padded widths from the global list (one more small all-gather per step; the strings do not depend on N)
    width_sync = None
    # (one pipeline per rank only: with --inflight > 1 every lane would issue the collective from its own host thread, in no fixed order)
    if (dist is not None and args.rec_mode == "strict" and args.workers == 1 and len(pools) == 1
            and os.environ.get("RD_BENCH_WIDTH_SYNC", "1") != "0"):
        from rapiddoc_amd.dist import GlobalLineWidths
        width_sync = GlobalLineWidths(dist)
        for pl in pools:
            pl.pipes[0].rec_width_sync = width_sync

    def compute(k=0, ticket=None, ahead=None):
        """One step on pool k.  `ticket`: (PageUploader ticket, set index) of pages already travelling; else the step's own set is taken
        (resident, or - pinned host pages - uploaded by run_batch on the spot).  `ahead`: the NEXT step's ticket (or, resident, its
        set index): its det + layout forwards are enqueued under this step's recognition (PagePipeline.run_batch `prefetch`)."""
        if ticket is None:
            si = step_no[0] % K_sets
            step_no[0] += 1
            pg = page_sets[si][0]
        else:
            tk, si = ticket
            pg = uploader.wait(tk)
        pf = None
        if ahead is not None and prefetch_on:
            pf = uploader.wait(ahead[0]) if upload else page_sets[ahead][0]
        res = pools[k].run_batch(pg, quads, det_maps_override=set_maps[si], prefetch=pf, page_keys=my_pages if width_sync else None)
        if ticket is not None:
            uploader.release(tk)
        return [(my_pages[i], [(t, s) for _, t, s in r.lines]) for i, r in enumerate(res)]

    def next_ticket():
        si = step_no[0] % K_sets
        step_no[0] += 1
        return uploader.submit(page_sets[si][0]), si

    def step():
        return gather_page_results(compute(0), dist)

    step_wall_ms = []

    def run_steps(n):
        """n steps; with --inflight > 1 they are dealt round-robin to `inflight` host threads (one pipeline each) and the
        result collectives are issued from this thread in step order."""
        if len(pools) == 1:
            out = None
            trace = os.environ.get("RD_BENCH_STEP_TIMES") == "1"      # developer: per-step host times on stderr
            nxt = next_ticket() if upload and n > 0 else None       # the first batch's upload is exposed, the others run under a batch
            for i in range(n):
                ts = time.perf_counter()
                if upload:
                    cur, nxt = nxt, (next_ticket() if i + 1 < n else None)
                    out = gather_page_results(compute(0, cur, nxt), dist)       # the last step has nothing to run ahead
                else:
                    out = gather_page_results(compute(0, None, (step_no[0] + 1) % K_sets if i + 1 < n else None), dist)
                if trace:
                    torch.cuda.synchronize()
                    print("step %.1f ms" % ((time.perf_counter() - ts) * 1e3), file=sys.stderr)
                step_wall_ms.append((time.perf_counter() - ts) * 1e3)      # host clock only: no synchronise is added for it
                if os.environ.get("RD_BENCH_STEP_LOG") == "1":           # developer: what each step built (no synchronise)
                    ps = [e.plan_stats() for q in pools for e in q.engines]
                    print("step %2d  %.1f ms  plans %d  captures %d  stats %s" % (i, step_wall_ms[-1], sum(p_["plans_built"] for p_ in ps),
                          sum(p_["graph_captures"] for p_ in ps), {k_: round(v_, 1) for k_, v_ in pool.stats.items() if k_.startswith("t_")}), file=sys.stderr)
            return out
        from concurrent.futures import ThreadPoolExecutor
        streams = [torch.cuda.Stream() for _ in pools]

        def work(i):
            torch.cuda.set_device(dev_index)
            with torch.cuda.stream(streams[i % len(pools)]):
                r = compute(i % len(pools))
                torch.cuda.current_stream().synchronize()
            return r
        # one thread per pipeline: a pipeline is never used by two steps at once
        with ThreadPoolExecutor(max_workers=len(pools)) as ex:
            lanes = [[] for _ in pools]
            futs = {}
            for i in range(n):
                lanes[i % len(pools)].append(i)
            def lane_run(k):
                return [(i, work(i)) for i in lanes[k]]
            done = {}
            for fut in [ex.submit(lane_run, k) for k in range(len(pools))]:
                for i, r in fut.result():
                    done[i] = r
        out = None
        for i in range(n):
            out = gather_page_results(done[i], dist)
        return out

    def fence():
        torch.cuda.synchronize()
        if dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(0, args.setup_steps)):       # clocks up, code objects loaded (set 0 only; not counted as warm-up, not timed)
        pool.run_batch(pages, quads, det_maps_override=text_maps)
    # Steady state before the clock starts (round 6).  A launch shape costs a plan the first time it is seen and a hipGraph capture +
    # instantiation the second (some the third) time, and every page set brings its own rec launch shapes: through round 5 the sets the
    # warm-up had not reached met the timed region unplanned and nearly every capture happened inside it - 26 plans and ~135 captures in 20
    # steps.  On a quiet host that costs little (mean step 77.8 vs median 76.8 ms); on a busy one the first ten timed steps ran at 100 - 170 ms
    # (RD_BENCH_STEP_LOG=1, profiles/r6_step_log.txt) and the SAME tree reported anything between 310 and 412 pages/s.  A service pays that
    # once per shape, not per page: `--setup-passes` (default 3) untimed passes over ALL page sets run here, through the very code path of the
    # timed steps (uploads, prefetch, collectives); `plan_cache_misses` and `graph_captures_in_timed_region` below then read 0, and the timed
    # steps are replays only.  --setup-passes 0 = rounds 4 - 5.
    for _ in range(max(0, args.setup_passes)):
        run_steps(K_sets)
    step_no[0] = 0
    step_wall_ms.clear()
    for _ in range(args.warmup):
        for k in range(len(pools)):
            gather_page_results(compute(k), dist)
    fence()
    plans_before = sum(e.plan_stats()["plans_built"] for q in pools for e in q.engines)
    captures_before = sum(e.plan_stats()["graph_captures"] for q in pools for e in q.engines)
    t0 = time.perf_counter()
    out = run_steps(args.steps)
    fence()
    dt = time.perf_counter() - t0
    # the product's gather of whole per-page results (dist.gather_page_dets, wire format v2 - what PageAnalyzer's List[List[dict]] travels
    # in): one call on this step's lines as OcrText-span dicts, every rank, outside the timed region (the timed
    # region gathers the flat (text, score) format v1)
    from rapiddoc_amd.dist import gather_page_dets
    last_local = [(i, [{"category_id": 15, "poly": [float(k) for k in range(8)], "score": s_, "text": t_} for t_, s_ in lines])
                  for i, lines in out if i in set(my_pages)]
    fence()
    tg = time.perf_counter()
    v2 = gather_page_dets(last_local, dist)
    gather_v2_ms = (time.perf_counter() - tg) * 1e3
    assert [i for i, _ in v2] == [i for i, _ in out]
    if os.environ.get("RD_BENCH_STOP_AFTER_TIMED") == "1":      # developer: a kernel trace whose tail is the timed steps (tools/trace_gaps.py)
        print("timed region: %.2f ms per step" % (dt / args.steps * 1e3), file=sys.stderr)
        return
    plan_stats = [e.plan_stats() for q in pools for e in q.engines]
    plan_misses = sum(p["plans_built"] for p in plan_stats) - plans_before
    captures_timed = sum(p["graph_captures"] for p in plan_stats) - captures_before
    h2d_ms = None
    if upload:          # one batch's upload alone, outside the timed region (inside it the copies run under the previous batch)
        torch.cuda.synchronize()
        th = time.perf_counter()
        uploader.wait(uploader.submit(page_sets[0][0]))
        torch.cuda.synchronize()
        h2d_ms = (time.perf_counter() - th) * 1e3
    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    n_lines = sum(len(l) for _, l in out)
    import zlib
    from rapiddoc_amd.dist import encode_page_results
    result_crc = zlib.crc32(encode_page_results(out))      # of the gathered, page-ordered result: equal for every N (strong)
    host_stats = {k: round(v, 2) for k, v in pool.stats.items()}
    # host time per step (everything the host does between GPU stages), max over ranks: eight ranks share one host
    host_keys = ("t_db_post_ms", "t_descs_ms", "t_rec_enqueue_ms", "t_decode_ms")
    host_ms = float(sum(pool.stats.get(k, 0.0) for k in host_keys))
    host_ms_max = host_ms
    if dist:
        t = torch.tensor([host_ms], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        host_ms_max = float(t.item())

    # (everything below runs on rank 0 only: the width collective of the strict mode is a collective of ALL ranks - off from here on)
    for pl in pools:
        for pp in pl.pipes:
            pp.rec_width_sync = None
    # ---- roofline of the dominant kernel: per-op HIP events (recorded by the library on the launch stream)
    roof = None
    if rank == 0:
        engines = pool.engines
        for e in engines:
            e.set_profiling(True)
            e.profile_log = []
        pool.run_batch(pages, quads, det_maps_override=text_maps)
        torch.cuda.synchronize()
        agg = defaultdict(lambda: [0.0, 0.0, 0.0, 0])
        tot_ms = 0.0
        for e in engines:
            for op in e.profile_log:
                name = op["kind"]
                if op["kind"].startswith(("conv", "deconv")):  # name = the HIP kernel instantiation rocprofv3 reports
                    split = op["cfg"].endswith("/h3")
                    name = "conv_igemm%s_kernel<%s,%s>" % ("_h3" if split else "", op["cfg"].replace("/h3", ""), "1x1" if op["kind"] == "conv1x1" else "kxk")
                    if op["cfg"].startswith("stream"):
                        name = "conv_stream_h3_kernel"
                    if op["cfg"].startswith("direct"):
                        name = "conv_direct_h3_kernel"
                    if op["cfg"].startswith("c3h1"):     # round 6: the one-accumulator direct 3x3 (kernels_conv3x3_h1.hip)
                        name = "conv3x3_h1_kernel"
                    if op["cfg"].startswith("dma"):   # LDS-DMA GEMM: 8-wavefront kernel, 16-wavefront one for K <= 384
                        name = "gemm_h3_dma16_kernel" if op["cfg"].startswith("dma16w") else "gemm_h3_dma_kernel"
                    if op["cfg"].startswith("h1w"):   # round 5: the single-accumulator split GEMM (kernels_gemm_h1.hip)
                        name = "gemm_h1_kernel"
                elif op["kind"] == "mixer_fused":
                    name = "lc_mixer_kernel<%s>" % op["cfg"][1:]
                elif op["kind"] == "mixer_fused_h3":
                    name = "lc_mixer_h3_kernel<%s>" % op["cfg"][1:]
                elif op["kind"] == "mixer_fused_res":
                    name = "lc_mixer_res_kernel<%s>" % op["cfg"][1:]
                elif op["kind"] == "mixer_fused_ws":
                    name = "lc_mixer_ws_kernel<%s>" % op["cfg"][1:]
                elif op["kind"] == "ctc_head_fused_h3":
                    name = "ctc_head_h3_kernel"
                elif op["kind"] == "ctc_head_fused":
                    name = "ctc_head_kernel"
                elif op["kind"] == "stem_fused":
                    name = "stem_fused_kernel<%s>" % op["cfg"][1:]
                a = agg[name]
                a[0] += op["flops"]; a[1] += op["bytes"]; a[2] += op["ms"]; a[3] += 1
                tot_ms += op["ms"]
            e.set_profiling(False)
        mfma = {k: v for k, v in agg.items() if k.startswith(("conv_igemm", "gemm_h3", "gemm_h1", "lc_mixer", "ctc_head", "conv_direct", "conv3x3_h1", "conv_stream", "stem_fused"))}
        dom = max(mfma, key=lambda k: mfma[k][2])
        fl, by, ms, n = mfma[dom]
        ach = fl / (ms * 1e-3) / 1e12
        # HBM bytes per launch: NOT measured in this run - read from the committed summary of the last rocprofv3 --pmc
        # FETCH_SIZE / WRITE_SIZE passes (tools/collect_profiles.sh -> profiles/pmc_traffic.json); null if that file has no
        # row for today's dominant kernel.  `traffic_source` says which file / collection the number comes from.
        traffic, traffic_source, traffic_launches = None, None, None
        tf = ROOT / "profiles" / "pmc_traffic.json"
        if tf.exists():
            tj = json.loads(tf.read_text())
            row = tj.get(dom)
            if row is None:          # the counter file names instantiations in full (gemm_h3_dma_kernel<false,true>): match on the kernel name
                cands = [k for k in tj if not k.startswith("_") and k.split("<")[0] == dom.split("<")[0] and ("<" not in dom)]
                row = tj[cands[0]] if len(cands) == 1 else None
            traffic = (row or {}).get("hbm_bytes_per_launch")
            traffic_launches = (row or {}).get("dispatches_per_step")
            if traffic is not None:
                traffic_source = "profiles/pmc_traffic.json (%s)" % tj.get("_collected", "rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE, separate passes")
        # a split-fp16 kernel issues 3 fp16 MFMAs per fp32 product: its ceiling in algorithmic (fp32) FLOPs is the dense
        # fp16 MFMA peak / 3
        peak = F16_MFMA_PEAK_TFLOPS / 3.0 if ("_h3" in dom or "_h1" in dom or "_ws_" in dom or "_res_" in dom or "stem_fused" in dom) else FP32_MFMA_PEAK_TFLOPS   # split-fp16 kernels
        roof = {"bound": "mfma", "kernel": dom, "achieved": round(ach, 3), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(ach / peak, 4), "traffic": traffic, "traffic_source": traffic_source,
                # the counter passes run the bench's own configuration (8 rec streams, same launch mix): their launch count per step must equal
                # this run's for "per launch" to mean the same thing on both sides; the per-step totals are given as well
                "traffic_launches_per_step": traffic_launches,
                "traffic_bytes_per_step": None if traffic is None or traffic_launches is None else int(traffic * traffic_launches),
                "algorithmic_bytes_per_step": round(by),
                "algorithmic_bytes_per_launch": round(by / n), "launches_per_step": n,
                "avg_launch_us": round(ms * 1e3 / n, 2), "avg_gflop_per_launch": round(fl / n / 1e9, 4),
                "all_mfma_kernels_tflops": round(sum(v[0] for v in mfma.values()) / (sum(v[2] for v in mfma.values()) * 1e-3) / 1e12, 3),
                # the other large MFMA kernels of the step, same accounting (each timed alone; split-fp16 kernels against 838.9, fp32-MFMA ones against 157.3)
                "top_mfma_kernels": [{"kernel": k, "ms_per_step": round(v[2], 3), "launches": v[3], "tflops": round(v[0] / (v[2] * 1e-3) / 1e12, 1),
                                      "frac": round(v[0] / (v[2] * 1e-3) / 1e12 /
                                                    (F16_MFMA_PEAK_TFLOPS / 3.0 if ("_h3" in k or "_h1" in k or "_ws_" in k or "_res_" in k or "stem_fused" in k) else FP32_MFMA_PEAK_TFLOPS), 4)}
                                     for k, v in sorted(mfma.items(), key=lambda kv: -kv[1][2])[:6]],
                "step_kernel_ms": round(tot_ms, 2),
                # summed over the concurrent streams (det / layout / 8 rec / tail): it exceeds ms_per_step when kernels overlap
                "kernel_ms_overlapped": True}
        if args.dump_profile:
            with open(args.dump_profile + ".ops.json", "w") as f:
                json.dump({"det": sum((q.det.profile_log for q in pool.pipes), []),
                           "rec": sum((e.profile_log for q in pool.pipes for e in q.rec_engines + [q.rec_tail]), []),
                           "layout": sum((q.layout.profile_log for q in pool.pipes), [])}, f)
            table = sorted(((k, v[3], v[2], v[0] / 1e9, v[1] / 1e6) for k, v in agg.items()), key=lambda r: -r[2])
            with open(args.dump_profile, "w") as f:
                f.write("kernel,launches,total_ms,gflop,algorithmic_MB,TFLOPs,GBs\n")
                for k, n_, ms_, gf, mb in table:
                    f.write("%s,%d,%.3f,%.2f,%.1f,%.2f,%.1f\n" % (k, n_, ms_, gf, mb, gf / ms_ if ms_ else 0, mb / ms_ if ms_ else 0))

    # ---- post-passes (N = 1 only, rank 0, after the timed region; each a few steps): the OTHER rec batching mode, the pure
    # fp32-MFMA precision mode, the layout backbone alone.  All measured in this very run - no constants in the line.
    extra = {}
    if rank == 0 and world == 1 and not args.no_extra_passes:
        def timed_steps(fn, n, w):
            for _ in range(w):
                fn()
            torch.cuda.synchronize()
            t0_ = time.perf_counter()
            for _ in range(n):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0_) / n
        other = "strict" if args.rec_mode == "throughput" else "throughput"
        pool2 = PagePipelinePool(states, rec_mode=other, **pool_kw)
        # (one resident page set; pipelined like the timed region: each call announces the next batch - the same tensor - so that its
        #  det + layout forwards run under this call's recognition)
        pf = pages if prefetch_on else None
        sec = timed_steps(lambda: pool2.run_batch(pages, quads, det_maps_override=text_maps, prefetch=pf), 3, 2)
        key = "strict_rec_batching" if other == "strict" else "throughput_rec_batching"
        extra[key] = {"pages_s": round(P / sec, 3), "ms_per_step": round(sec * 1e3, 3), "steps": 3, "warmup": 2, "front_prefetch": bool(prefetch_on),
                      "rec_launch_batches": int(pool2.stats.get("rec_batches", 0)),
                      "rule": _strict_rule(n_lines) if other == "strict" else _throughput_rule(args)}
        del pool2
        if pipe.det.precision == "auto":
            for e in pool.engines:
                e.set_precision("fp32")
            sec = timed_steps(lambda: pool.run_batch(pages, quads, det_maps_override=text_maps, prefetch=pf), 2, 1)
            extra["fp32_precision_mode"] = {"pages_s": round(P / sec, 3), "ms_per_step": round(sec * 1e3, 3), "steps": 2, "warmup": 1, "front_prefetch": bool(prefetch_on),
                                            "what": "RD_PRECISION=fp32: every dense layer on v_mfma_f32_32x32x2_f32, same pages"}
            for e in pool.engines:
                e.set_precision("auto")
        extra["formula"] = measure_formula()
        m = measure_backbone(pipe, pages, 5, 2)
        extra["backbone"] = {"metric": "pages/sec (PP-DocLayout backbone only: pre-process + PPHGNetV2-B4 @800x800)",
                             "pages_s": round(P * 5 / m["dt"], 3), "ms_per_step": round(m["dt"] / 5 * 1e3, 3), "steps": 5, "warmup": 2,
                             "gflop_per_page": m["gflop_per_page"], "roofline": m["roofline"]}
        # the same backbone with every dense layer on v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, peak 157.3 TFLOP/s): the arithmetic
        # north_star's ">= 40 % MFMA roofline" was written for, next to the default mode above
        if pipe.layout is not None and pipe.layout.precision == "auto":
            pipe.layout.set_precision("fp32")
            m32 = measure_backbone(pipe, pages, 3, 1)
            pipe.layout.set_precision("auto")
            extra["backbone"]["fp32_mode"] = {"pages_s": round(P * 3 / m32["dt"], 3), "ms_per_step": round(m32["dt"] / 3 * 1e3, 3),
                                              "tflops": m32["roofline"]["achieved"], "peak": m32["roofline"]["peak"],
                                              "frac": m32["roofline"]["frac"], "kernel_ms": m32["roofline"]["kernel_ms"]}
        extra["s2_dropin"] = measure_s2_dropin(states, pipe, pages, text_maps, dev_index)

    if rank == 0:
        total_pages = n_global * args.steps
        rec = {
            "metric": "pages/sec (layout+OCR det/rec)", "value": round(total_pages / dt, 3), "unit": "pages/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "PP-DocLayout backbone (PPHGNetV2-B4 @800x800) + PP-OCRv6-small det (960x704) + rec "
                                   "(45 lines/page, fused CTC) on %d synthetic 1684x1191 pages per GPU (rank r takes pages r, r+N, ... of one %d-page list)" % (P, n_global),
                       "precision": "auto: fp32 in / fp32 accumulate / fp32 out; products of the channel mixers, the CTC head and the "
                                    "wide convs on split-fp16 MFMA (x = hi + lo, 3 MFMAs per product; error vs fp64 at or below the fp32 "
                                    "MFMA kernels', tests/test_gpu_parity.py::test_fused_mixer_kernels_match_fp64 / test_split_gemm_*), "
                                    "fp32 MFMA for the rest; range-guarded with fp32 fallback (DESIGN.md s3)"
                                    if pipe.det.precision == "auto" else pipe.det.precision,
                       "rec_batching": "strict: " + _strict_rule(n_lines) if args.rec_mode == "strict" else
                                       "throughput (%s; the reference's own batching is timed in strict_rec_batching)" % _throughput_rule(args),
                       "rec_mode": args.rec_mode,
                       "rec_launch_batches": int(pool.stats.get("rec_batches", 0)),
                       "pages_per_gpu": P, "global_pages": n_global, "pages_gathered": len(out), "result_crc32": result_crc,
                       "gather_page_dets_v2_ms": round(gather_v2_ms, 2),      # one gather of this step's results as per-page dict lists (wire format v2), all ranks
                       "rec_width_sync": (None if width_sync is None else {"collective_calls": width_sync.calls,
                                          "what": "global argsort / chunks of 6 over the lines of all ranks (dist.GlobalLineWidths)"}),
                       "page_sets_cycled": K_sets, "setup_steps": max(0, args.setup_steps), "setup_passes_over_all_page_sets": max(0, args.setup_passes),
                       "pages_start_in": "pinned host memory: every step's %.0f MB are uploaded inside the timed region (PageUploader: copy "
                                         "stream, batch i + 1 under batch i; the first batch's copy is exposed)" % (pages_np.nbytes / 1e6)
                                         if upload else "HBM (resident, --resident-pages)",
                       "h2d_ms_per_batch_alone": None if h2d_ms is None else round(h2d_ms, 3),
                       # this rank's host-clock time of each timed step (a step returns when its last line is decoded): the mean
                       # is what `value` is made of, the median is the steady state, the maximum shows a stalled step
                       "step_wall_ms": ({"median": round(float(np.median(step_wall_ms)), 2), "min": round(min(step_wall_ms), 2),
                                         "max": round(max(step_wall_ms), 2), "first": round(step_wall_ms[0], 2)} if step_wall_ms else None),
                       "gpu_max_hw_queues": os.environ.get("GPU_MAX_HW_QUEUES"),
                       "front_prefetch": bool(prefetch_on),       # det + layout of batch i + 1 enqueued under the recognition of batch i
                       "plan_cache_misses": int(plan_misses),     # per-shape plans built INSIDE the timed region (new rec / tail shapes of unseen page sets)
                       "hipgraph": {"captures": int(sum(p["graph_captures"] for p in plan_stats)), "replays": int(sum(p["graph_replays"] for p in plan_stats))},
                       "graph_captures_in_timed_region": int(captures_timed),
                       "lines_per_step": n_lines, "host_stage_ms": host_stats,
                       "host_ms_per_step_max_over_ranks": round(host_ms_max, 2), "cores_per_rank": cores_per_rank,
                       "range_fallbacks": int(sum(e.range_fallbacks for q in pools for e in q.engines)),   # engines that left the split-fp16 mode (0 = the dtype claim holds)
                       "parallelism": "page-sharded dp%d; %d page batch(es) in flight per GPU" % (world, len(pools)),
                       "backend": group_info["backend"], "world_size": group_info["world_size"],
                       "rccl_ranks_seen": group_info["rccl_ranks_seen"], "distinct_devices": group_info["distinct_devices"],
                       "devices": group_info["devices"],
                       "layout_head": "absent (ONNX-only in the reference; backbone only)",
                       "det_postprocess": "DB post-process runs on maps rendered from the generator's line boxes "
                                          "(random-weight det output has no text); its boxes drive crop+rec"},
            "roofline": roof,
        }
        rec.update(extra)
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(states, cores_per_rank or os.cpu_count() or 1)
        print(json.dumps(rec), flush=True)
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
