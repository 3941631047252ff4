"""Developer: per-call cost of the rec S2 session (lazy / eager) on chunks of six lines.  python tools/mb_s2_calls.py [torch threads]"""
import sys, time
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import numpy as np
import torch
if len(sys.argv) > 1:
    torch.set_num_threads(int(sys.argv[1]))
print("torch threads", torch.get_num_threads())
import bench
from rapiddoc_amd.session import Mi355RecSession
sess = Mi355RecSession(bench.load_states()["ppocrv6_rec"], 0)
rng = np.random.default_rng(0)
widths = [320, 640, 1000, 1400]
xs = {w: rng.uniform(-1, 1, (6, 3, 48, w)).astype(np.float32) for w in widths}
for lazy in (True, False):
    sess.lazy_softmax = lazy
    for w in widths:
        for _ in range(6):
            p = sess(xs[w]); p.argmax(axis=2)
        sess.host_ms = {k: 0 for k in sess.host_ms}
        s0 = sess.engine.plan_stats()
        t0 = time.perf_counter()
        for _ in range(20):
            p = sess(xs[w]); a = p.argmax(axis=2); m = p.max(axis=2)
        dt = (time.perf_counter() - t0) / 20 * 1e3
        s1 = sess.engine.plan_stats()
        print("lazy=%d W=%4d  call %.3f ms  (stage_in %.3f, forward_and_wait %.3f)  graph replays %d" % (
            lazy, w, dt, sess.host_ms["stage_in"] / 20, sess.host_ms["forward_and_wait"] / 20, s1["graph_replays"] - s0["graph_replays"]))
