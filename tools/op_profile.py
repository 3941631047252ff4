"""Per-op time table of one forward (rd_set_profiling: every op is bracketed by events, so the times are serialised ones):
python tools/op_profile.py ppocrv6_rec 64 48 1056 | ppocrv6_det 8 960 704 | pphgnetv2_b4 8 800 800"""
import sys, collections
import torch
sys.path.insert(0, ".")
from rapiddoc_amd.engine import RdEngine
from rapiddoc_amd import weights as W

kind, B, H, Wd = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
eng = RdEngine(kind, 0, guard="off")
eng.load_weights(W.synth_state_dict(W.load_manifest(f"tests/golden/manifest_{kind}.json"), 0))
x = torch.rand((B, 3, H, Wd), device="cuda") * 2 - 1
fn = {"ppocrv6_rec": eng.rec_forward, "ppocrv6_det": eng.det_forward}.get(kind, getattr(eng, "backbone_forward", None))
for _ in range(3):
    fn(x)
torch.cuda.synchronize()
eng.set_profiling(True)
eng.profile_log.clear()
for _ in range(5):
    fn(x)
torch.cuda.synchronize()
log = eng.profile_log
eng.set_profiling(False)
agg = collections.OrderedDict()
for r in log:
    k = (r.get("name"), r.get("kind"), r.get("cfg", "") + " " + r.get("shape", ""))
    a = agg.setdefault(k, [0.0, 0, r.get("flops", 0.0), r.get("bytes", 0.0)])
    a[0] += r.get("ms", 0.0); a[1] += 1
tot = sum(a[0] for a in agg.values()) / 5
print(f"total {tot:.3f} ms per forward, {len(agg)} ops")
rows = sorted(agg.items(), key=lambda kv: -kv[1][0])
for (name, kind_, kern), (ms, n, fl, by) in rows[:int(sys.argv[5]) if len(sys.argv) > 5 else 45]:
    ms /= n
    print(f"{ms*1e3:9.1f} us  {fl/ms/1e9 if ms else 0:7.1f} TF/s {by/ms/1e6 if ms else 0:8.1f} GB/s  {kind_:10s} {kern[:44]:44s} {name}")
