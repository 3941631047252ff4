"""GPU: the multi-process path of bench.py on ONE GPU (two ranks over gloo, RD_BENCH_BACKEND=gloo): rank r takes pages
{i : i mod 2 = r} of one global page list, results travel through the flat byte format and one padded all-gather, and the
page-ordered result must be byte-identical (crc32) to the single-process run over the same global list.  rec batches of one
line keep the recogniser's input independent of which other lines share a rank (LightSVTR attends over padded columns)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
# (--rec-mode throughput with one line per rec batch: a line's result then does not depend on which other lines its rank holds; in the
#  default strict mode a line is padded like its chunk of six of the rank's pooled lines, exactly as the reference pools a page batch)
COMMON = ["--rec-mode", "throughput", "--scaling", "strong", "--global-pages", "4", "--rec-chunking", "fixed", "--rec-batch", "1", "--rec-streams", "2", "--steps", "1",
          "--warmup", "0", "--setup-steps", "0", "--no-cpu-baseline"]


def _last_json(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def test_two_ranks_on_one_gpu_equal_one_rank():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    one = subprocess.run([sys.executable, str(ROOT / "bench.py"), *COMMON], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _last_json(one.stdout)
    env2 = dict(env, RD_BENCH_BACKEND="gloo")
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29741", str(ROOT / "bench.py"), "--gpus", "2", *COMMON],
                         capture_output=True, text=True, timeout=900, env=env2, cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    b = _last_json(two.stdout)
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert a["config"]["pages_gathered"] == b["config"]["pages_gathered"] == 4
    assert b["config"]["pages_per_gpu"] == 2
    assert a["config"]["lines_per_step"] == b["config"]["lines_per_step"] == 180
    assert a["config"]["result_crc32"] == b["config"]["result_crc32"]


def test_two_ranks_weak_scaling_cover_the_same_global_list():
    """--scaling weak: 2 pages per rank x 2 ranks = the same 4-page global list (rank r takes pages r, r + 2): same crc as the
    single-process run over 4 pages."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    base = ["--rec-mode", "throughput", "--rec-chunking", "fixed", "--rec-batch", "1", "--rec-streams", "2", "--steps", "1", "--warmup", "0", "--setup-steps", "0", "--no-cpu-baseline"]
    one = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--scaling", "weak", "--pages", "4", *base], capture_output=True, text=True,
                         timeout=900, env=env, cwd=ROOT)
    assert one.returncode == 0, one.stderr[-2000:]
    a = _last_json(one.stdout)
    two = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29743", str(ROOT / "bench.py"), "--gpus", "2", "--scaling", "weak", "--pages", "2", *base],
                         capture_output=True, text=True, timeout=900, env=dict(env, RD_BENCH_BACKEND="gloo"), cwd=ROOT)
    assert two.returncode == 0, two.stderr[-2000:]
    b = _last_json(two.stdout)
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["config"]["pages_per_gpu"] == 2
    assert a["config"]["pages_gathered"] == b["config"]["pages_gathered"] == 4
    assert a["config"]["result_crc32"] == b["config"]["result_crc32"]
