"""GPU: the multi-process path of bench.py on ONE GPU (two ranks over gloo, RD_BENCH_BACKEND=gloo): rank r takes pages
{i : i mod 2 = r} of one global page list, results travel through the flat byte format and one padded all-gather, and the
page-ordered result must be byte-identical (crc32) to the single-process run over the same global list - in the DEFAULT (strict)
rec mode: a line is padded like its chunk of six of the GLOBAL pooled, sorted line list (the reference pools a whole page batch,
rapid_ocr.py:404-449), which every rank rebuilds from one more small all-gather (dist.GlobalLineWidths; VERDICT r4 missing #3)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
COMMON = ["--scaling", "strong", "--global-pages", "4", "--rec-streams", "2", "--steps", "1", "--warmup", "0", "--setup-steps", "0", "--setup-passes", "0", "--no-cpu-baseline",
          "--no-extra-passes", "--vary-pages", "1"]
# rounds 3-4 needed this to get equal strings: one line per rec batch in the throughput mode (still covered below)
THROUGHPUT_1 = ["--rec-mode", "throughput", "--rec-chunking", "fixed", "--rec-batch", "1"]


def _last_json(out: str) -> dict:
    lines = [l for l in out.splitlines() if l.startswith("{")]
    assert lines, out[-2000:]
    return json.loads(lines[-1])


def _run(cmd, env, timeout=420):
    """subprocess.run with the child in a process group of its own that is killed as a whole on a timeout: a hung multi-rank run
    (a collective one rank never joins) must not leave its ranks behind on the GPU and the host cores of the tests that follow."""
    import signal
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT, start_new_session=True)
    try:
        out, err = p.communicate(timeout=timeout)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGKILL)
        out, err = p.communicate()
        raise AssertionError("timed out after %d s: %s\n%s" % (timeout, " ".join(cmd[-12:]), err[-1500:]))
    assert p.returncode == 0, err[-2000:]
    return out


def _clean_env(**kw):
    """The caller's environment without a launcher's variables: a bare `python bench.py --gpus N` must start its ranks itself."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LOCAL_WORLD_SIZE", "MASTER_PORT",
                                                            "TORCHELASTIC_RUN_ID", "GROUP_RANK", "ROLE_RANK")}
    env.update(MASTER_ADDR="127.0.0.1", **kw)
    return env


def _one_and_two(extra, port, env_two=None):
    env = _clean_env()
    one = _run([sys.executable, str(ROOT / "bench.py"), *COMMON, *extra], env)
    env2 = dict(env, RD_BENCH_BACKEND="gloo", **(env_two or {}))
    two = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                "--master-port", str(port), str(ROOT / "bench.py"), "--gpus", "2", *COMMON, *extra], env2)
    return _last_json(one), _last_json(two)


def test_bare_gpus_2_starts_two_ranks_itself_and_equals_one_rank_in_the_default_strict_mode():
    """`python bench.py --gpus 2` with no launcher around it: bench.py re-executes itself under torch.distributed.run,
    the line says n_gpus 2, the process group reports two ranks, and the page-ordered result equals the single-process one."""
    env = _clean_env()
    a = _last_json(_run([sys.executable, str(ROOT / "bench.py"), *COMMON], env))
    b = _last_json(_run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *COMMON], dict(env, RD_BENCH_BACKEND="gloo")))
    assert a["n_gpus"] == 1 and b["n_gpus"] == 2 and b["scaling"] == "strong"
    assert a["config"]["world_size"] == 1 and a["config"]["backend"] is None
    assert b["config"]["world_size"] == 2 and b["config"]["backend"] == "gloo" and b["config"]["rccl_ranks_seen"] == 2
    assert len(b["config"]["devices"]) == 2 and b["config"]["distinct_devices"] == 1       # two ranks of the test mode on the box's one GPU
    assert a["config"]["rec_mode"] == b["config"]["rec_mode"] == "strict"
    assert a["config"]["pages_gathered"] == b["config"]["pages_gathered"] == 4
    assert b["config"]["pages_per_gpu"] == 2
    assert a["config"]["lines_per_step"] == b["config"]["lines_per_step"] == 180
    assert b["config"]["rec_width_sync"]["collective_calls"] >= 1 and a["config"]["rec_width_sync"] is None
    assert a["config"]["result_crc32"] == b["config"]["result_crc32"]


def test_four_pages_over_four_ranks_one_page_each_equal_one_rank():
    """The smallest shards there are: one page per rank, so every rank's launches hold a quarter of the lines -
    and the gathered, page-ordered strings and confidences are still the single-process run's, byte for byte, in the default precision
    mode (kernels are picked by the layer, not by the launch size: tests/test_gpu_launch_invariance.py)."""
    env = _clean_env()
    a = _last_json(_run([sys.executable, str(ROOT / "bench.py"), *COMMON], env))
    b = _last_json(_run([sys.executable, str(ROOT / "bench.py"), "--gpus", "4", *COMMON], dict(env, RD_BENCH_BACKEND="gloo"), timeout=600))
    assert b["n_gpus"] == 4 and b["config"]["world_size"] == 4 and b["config"]["pages_per_gpu"] == 1
    assert "auto" in b["config"]["precision"] and b["config"]["range_fallbacks"] == 0
    assert a["config"]["pages_gathered"] == b["config"]["pages_gathered"] == 4
    assert a["config"]["result_crc32"] == b["config"]["result_crc32"]


def test_more_ranks_than_devices_over_rccl_fails_loudly():
    """A bare `--gpus 8` on a box with fewer devices must not print an N = 1 number: one rank per GPU over RCCL needs eight devices."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip("eight devices visible")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "8", *COMMON], capture_output=True, text=True, cwd=ROOT,
                       env=_clean_env(), timeout=300)
    assert p.returncode != 0 and "device(s) visible" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_gpus_flag_must_match_the_launchers_world_size():
    env = _clean_env(RD_BENCH_BACKEND="gloo", WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--gpus", "2", *COMMON], capture_output=True, text=True, cwd=ROOT, env=env, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in p.stderr


def test_without_the_width_collective_the_strict_strings_depend_on_the_rank_count():
    """The control: the same two-rank run with the collective switched off pools per rank - the crc differs (if it did not, the test
    above would prove nothing about the collective)."""
    a, b = _one_and_two([], 29745, env_two={"RD_BENCH_WIDTH_SYNC": "0"})
    assert b["config"]["rec_width_sync"] is None
    assert a["config"]["result_crc32"] != b["config"]["result_crc32"]


def test_two_ranks_on_one_gpu_equal_one_rank_throughput_mode_single_line_batches():
    a, b = _one_and_two(THROUGHPUT_1, 29747)
    assert a["config"]["pages_gathered"] == b["config"]["pages_gathered"] == 4
    assert a["config"]["result_crc32"] == b["config"]["result_crc32"]


def test_two_ranks_weak_scaling_cover_the_same_global_list():
    """--scaling weak: 2 pages per rank x 2 ranks = the same 4-page global list (rank r takes pages r, r + 2): same crc as the
    single-process run over 4 pages."""
    env = _clean_env()
    base = ["--rec-streams", "2", "--steps", "1", "--warmup", "0", "--setup-steps", "0", "--setup-passes", "0", "--no-cpu-baseline", "--no-extra-passes", "--vary-pages", "1"]
    a = _last_json(_run([sys.executable, str(ROOT / "bench.py"), "--scaling", "weak", "--pages", "4", *base], env))
    b = _last_json(_run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                         "--master-port", "29743", str(ROOT / "bench.py"), "--gpus", "2", "--scaling", "weak", "--pages", "2", *base],
                        dict(env, RD_BENCH_BACKEND="gloo")))
    assert b["n_gpus"] == 2 and b["scaling"] == "weak" and b["config"]["pages_per_gpu"] == 2
    assert a["config"]["pages_gathered"] == b["config"]["pages_gathered"] == 4
    assert a["config"]["result_crc32"] == b["config"]["result_crc32"]
