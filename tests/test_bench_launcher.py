"""bench.py starts its own ranks: `python bench.py --gpus 2` from a plain shell must spawn two
fresh rank processes (torch.distributed.run), run the sharded sweep and relay rank 0's JSON
line.  Here on the CPU: gloo, the tiny workload, the oracle-backed engine double."""
import json
import os
import subprocess
import sys

import pytest

from util import ROOT


def _run(extra, env_extra=None):
    env = dict(os.environ, PYTHONPATH=os.pathsep.join([os.path.join(ROOT, "tests"), ROOT]))
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.update(env_extra or {})
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "mini", "--device", "cpu",
           "--engine-factory", "bench_engine_double:make", "--steps", "2", "--warmup", "1", "--cpu-poses", "0"] + extra
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_plain_shell_launch_two_ranks_strong_default_and_weak_record():
    out = _run(["--gpus", "2"])
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2
    assert out["scaling"] == "strong" and out["config"]["poses"] == 700 and out["config"]["poses_per_gpu"] == 350
    assert out["weak"]["scaling"] == "weak" and out["weak"]["poses"] == 1400 and out["weak"]["poses_per_gpu"] == 700
    assert out["steps"] == 2 and out["warmup"] == 1 and out["value"] > 0 and out["weak"]["value"] > 0
    assert abs(out["value"] - 699 * 2 / (out["ms_per_step"] * 2e-3)) / out["value"] < 1e-3
    for key in ("metric", "unit", "higher_is_better", "vs_baseline", "dtype", "data", "config"):
        assert key in out
    assert out["dtype"] == "f64" and out["vs_baseline"] is None


def test_single_rank_goes_through_the_same_path():
    out = _run(["--gpus", "1", "--force-sharded"])
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and "weak" not in out
    assert out["config"]["poses"] == 700


def test_gpus_flag_must_match_the_rank_environment():
    env = dict(os.environ, WORLD_SIZE="3", RANK="0", PYTHONPATH=os.path.join(ROOT, "tests"))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--device", "cpu",
                        "--engine-factory", "bench_engine_double:make"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert p.returncode != 0 and "WORLD_SIZE" in p.stderr
