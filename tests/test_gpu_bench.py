"""bench.py as the driver runs it: a bare command (no torchrun) that starts its own ranks for --gpus N > 1 and prints ONE JSON
line carrying the contract fields.  On the 1-GPU box the two ranks share cuda:0 and talk over gloo: a functional check of the
N > 1 path (sharding, barrier, MAX / all-gather of the counters), not a scaling number."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, timeout=timeout, cwd=ROOT, env=env)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    return json.loads(lines[0])


def _contract(d):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
              "roofline"):
        assert k in d, k
    assert "workload" in d["config"] and d["value"] > 0 and d["roofline"]["bound"] in ("hbm", "valu_issue", "mfma")
    r = d["roofline"]
    hbm = r if r["bound"] == "hbm" else r["hbm"]
    for k in ("achieved", "peak", "unit", "frac", "traffic"):
        assert k in hbm, k


def test_two_ranks_bare_command():
    d = _bench("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "2", "--warmup", "1", "--batch", "64", "--ring", "128", "--min-seconds", "0",
               "--no-cpu-baseline")
    _contract(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["per_rank"]) == 2
    assert d["per_rank"][0]["frames"] == d["per_rank"][1]["frames"] == 2 * 64
    assert d["config"]["frames_with_error_status"] == 0 and d["config"]["mean_keypoints_per_frame"] > 900


def test_single_rank_configs_and_min_duration():
    d = _bench("--config", "vga_extract", "--steps", "2", "--warmup", "1", "--batch", "128", "--ring", "256", "--min-seconds", "0.3", "--no-cpu-baseline")
    _contract(d)
    assert d["n_gpus"] == 1 and "extract @640x480" in d["metric"] and d["repeats"] >= 1 and d["timed_steps"] == d["repeats"] * 2
    assert d["timed_seconds"] >= 0.12          # calibrated from one untimed block: the region may come out a little short of --min-seconds
    m = _bench("--config", "match100k", "--steps", "2", "--warmup", "1", "--min-seconds", "0", "--no-cpu-baseline")
    _contract(m)
    assert m["unit"] == "pairs/s" and m["roofline"]["bound"] == "mfma" and m["value"] > 5e11
