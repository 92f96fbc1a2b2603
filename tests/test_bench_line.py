"""The line the driver parses (bench.py's LAST stdout line) stays small: round 4's 23 KB line did not parse in the driver's record
(VERDICT r04 #1).  Fed with a full report of the round-4 shape (profiles/r04_bench.json: headline + seven `also` reports) and with
the same at eight ranks."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _full():
    d = json.load(open(os.path.join(ROOT, "profiles", "r04_bench.json")))
    also = d.pop("also")
    return d, also


def test_compact_line_one_rank():
    full, also = _full()
    line = bench.compact_line(full, also)
    text = json.dumps(line)
    assert len(text) <= 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "timed_steps", "timed_seconds", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "also_summary"):
        assert k in line, k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    for k in ("workload", "frames_per_step_per_gpu", "lanes", "parity_checked_frames", "parity_mismatches", "library_build_id"):
        assert k in line["config"], k
    assert line["value"] == full["value"] and line["roofline"]["frac"] == full["roofline"]["frac"]
    assert set(line["also_summary"]) == set(also)
    for k, v in also.items():
        assert line["also_summary"][k]["value"] == v["value"]
    assert not text.startswith(bench.DETAIL_PREFIX) and text.startswith("{")


def test_compact_line_eight_ranks():
    full, also = _full()
    full["n_gpus"] = 8
    full["per_rank"] = [dict(copy.deepcopy(full["per_rank"][0]), rank=r, device=r) for r in range(8)]
    for v in also.values():
        if v.get("per_rank"):
            v["per_rank"] = [dict(v["per_rank"][0], rank=r) for r in range(8)]
    text = json.dumps(bench.compact_line(full, also))
    assert len(text) <= 6144, len(text)
    line = json.loads(text)
    assert len(line["per_rank"]["rows"]) == 8 and line["per_rank"]["columns"][0] == "rank"


def test_compact_line_of_the_final_build_with_live_traffic():
    """round 5's final shape: eight `also` rows (the int8 matcher line beside the FP4 and popcount ones), `roofline.traffic` measured in the run,
    the step's total traffic in `roofline_pipeline` — one rank <= 4 KB, eight ranks <= 6 KB"""
    d = json.load(open(os.path.join(ROOT, "profiles", "r05b_bench_live_traffic.json")))
    also = d.pop("also")
    d.pop("line", None)
    line = bench.compact_line(d, also)
    text = json.dumps(line)
    assert len(text) <= 4096, len(text)
    assert set(line["also_summary"]) == {"vga_extract", "hd1080", "match100k", "match100k_int8", "match100k_popcount", "vga_noise", "vga_midtex", "vga_lowtex"}
    assert line["roofline"]["traffic_measured_in_this_run"] is True and line["roofline"]["traffic"] == d["roofline"]["traffic"]
    assert line["roofline_pipeline"]["traffic"] == sum(d["traffic_per_stage_this_run"].values())
    assert line["also_summary"]["match100k"]["roofline_bound"] == "mfma" and "roofline_bound" not in line["also_summary"]["hd1080"]       # "hbm" rows omit the key
    d["n_gpus"] = 8
    d["per_rank"] = [dict(copy.deepcopy(d["per_rank"][0]), rank=r, device=r) for r in range(8)]
    assert len(json.dumps(bench.compact_line(d, {k: also[k] for k in ("hd1080", "match100k")}))) <= 6144


def test_live_traffic_refuses_to_nest_under_a_profiler(monkeypatch):
    """the child runs of `measure_live_traffic` would inherit a profiler's environment: the replayed table stays in the line then"""
    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "x")
    import shutil
    if not (shutil.which("rocprofv3") or os.path.exists("/opt/rocm/bin/rocprofv3")):
        import pytest
        pytest.skip("no rocprofv3 in this image")
    lt, why = bench.measure_live_traffic(1024, timeout_s=1.0)
    assert lt is None and "profiler" in why


def test_match_report_compacts():
    _, also = _full()
    line = bench.compact_line(also["match100k"])
    assert line["roofline"]["bound"] == "mfma" and "hbm" in line["roofline"] and line["roofline"]["traffic"] is None
    assert len(json.dumps(line)) <= 4096


def test_bare_multi_gpu_command_refuses_missing_devices():
    """`python bench.py --gpus N` starts its own ranks only when the node shows N devices: on a box with fewer it says so
    and exits 2 instead of spawning ranks that die one by one (this container has no GPU at all)."""
    import subprocess
    import torch
    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(max(n, 2))], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode == 2 and "device(s)" in r.stderr and r.stdout.strip() == "", (r.returncode, r.stderr[-300:])
