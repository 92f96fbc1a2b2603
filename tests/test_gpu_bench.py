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
    out = res.stdout.splitlines()
    lines = [ln for ln in out if ln.startswith("{")]
    assert len(lines) == 1 and out[-1] == lines[0], res.stdout[-3000:]           # ONE JSON line, and it is the last line of stdout
    line = json.loads(lines[0])
    # the driver's record keeps a bounded tail of stdout: the line must stay small (23 KB did not parse)
    assert len(lines[0]) <= (4096 if line["n_gpus"] == 1 else 6144), len(lines[0])
    for k in ("roofline", "config", "timed_steps", "timed_seconds"):
        assert k in line, k
    # the full reports travel in front of it as "#detail <name> <json>" lines; the tests read the headline's and hang the others under `also`
    detail = {}
    for ln in out:
        if ln.startswith("#detail "):
            _, name, body = ln.split(" ", 2)
            detail[name] = json.loads(body)
    d = detail["headline"]
    assert d["value"] == line["value"] and d["ms_per_step"] == line["ms_per_step"] and d["config"]["workload"] == line["config"]["workload"]
    assert line["roofline"]["frac"] == d["roofline"]["frac"] and line["roofline"]["bound"] == d["roofline"]["bound"]
    d["also"] = {k[5:]: v for k, v in detail.items() if k.startswith("also.")}
    d["line"] = line
    if d["also"]:
        assert set(line["also_summary"]) == set(d["also"])
        for k, v in d["also"].items():
            assert line["also_summary"][k]["value"] == v["value"] and line["also_summary"][k]["parity_mismatches"] == v["config"]["parity_mismatches"]
    return d


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
    """the N > 1 path of all three configurations of the default line: VGA stream per rank, 1080p stream per rank (BASELINE configs[3]),
    100k x 100k with the queries sharded by rank"""
    import time
    t0 = time.monotonic()
    d = _bench("--gpus", "2", "--backend", "gloo", "--share-device", "--steps", "2", "--warmup", "1", "--batch", "64", "--ring", "128", "--min-seconds", "0",
               "--also-min-seconds", "0", "--also-match-min-seconds", "0", "--no-cpu-baseline")
    assert time.monotonic() - t0 < 120.0                 # the N > 1 default run is a short one (sampled parity, no family entries)
    _contract(d)
    assert set(d["also"]) == {"hd1080", "match100k"} and "EVERY" not in d["config"]["parity_note"]      # sampled parity by default at N > 1
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and len(d["per_rank"]) == 2
    assert d["per_rank"][0]["frames"] == d["per_rank"][1]["frames"] == 2 * 64
    assert d["config"]["frames_with_error_status"] == 0 and d["config"]["mean_keypoints_per_frame"] > 900
    assert d["config"]["parity_checked_frames"] >= 16 and d["config"]["parity_mismatches"] == 0       # both ranks checked their last step
    assert d["config"]["host_submit_ms_per_step"] > 0
    hd, mt = d["also"]["hd1080"], d["also"]["match100k"]
    assert hd["scaling"] == "weak" and len(hd["per_rank"]) == 2 and hd["per_rank"][0]["frames"] == hd["per_rank"][1]["frames"] == 2 * 256
    assert hd["config"]["parity_checked_frames"] >= 16 and hd["config"]["parity_mismatches"] == 0
    assert mt["scaling"] == "strong" and mt["config"]["queries_per_gpu"] == 50000 and len(mt["per_rank"]) == 2
    assert mt["config"]["parity_checked_rows"] >= 100 and mt["config"]["parity_mismatches"] == 0


def test_default_line_carries_every_baseline_config():
    """the command the driver runs, shortened: headline VGA keys unchanged, `also` holds hd1080 (BASELINE configs[2]) and match100k
    (configs[4]) with their own roofline and cpu_baseline; the parity leg ran on the timed shapes"""
    d = _bench("--steps", "2", "--warmup", "1", "--min-seconds", "0.3", "--also-min-seconds", "0.2", "--also-match-min-seconds", "0.2", "--cpu-seconds", "1",
               "--cpu-allcores-seconds", "0", "--also-cpu-seconds", "1", "--cpu-reference-seconds", "1")
    _contract(d)
    _contract(d["line"])
    assert d["line"]["cpu_baseline"]["value"] == d["cpu_baseline"]["value"] and d["line"]["cpu_baseline"]["cores"] == 1
    assert d["line"]["config"]["parity_checked_frames"] == 1024 and d["line"]["config"]["library_build_id"]
    assert "640x480" in d["metric"] and d["roofline"]["bound"] == "hbm" and d["roofline"]["kernel"] and "cpu_baseline" in d
    assert d["config"]["frames_per_step_per_gpu"] == 1024 and d["config"]["lanes"] == 4
    assert d["config"]["parity_mismatches"] == 0
    hd, mt = d["also"]["hd1080"], d["also"]["match100k"]
    assert "1920x1080" in hd["metric"] and hd["value"] > 0 and hd["roofline"]["bound"] == "hbm" and hd["cpu_baseline"]["value"] > 0
    assert hd["config"]["parity_checked_frames"] == 256 and hd["config"]["parity_mismatches"] == 0          # every frame of the last step
    assert d["config"]["parity_checked_frames"] == 1024
    # configs[1] and the other frame families ride along (VERDICT r03 missing #2 / #3)
    ex = d["also"]["vga_extract"]
    assert "extract @640x480" in ex["metric"] and ex["value"] > 0 and ex["config"]["parity_mismatches"] == 0
    for key, name in (("vga_noise", "S-noise"), ("vga_midtex", "S-midtex"), ("vga_lowtex", "S-lowtex")):
        f = d["also"][key]
        assert name in f["config"]["workload"] and f["value"] > 0 and f["config"]["parity_checked_frames"] == 1024 and f["config"]["parity_mismatches"] == 0, key
        assert f["stage_ms_per_step"]["fast_cells"] > 0
    # round 6: the correlated stream — every frame of the step against the oracle like the others, and a camera-like share of accepted matches
    wv = d["also"]["vga_warp"]
    assert "S-warp" in wv["config"]["workload"] and wv["config"]["parity_checked_frames"] == 1024 and wv["config"]["parity_mismatches"] == 0
    assert wv["config"]["accepted_match_rate_last_step"] > 0.30 and d["line"]["also_summary"]["vga_warp"]["accepted_match_rate"] > 0.30
    assert d["config"]["accepted_match_rate_last_step"] < 0.05             # S-blocks: independent images
    cb = d["cpu_baseline"]
    assert cb["iterations"] >= 20 and cb["p10_ms"] <= cb["median_ms"] <= cb["p90_ms"] and cb["match_variants"]["popcountll"] > 0
    if os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_orbextractor.so")):
        assert d["cpu_baseline_reference_source"]["kind"] == "reference" and d["cpu_baseline_reference_source"]["value"] > 0
    assert mt["unit"] == "pairs/s" and mt["roofline"]["bound"] == "mfma" and mt["cpu_baseline"]["value"] > 0
    assert mt["roofline"]["per_call_ms"]["calls"] == 30 and mt["roofline"]["per_call_ms"]["min"] <= mt["roofline"]["avg_launch_ms"]
    assert mt["config"]["parity_checked_rows"] >= 50 and mt["config"]["parity_mismatches"] == 0
    assert mt["roofline"]["kernel"] == "k_match_split_mfma4" and mt["roofline"]["peak"] == 10000.0 and mt["dtype"].startswith("fp4")      # the default form (round 5)
    pc, i8 = d["also"]["match100k_popcount"], d["also"]["match100k_int8"]
    assert pc["roofline"]["bound"] == "valu_issue" and pc["roofline"]["kernel"] == "k_match_split" and 0 < pc["value"] < i8["value"] < mt["value"]
    assert i8["roofline"]["bound"] == "mfma" and i8["roofline"]["kernel"] == "k_match_split_mfma" and i8["roofline"]["peak"] == 5000.0 and i8["dtype"].startswith("i8")
    for o in (pc, i8):
        assert o["config"]["parity_mismatches"] == 0 and o["config"]["best_distance_checksum"] == mt["config"]["best_distance_checksum"]


def test_traffic_is_measured_in_the_run():
    """VERDICT r04 weak #1b: `roofline.traffic` of the default command comes from two rocprofv3 --pmc child runs of the serial command made by
    bench.py itself (here forced on a shortened command), not only from the table replayed out of profiles/"""
    d = _bench("--steps", "2", "--warmup", "1", "--min-seconds", "0.2", "--no-also", "--no-cpu-baseline", "--live-traffic", "on")
    rf = d["roofline"]
    assert rf["traffic_source"] == "measured in this run", rf.get("traffic_source")
    # round 6: with the blur computed per keypoint window the description kernel (k_describe_od) and k_fast_cells are the two longest kernels
    assert rf["kernel"] in ("fast_cells", "describe")
    if rf["kernel"] == "fast_cells":
        assert 0.9 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 1.3       # the band staging reads each level once (+ halos)
    else:
        assert 0.3 < rf["traffic"] / rf["algorithmic_bytes_per_launch"] < 2.5       # one 43 x 48-byte window per key point (neighbouring windows share lines in L2) against patch + taps + outputs
    assert d["line"]["roofline"]["traffic"] == rf["traffic"] and d["line"]["roofline"]["traffic_measured_in_this_run"] is True
    per_stage = d["traffic_per_stage_this_run"]
    assert set(per_stage) >= {"pyramid", "fast_cells", "describe", "match"} and d["roofline_pipeline"]["traffic"] == sum(per_stage.values())
    if rf.get("traffic_replayed"):           # the committed table of this build agrees with what the run measured
        assert abs(rf["traffic_replayed"] / rf["traffic"] - 1) < 0.05


def test_single_rank_configs_and_min_duration():
    d = _bench("--config", "vga_extract", "--steps", "2", "--warmup", "1", "--batch", "128", "--ring", "256", "--min-seconds", "0.3", "--no-cpu-baseline")
    _contract(d)
    assert d["n_gpus"] == 1 and "extract @640x480" in d["metric"] and d["repeats"] >= 1 and d["timed_steps"] == d["repeats"] * 2
    assert d["timed_seconds"] >= 0.12          # calibrated from one untimed block: the region may come out a little short of --min-seconds
    m = _bench("--config", "match100k", "--steps", "2", "--warmup", "1", "--min-seconds", "0", "--no-cpu-baseline")
    _contract(m)
    assert m["unit"] == "pairs/s" and m["roofline"]["bound"] == "mfma" and m["value"] > 5e11


def test_rccl_path_at_world_size_one():
    """the `nccl` (= RCCL) backend had never executed — at world size 1 bench.py returned before any collective.
    --force-dist builds the process group anyway: lazy communicator at the first collective, barrier(device_ids), the float64
    all_reduce(MAX) and all_gather on device tensors all run on the GPU exactly as they do at N > 1."""
    d = _bench("--gpus", "1", "--backend", "nccl", "--force-dist", "--steps", "2", "--warmup", "1", "--batch", "128", "--ring", "256", "--min-seconds", "0.2",
               "--no-also", "--no-cpu-baseline")
    _contract(d)
    assert d["n_gpus"] == 1 and len(d["per_rank"]) == 1 and d["per_rank"][0]["device"] == 0 and d["per_rank"][0]["host_submit_ms"] > 0
    assert d["config"]["parity_checked_frames"] == 128 and d["config"]["parity_mismatches"] == 0 and d["repeats"] >= 1
    m = _bench("--gpus", "1", "--backend", "nccl", "--force-dist", "--config", "match100k", "--steps", "2", "--warmup", "1", "--min-seconds", "0", "--no-cpu-baseline")
    assert m["config"]["parity_mismatches"] == 0 and m["value"] > 5e11
