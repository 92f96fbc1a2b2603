#!/usr/bin/env python3
"""bench.py — throughput of the ORB front-end hot path on MI355X.

A step = one pass of the hot path over one batch of synthetic input already resident in HBM.  `--config` names the workload
(BASELINE.json `configs`):
  vga          (default; the configuration BASELINE.json's `metric` is quoted on) 640x480 stream, 8 levels, nFeatures 1000:
               extract (pyramid -> FAST/NMS -> selection -> blur -> orientation + rBRIEF) for `batch` frames, then brute-force
               Hamming top-2 matching of every frame's descriptors against the previous frame's
  vga_extract  configs[1] verbatim: the same stream, extract only
  hd1080       configs[2]: 1920x1080 stream, nFeatures 2000, extract + match
  match100k    configs[4]: one 100,000 x 100,000 top-2 match of 256-bit descriptors per step

A step's `batch` consecutive frames go through `--lanes` (default 4) lanes of batch/lanes consecutive frames, each lane with its
own extractor handle and HIP stream; a lane matches its own frames, and lanes meet only where the frame-to-frame match crosses
a lane border (event-ordered hand-off of one frame's descriptors).  The per-kernel roofline numbers come from a short serial
pass after the timed region (every kernel alone on the chip).

  python bench.py --gpus N --steps K --warmup W
N > 1: one rank per GPU, every rank works on its own stream (weak scaling, no data-path collective); RCCL only takes the MAX of
the timing and gathers the counters.  Run as a plain command it spawns its N ranks itself; under torch.distributed.run it uses
the ranks it is given.  The timed region is `repeats` x K steps, with `repeats` chosen so that it lasts at least
`--min-seconds` (default 6 s, so that a 5 s device sampler must land inside it; 0 = exactly K steps).  Prints ONE JSON line on rank 0.

After the timed region (never inside it) every rank compares the LAST timed step's outputs with the CPU oracle — EVERY frame of
the step (`--parity all`, the default: one oracle instance per host thread; `--parity sample` = lane and step borders + interior
frames only): keypoints + descriptors byte for byte, top-2 match against the previous frame integer for integer — and the line
carries `config.parity_checked_frames` / `config.parity_mismatches`; a mismatch makes every rank exit 1.  The default run
(`--config vga`) then also runs, for >= 1.5 s each with the same parity leg, and embeds under `also` (headline keys unchanged;
`--no-also` skips them): BASELINE.json's other GPU configurations — `vga_extract` (configs[1]), `hd1080` (configs[2]; at N > 1 this
is configs[3], one 1080p stream per GPU), `match100k` (configs[4], FP4 MFMA kernels; `match100k_int8` = the same through the int8 MFMA
kernels of rounds 2-4, `match100k_popcount` = through the xor + popcount kernels the north_star names) — and the headline configuration on the other synthetic
families, whose corner statistics (and therefore FAST / selection cost) differ: `vga_noise` (SURVEY 8d's worst case), `vga_midtex`
(textured: several times more corners at threshold 7 than at 20, no fallback cells), `vga_lowtex` (every cell takes the fallback).
"""
import argparse
import json
import math
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec
HBM_ACHIEVABLE_GBS = 6290.0  # ... 6.29 TB/s measured (float4 copy)
SIMD_CYCLES_PER_S = 256 * 4 * 2.4e9
I8_MFMA_PEAK_TOPS = 5000.0   # dense int8 = 2x the 2.5 PFLOP/s bf16 dense peak (MI355X_MICROARCH.md, matrix-core table)
FP4_MFMA_PEAK_TOPS = 10000.0  # dense FP4 (block-scaled MFMA) = 2x the int8 / FP8 rate (same table: ~10 PF dense, 9.1 measured)

CONFIGS = {
    "vga": dict(width=640, height=480, nfeatures=1000, batch=1024, ring=2048, match=True, baseline_config=1),
    "vga_extract": dict(width=640, height=480, nfeatures=1000, batch=1024, ring=2048, match=False, baseline_config=1),
    "hd1080": dict(width=1920, height=1080, nfeatures=2000, batch=256, ring=512, match=True, baseline_config=2),
    "match100k": dict(n=100000, baseline_config=4),
}


# ------------------------------------------------------------------------------------------------ helpers
def level_sizes(w, h, nlevels=8, sf=1.2):
    import numpy as np
    inv = [np.float32(1.0)]
    isf = np.float32(1.0 / np.float64(np.float32(sf)))
    for _ in range(1, nlevels):
        inv.append(np.float32(inv[-1] * isf))
    return [(int(np.rint(np.float32(w) * s)), int(np.rint(np.float32(h) * s))) for s in inv]


def algorithmic_bytes(w, h, nkp, nlevels=8):
    """per-frame algorithmic bytes of the whole extraction and of each stage."""
    sizes = level_sizes(w, h, nlevels)
    P = [a * b for a, b in sizes]
    p_total, p0 = sum(P), P[0]
    per_stage = {
        "pyramid": sum(P[:-1]) + sum(P[1:]),          # every source level read once, every derived level written once
        "fast_cells": p_total,                        # every pyramid pixel enters the ring test once
        "blur": 2 * p_total,                          # read + write of every level
        "describe": nkp * (749 + 512 + 60),           # patch + BRIEF taps + outputs per keypoint
    }
    a_extract = p_total + (p_total - p0) + 60 * nkp
    a_match = 32 * (nkp + nkp) + 12 * nkp
    return a_extract, a_match, per_stage


def effective_cores():
    """(threads this process may run on, cores the container's CPU quota allows): the GPU boxes expose 256 hardware threads but
    the pod's cgroup caps the CPU time (cpu.max = 1600000 100000 -> 16 cores), which is what an all-core run can use."""
    aff = len(os.sched_getaffinity(0))
    quota = None
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            quota = float(q) / float(p)
    except Exception:
        try:
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    return aff, quota


def host_threads(world=1):
    """host threads one rank may use for its CPU-side legs (frame synthesis, the parity leg's oracle pool): its share of the cores the
    cgroup grants (LOCAL_WORLD_SIZE ranks share a node)"""
    aff, quota = effective_cores()
    cores = max(1, min(aff, int(math.ceil(quota)) if quota else aff))
    local_world = max(1, int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    return max(1, min(32, cores // local_world))


def _cpu_worker(args):
    """One oracle instance (instances are not re-entrant, like the reference's extractor) on its own frames.
    kind "port" = oracle/liborb_oracle.so; "reference" = oracle/_ref/libref_orbextractor.so (the reference's own src/ORBextractor.cc
    compiled where it lies; its OpenCV pixel primitives resolve to the oracle's restatements)."""
    idx, w, h, nfeat, seconds, do_match, kind = args
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from orb_slam_amd import synth
    o = orc.RefExtractor(nfeat) if kind == "reference" else orc.OracleExtractor(nfeatures=nfeat)
    imgs = synth.frames(w, h, synth.BLOCKS, 6000 + 64 * idx, 8)      # synthesis is outside the timed loop
    prev = o(imgs[0])[1]                                              # warm-up frame (page faults), not timed
    done, t, per = 0, time.perf_counter(), []
    while time.perf_counter() - t < seconds:
        t1 = time.perf_counter()
        _, d = o(imgs[(done + 1) % len(imgs)])
        if do_match and len(d) and len(prev):
            orc.match_top2(d, prev)
        prev = d
        done += 1
        per.append(time.perf_counter() - t1)
    return done, time.perf_counter() - t, per


def _percentiles_ms(per):
    import numpy as np
    a = np.sort(np.asarray(per)) * 1e3
    return {"median_ms": round(float(np.median(a)), 4), "p10_ms": round(float(a[int(0.1 * (len(a) - 1))]), 4),
            "p90_ms": round(float(a[int(0.9 * (len(a) - 1))]), 4), "iterations": len(a)}


def cpu_baseline(w, h, nfeat, do_match, seconds, kind="port"):
    """Oracle (scalar CPU restatement of the reference algorithm) timed on this host, 1 core, bounded sample; per-frame median and
    p10 / p90 besides the mean rate.  The reference runs its extractor on the single Tracking thread
    (src/Tracking.cc:199-202), hence cores=1."""
    done, el, per = _cpu_worker((0, w, h, nfeat, seconds, do_match, kind))
    out = {"value": round(done / el, 2), "unit": "frames/s", "cores": 1, "kind": kind,
           "sample": "%d S-blocks %dx%d frames, %s extract (nFeatures %d)%s, %.1f s"
                     % (done, w, h, "oracle" if kind == "port" else "oracle/_ref/libref_orbextractor.so (reference src/ORBextractor.cc)", nfeat,
                        " + scalar top-2 match vs previous frame" if do_match else "", el)}
    out.update(_percentiles_ms(per))
    return out


def cpu_match_variants(nq=32, nt=20000, seconds=1.0):
    """the two scalar matcher baselines of SURVEY.md 8d on one core: the reference's bit-trick popcount (src/ORBmatcher.cc:1794-1810)
    and a __builtin_popcountll variant, pairs/s"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from orb_slam_amd import synth
    Q, T = synth.descriptors(nq, 11), synth.descriptors(nt, 12)
    out = {}
    for name, fn in (("bit_trick", orc.match_top2), ("popcountll", orc.match_top2_popcountll)):
        fn(Q, T)
        n, t = 0, time.perf_counter()
        while time.perf_counter() - t < seconds:
            fn(Q, T)
            n += 1
        out[name] = round(n * float(nq) * nt / (time.perf_counter() - t), 1)
    out["unit"] = "pairs/s"
    out["sample"] = "%d x %d descriptors per call, 1 core, %.0f s each" % (nq, nt, seconds)
    return out


def cpu_baseline_allcores(w, h, nfeat, do_match, seconds):
    """The same on every core this process may use: one oracle instance per worker process, frames sharded."""
    import multiprocessing as mp
    aff, quota = effective_cores()
    workers = max(1, min(aff, int(math.ceil(quota)) if quota else aff))
    with mp.get_context("spawn").Pool(workers) as pool:
        res = pool.map(_cpu_worker, [(i, w, h, nfeat, seconds, do_match, "port") for i in range(workers)])
    res = [(n, t) for n, t, _ in res]
    fps = sum(n / t for n, t in res)
    return {"value": round(fps, 1), "unit": "frames/s", "cores": workers, "kind": "port", "host_threads": aff,
            "cgroup_cpu_quota_cores": quota,
            "sample": "%d worker processes x %.0f s, one oracle extractor each, %d frames in all" % (workers, seconds, sum(n for n, _ in res))}


def load_replayed_counters(build_id, traffic_file="traffic.json"):
    """profiles/traffic.json (rocprofv3 --pmc passes over the serial command) and profiles/valu_mix.json (static opcode mix) carry
    the hash of the kernel sources they were measured on; they are used only when it equals the loaded library's."""
    out = {"traffic": None, "mix": None, "note": None, "clock": None}
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", traffic_file)))
        if tj.get("src_hash") == build_id:
            out["traffic"] = tj
        else:
            out["note"] = "profiles/%s was measured on other kernel sources (hash %s, library %s): not replayed" % (traffic_file, tj.get("src_hash"), build_id)
    except Exception as e:
        out["note"] = "profiles/%s unreadable: %s" % (traffic_file, e)
    try:
        mj = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))
        if mj.get("src_hash") == build_id:
            out["mix"] = mj["kernels"]
    except Exception:
        pass
    try:
        # the shader clock every hot kernel actually holds (tools/run_pmc_clock.sh -> profiles/clock.json): the issue rooflines are priced at
        # the nominal 2.4 GHz AND at that clock; only the long dispatches give a usable figure (values above nominal are dropped)
        cj = json.load(open(os.path.join(ROOT, "profiles", "clock.json")))
        if cj.get("src_hash") == build_id:
            out["clock"] = cj
    except Exception:
        pass
    return out


def measure_live_traffic(frames_per_launch, timeout_s=150.0):
    """HBM-side bytes per launch of every kernel, MEASURED IN THIS RUN (VERDICT r04, weak #1b: the line's `traffic` used to be replayed from
    profiles/traffic.json only): two child runs of the serial command (`--lanes 1 --region-timing`, ORBX_OVERLAP=0: one launch per kernel over
    all frames) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` (separate passes, kernel trace only — the guide's recipe), after every
    timed region of this process.  bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (gfx950: FETCH_SIZE reports half the bytes of a streaming read,
    profiles/r02_fetch_calibration.txt), averaged over the kernel's dispatches; the pyramid stage = 7 k_resize launches.  Returns (per-stage
    bytes, note); (None, why) when rocprofv3 is missing, times out or leaves no counter file — the replayed value then stays in the line."""
    import csv, glob, re, shutil, signal, tempfile
    rp = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(rp):
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) or k == "HSA_TOOLS_LIB" for k in os.environ):
        return None, "this run is itself under a profiler (its environment would reach the child runs)"
    stage = {"k_resize": "pyramid", "k_fast_cells": "fast_cells", "k_quota": "quota", "k_cell_select": "cell_select", "k_level_select": "level_select",
             "k_blur": "blur", "k_blur_mfma": "blur", "k_describe": "describe", "k_describe_od": "describe", "k_match_batch": "match",
             "k_match_batch_mfma": "match"}
    td = tempfile.mkdtemp(prefix="orbx_live_traffic_", dir="/tmp")
    env = {k: v for k, v in os.environ.items()                      # the child runs are plain one-rank commands, whatever launched this one
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "GROUP_RANK", "ROLE_RANK", "MASTER_ADDR", "MASTER_PORT")
           and not k.startswith("TORCHELASTIC_")}
    env.update(ORBX_OVERLAP="0", TMPDIR="/tmp")
    acc, t0 = {}, time.perf_counter()
    try:
        for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
            cmd = [rp, "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", td, "-o", ctr.lower(), "--", sys.executable,
                   os.path.join(ROOT, "bench.py"),
                   "--steps", "3", "--warmup", "1", "--batch", str(frames_per_launch), "--no-cpu-baseline", "--lanes", "1", "--region-timing",
                   "--min-seconds", "0",
                   "--no-also", "--no-parity", "--live-traffic", "off"]
            pr = subprocess.Popen(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, start_new_session=True)
            try:
                pr.wait(timeout=timeout_s)
            except subprocess.TimeoutExpired:
                try:
                    os.killpg(pr.pid, signal.SIGKILL)       # the profiler AND the child it started (its own process group)
                except Exception:
                    pass
                return None, "rocprofv3 --pmc %s pass exceeded %.0f s" % (ctr, timeout_s)
            files = glob.glob(os.path.join(td, "**", "%s_counter_collection.csv" % ctr.lower()), recursive=True)
            if pr.returncode != 0 or not files:
                return None, "rocprofv3 --pmc %s pass: exit %s, %d counter files" % (ctr, pr.returncode, len(files))
            for r in csv.DictReader(open(files[0])):
                m = re.search(r"orbx::(k_[a-z_]+)", r["Kernel_Name"])
                if m and r["Counter_Name"] == ctr:
                    acc.setdefault((m.group(1), ctr), []).append(float(r["Counter_Value"]))
    except Exception as e:                          # never let the measurement aid break the line
        return None, "live traffic pass failed: %r" % (e,)
    finally:
        shutil.rmtree(td, ignore_errors=True)
    out, disp = {}, {}
    for k, st in stage.items():
        f, w = acc.get((k, "FETCH_SIZE")), acc.get((k, "WRITE_SIZE"))
        if not f:
            continue
        n = 7 if k == "k_resize" else 1
        out[st] = int((2.0 * sum(f) / len(f) + (sum(w) / len(w) if w else 0.0)) * 1024 * n)
        disp[st] = len(f)
    if not out:
        return None, "no kernel of the library in the counter files"
    return out, ("HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, in KB) of one launch of this kernel over all %d frames, "
                 "MEASURED IN THIS RUN: two child runs of the "
                 "serial command under rocprofv3 --pmc (FETCH_SIZE, WRITE_SIZE; kernel trace only) after the timed regions, %d dispatches averaged, %.0f s"
                 % (frames_per_launch, disp.get("fast_cells", 0), time.perf_counter() - t0))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """GPUs this process would see, without creating a HIP context in the parent (the ranks are exec'ed, not forked)"""
    import torch
    return torch.cuda.device_count()


def spawn_ranks(n, share_device=False, timeout=1500.0):
    """`python bench.py --gpus N` as a plain command: start the N ranks (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* as
    torch.distributed.run would set them); rank 0 inherits stdout and prints the JSON line.  Refuses to start when the node shows
    fewer than N devices (unless every rank is told to share cuda:0).  A rank that dies takes the others with it after a grace
    period, and the whole run is bounded by `timeout`: a hung rank must not hang the command (rank 0's line, if it was printed,
    is already on stdout)."""
    if not share_device:
        have = visible_gpus()
        if have < n:
            sys.stderr.write("bench.py: --gpus %d but this node shows %d device(s) (HIP_VISIBLE_DEVICES=%s)\n"
                             % (n, have, os.environ.get("HIP_VISIBLE_DEVICES")))
            return 2
    port = free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    t0, rc, failed_at = time.monotonic(), 0, None
    alive = list(procs)
    while alive:
        time.sleep(0.2)
        for p in list(alive):
            c = p.poll()
            if c is not None:
                alive.remove(p)
                rc = max(rc, abs(c))
                if c != 0 and failed_at is None:
                    failed_at = time.monotonic()
        now = time.monotonic()
        if alive and ((failed_at is not None and now - failed_at > 30.0) or now - t0 > timeout):
            sys.stderr.write("bench.py: %s; killing %d remaining rank(s)\n"
                             % ("a rank failed" if failed_at is not None else "no result after %.0f s" % timeout, len(alive)))
            for p in alive:
                p.kill()
            for p in alive:
                p.wait()
            return max(rc, 124 if failed_at is None else rc)
    return rc


# ------------------------------------------------------------------------------------------------ the extraction configs
def run_frontend(a, cfg, world, rank, local_rank, dist, torch):
    from orb_slam_amd import capi, dist_util, synth
    from orb_slam_amd.pipeline import LanePipeline
    dev = torch.device("cuda", local_rank)
    w, h, B, nfeat, do_match = cfg["width"], cfg["height"], cfg["batch"], cfg["nfeatures"], cfg["match"]
    ring = max(cfg["ring"] // B, 1) * B
    # one independent image stream per rank; the ranks of a node share its host cores, so each synthesises on its share of them
    frames = synth.frames(w, h, a.family, dist_util.stream_first_index(rank, ring), ring, threads=host_threads(world))
    d_img = torch.from_numpy(frames).to(dev)
    del frames
    pipe = LanePipeline(w, h, B, lanes=a.lanes, nfeatures=nfeat, device=local_rank, do_match=do_match)   # orb_slam_amd/pipeline.py
    G, b, cap = pipe.G, pipe.b, pipe.cap
    if a.warmup < 1:
        raise SystemExit("--warmup must be >= 1 (first-call costs would land in the timed region)")
    # explicit, blocking stream-placement probe (ORBX_LANE_PLACEMENT=k skips it), outside every timed loop; the ranks of a node probe one
    # after the other (they share host cores, and with --share-device the GPU): a probe timed beside another rank's probe measures the neighbour
    for r in range(world):
        if r == rank:
            pipe.tune(d_img.data_ptr())
        if world > 1:
            dist_util.barrier(dist, a.backend, local_rank)

    def step(i, timed):
        pipe.step(d_img.data_ptr() + ((i * B) % ring) * w * h, timed=timed)

    for i in range(a.warmup):
        step(i, False)
    torch.cuda.synchronize(dev)
    # host cost of queueing one step (4 lanes x ~15 launches + the hand-off copies and events): on an idle queue, i.e. not throttled by
    # the device — what one rank asks of its share of the host cores
    th = time.perf_counter()
    for i in range(2):
        step(a.warmup + i, False)
    host_submit_ms = (time.perf_counter() - th) * 500.0
    torch.cuda.synchronize(dev)
    # repeats of the K-step block so that the timed region lasts >= --min-seconds: one untimed calibration block tells how long a
    # block takes (the warm-up steps carry first-call costs); every rank must run the same count
    repeats = 1
    if a.min_seconds > 0:
        tc = time.perf_counter()
        for i in range(a.steps):
            step(a.warmup + 2 + i, False)
        torch.cuda.synchronize(dev)
        tc = time.perf_counter() - tc
        repeats = max(1, int(math.ceil(a.min_seconds / max(tc, 1e-6))))
    repeats = int(dist_util.agree_max(dist, repeats, dev if a.backend == "nccl" else torch.device("cpu")))
    nsteps = repeats * a.steps
    first = a.warmup + 2 + (a.steps if a.min_seconds > 0 else 0)      # index of the first timed step (the stream just continues)
    pipe.stage_timing(2 if a.region_timing else 0)
    dist_util.barrier(dist, a.backend, local_rank)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(nsteps):
        step(first + i, a.region_timing)
    torch.cuda.synchronize(dev)
    dist_util.barrier(dist, a.backend, local_rank)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    # parity leg (outside the timed region): the outputs the LAST timed step left on the device against the CPU oracle
    parity = {"frames": 0, "mismatches": 0, "detail": []}
    t_par = time.perf_counter()
    if a.parity != "none":
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import parity_sample
        last = first + nsteps - 1
        i0 = dist_util.stream_first_index(rank, ring)

        def host_frame(j):
            return synth.frame(w, h, a.family, i0 + (last * B + j) % ring)

        sample = range(B) if a.parity == "all" else parity_sample.sample_indices(B, G)
        parity = parity_sample.check_step(pipe, host_frame, sample, nfeat, threads=host_threads(world))
    t_par = time.perf_counter() - t_par

    stage = pipe.stage_times()
    pipe.stage_timing(0)
    kp_mean = float(pipe.counts().float().mean().item())
    bad_status = int((pipe.status() != 0).sum().item())
    accepted, accepted_rate = -1, None
    if do_match:
        last = pipe.lanes[G - 1]
        best = last.match[1, b - 1, :cap].cpu().numpy()
        sec = last.match[2, b - 1, :cap].cpu().numpy()
        nq = int(last.n[b].item())
        accepted = capi.count_accepted(best[:nq], sec[:nq], 50, 0.6)
        # the accept rule of SearchByBoW (src/ORBmatcher.cc:224-226: best <= TH_LOW && (float)best < 0.6f * (float)second) over EVERY frame of the last
        # step: what fraction of the key points finds its partner in the previous frame (S-warp: a camera-like stream; S-blocks: independent images)
        acc_n = acc_q = 0
        for ln in pipe.lanes:
            nn = ln.n[1:b + 1].to(torch.int64)
            live = torch.arange(cap, device=dev)[None, :] < nn[:, None]
            bb, ss = ln.match[1, :b, :cap], ln.match[2, :b, :cap]
            ok = live & (bb <= 50) & (bb.to(torch.float32) < torch.tensor(0.6, dtype=torch.float32, device=dev) * ss.to(torch.float32))
            acc_n += int(ok.sum().item()); acc_q += int(nn.sum().item())
        accepted_rate = round(acc_n / max(acc_q, 1), 4)

    def serial_pass(nser):
        """The same step with every kernel alone on the chip: one extractor over all B frames, one launch per kernel, the match
        on the same stream.  Per-kernel roofline numbers come from here; in the timed region above the lanes co-run, so a
        kernel's duration there includes sharing the CUs with the kernels of the other lanes."""
        keep = os.environ.get("ORBX_OVERLAP")
        os.environ["ORBX_OVERLAP"] = "0"          # read by orbx_create: no blur side stream either, every kernel alone on the chip
        try:
            ex1 = capi.ORBextractor(nfeatures=nfeat, device=local_rank, max_batch=B)
        finally:
            if keep is None:
                del os.environ["ORBX_OVERLAP"]
            else:
                os.environ["ORBX_OVERLAP"] = keep
        main = torch.cuda.current_stream(dev)
        kps1 = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
        desc1 = torch.zeros((B + 1, cap, 32), dtype=torch.uint8, device=dev)
        n1 = torch.zeros(B + 1, dtype=torch.int32, device=dev)
        match1 = torch.zeros((3, B, cap), dtype=torch.int32, device=dev)
        evs = []
        for i in range(2 + nser):
            if i == 2:
                torch.cuda.synchronize(dev)
                ex1.stage_timing(2)
            f0 = (i * B) % ring
            ex1.extract_batch_device(d_img.data_ptr() + f0 * w * h, B, w, h, w, w * h, kps1.data_ptr(), desc1[1].data_ptr(), n1[1:].data_ptr(), cap, 0,
                                     main.cuda_stream)
            if do_match:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(main)
                capi.match_top2_batch_device(desc1[1].data_ptr(), n1[1:].data_ptr(), desc1[0].data_ptr(), n1.data_ptr(), B, cap,
                                             match1[0].data_ptr(), match1[1].data_ptr(), match1[2].data_ptr(), main.cuda_stream)
                e1.record(main)
                if i >= 2:
                    evs.append((e0, e1))
                desc1[0].copy_(desc1[B], non_blocking=True)
                n1[0:1].copy_(n1[B:B + 1], non_blocking=True)
        torch.cuda.synchronize(dev)
        st1 = ex1.stage_times()
        ex1.stage_timing(0)
        ms1 = {k: (ms / n if n else 0.0) for k, (ms, n) in st1.items()}
        if do_match:
            ms1["match"] = sum(e0.elapsed_time(e1) for e0, e1 in evs) / max(len(evs), 1)
        ex1.close()
        return ms1

    # RCCL: the only collectives of the run (MAX of the timing, all-gather of the counters)
    ni = getattr(a, "numa_info", None) or {}
    tmax, counters, rows = dist_util.reduce_run(dist, elapsed, [nsteps * B, kp_mean * nsteps * B, bad_status, elapsed, parity["frames"],
                                                                parity["mismatches"], host_submit_ms, float(torch.cuda.current_device()),
                                                                float(ni.get("pci_bus", -1)), float(ni.get("numa_node", -1)), float(ni.get("host_threads", 0)),
                                                                float(ni.get("bound", 0))],
                                                dev if a.backend == "nccl" else torch.device("cpu"))
    placement = pipe.placement
    if rank != 0:
        pipe.close()
        return {"_mismatches": int(counters[5])}
    total_frames = float(counters[0])
    a_extract, a_match, per_stage = algorithmic_bytes(w, h, nfeat)
    region_ms = {k: (ms / n if n else 0.0) for k, (ms, n) in stage.items()}      # per LAUNCH: one lane's slice of b frames
    concurrent = G > 1 or not a.region_timing        # without in-region timing the serial pass is the only per-kernel timing
    stage_ms = serial_pass(10) if concurrent else dict(region_ms)
    dom = max(stage_ms, key=lambda k: stage_ms[k])
    dom_frame_bytes = per_stage.get(dom, a_match if dom == "match" else 0)
    dom_bytes = dom_frame_bytes * B
    dom_gbs = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
    kernel_ms = sum(stage_ms.values())
    pipe_bytes = (a_extract + (a_match if do_match else 0)) * B
    step_ms = tmax / nsteps * 1e3
    pipe_gbs = pipe_bytes / (step_ms * 1e-3) / 1e9
    value = total_frames / tmax

    # counters replayed from profiles/ (PMC passes cannot run inside this process): only when measured on THIS build
    # the two workloads with committed counter passes: VGA / 1000 (profiles/traffic.json) and 1080p / 2000 (profiles/traffic_hd1080.json)
    wl_tag = {(640, 480, 1000, 1): "vga_640x480_nf1000", (1920, 1080, 2000, 1): "hd_1920x1080_nf2000"}.get((w, h, nfeat, a.family)) if do_match else None
    traffic_file = "traffic_hd1080.json" if wl_tag == "hd_1920x1080_nf2000" else "traffic.json"
    rep = load_replayed_counters(capi.build_id(), traffic_file)
    traffic, valu_insts, valu_launch = None, None, None
    tj = rep["traffic"]
    same_workload = tj is not None and wl_tag is not None and tj.get("workload") == wl_tag
    if same_workload:
        if tj.get("batch") == B:
            traffic = tj.get("per_launch_bytes", {}).get(dom)
        pf = tj.get("valu_wave_insts_per_frame", {})
        if pf and (not do_match or "match" in pf):
            valu_insts = {k: v for k, v in pf.items() if do_match or k != "match"}
            valu_launch = pf.get(dom, 0) * B
    hbm = {"bound": "hbm", "achieved": round(dom_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(dom_gbs / HBM_PEAK_GBS, 5),
           "frac_of_achievable": round(dom_gbs / HBM_ACHIEVABLE_GBS, 5), "achievable_peak": HBM_ACHIEVABLE_GBS, "traffic": traffic,
           "algorithmic_bytes_per_launch": dom_bytes,
           "traffic_note": "HBM-side bytes (2 x FETCH_SIZE + WRITE_SIZE, in KB) of one launch of this kernel over all %d frames, rocprofv3 --pmc passes "
                           "over the serial command (profiles/%s; replayed only when its source hash equals the library's)" % (B, traffic_file)
                           if traffic else rep["note"]}
    timing = ("serial pass after the timed region: 10 steps, one launch per kernel over all %d frames, nothing else on the chip "
              "(HIP events on the launch stream)" % B) if concurrent else "timed region (one stream)"
    # `roofline` follows the contract: the dominant kernel's ALGORITHMIC bytes per launch / its average launch
    # duration against the HBM peak, `traffic` = PMC-measured HBM bytes per launch.  The resource that actually binds this integer
    # kernel is VALU issue (SURVEY.md §8d predicted it, the counters confirm it); that model rides along as roofline.valu_issue.
    roofline = dict(hbm)
    roofline.update({"kernel": dom, "avg_launch_ms": round(stage_ms[dom], 4), "frames_per_launch": B, "timing": timing})
    mix = rep["mix"]
    if mix and wl_tag == "hd_1920x1080_nf2000" and "blur_valu" in mix:
        mix = dict(mix, blur=mix["blur_valu"])         # levels wider than 1024 px keep the VALU form of the blur (k_blur), VGA-class ones run k_blur_mfma
        if "fast_cells_large" in mix:
            mix["fast_cells"] = mix["fast_cells_large"]    # ... and the large launch shape of k_fast_cells (two dwords per lane and round)
    stage_kernel = {"pyramid": "k_resize", "fast_cells": "k_fast_cells", "blur": "k_blur" if wl_tag == "hd_1920x1080_nf2000" else "k_blur_mfma",
                    "describe": "k_describe", "match": "k_match_batch_mfma4", "cell_select": "k_cell_select", "level_select": "k_level_select",
                    "quota": "k_quota"}
    clocks = (rep["clock"] or {}).get("hd" if wl_tag == "hd_1920x1080_nf2000" else "vga", {}) if wl_tag else {}

    def clock_ghz(stage):          # measured shader clock under this kernel, GHz (nominal 2.4 where there is no usable measurement)
        c = clocks.get(stage_kernel.get(stage, ""), 2.4)
        return c if 1.0 < c <= 2.4 else 2.4
    if valu_launch and mix and dom in mix:
        cpi = mix[dom]["cycles_per_inst_lo"]
        ach = valu_launch / (stage_ms[dom] * 1e-3)
        peak = SIMD_CYCLES_PER_S / cpi
        roofline["valu_issue"] = {
            "bound": "valu_issue", "kernel": dom, "achieved": round(ach / 1e9, 2), "peak": round(peak / 1e9, 2), "unit": "G wave-insts/s",
            "frac": round(ach / peak, 4), "clock_ghz": clock_ghz(dom) if clocks else None,
            "frac_at_clock": round(ach / (peak * clock_ghz(dom) / 2.4), 4) if clocks else None,
            "wave_insts_per_launch": int(valu_launch), "cycles_per_inst": cpi,
            "pricing": "SQ_INSTS_VALU of the kernel (profiles/%s) priced with its static opcode mix (profiles/valu_mix.json): 2 cycles per "
                       "wave64 instruction for mov/add/sub/and/or/xor/bitop3/right shifts/f32 add-mul-fma, 4 for every other measured opcode "
                       "(profiles/r01_valu_issue_rates.txt, r02_valu_issue_rates2.txt)" % traffic_file}
    out = {
        "metric": "frames/s ORB %s @%dx%d, %d kp" % ("extract+match" if do_match else "extract", w, h, nfeat),
        "value": round(value, 1),
        "unit": "frames/s",
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "repeats": repeats,
        "timed_steps": nsteps,
        "timed_seconds": round(tmax, 3),
        "ms_per_step": round(step_ms, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic",
        "config": {"workload": "%s: %dx%d grayscale stream, 8 levels, nFeatures %d, %s frames, extract%s (BASELINE.json configs[%d]%s)" % (
                       a.config, w, h, nfeat, synth.FAMILY_NAMES.get(a.family, str(a.family)),
                       " + Hamming top-2 match vs previous frame" if do_match else " only", cfg["baseline_config"],
                       " + the frame-to-frame match of the metric" if a.config == "vga" else ""),
                   "frames_per_step_per_gpu": B, "resident_frames_per_gpu": ring,
                   "parallelism": "one image stream per GPU; a step's %d frames go through %d lanes of %d consecutive frames "
                                  "(own extractor handle + HIP stream each), frame-to-frame matches across lane borders via event-ordered hand-off" % (B, G, b),
                   "lanes": G, "lane_placement": placement,
                   "parity_checked_frames": int(counters[4]), "parity_mismatches": int(counters[5]),
                   "parity_note": "outputs of the LAST timed step vs the CPU oracle, after the timed region, on every rank: %s; keypoints + descriptors "
                                  "byte-equal, top-2 match vs the previous frame integer-equal (%.1f s on %d host threads)"
                                  % ("EVERY frame of the step" if a.parity == "all" else "first + last frame of every lane, the step border, interior frames",
                                     t_par, host_threads(world)),
                   "parity_detail": parity["detail"],
                   "host_submit_ms_per_step": round(float(counters[6]) / world, 4),
                   "mean_keypoints_per_frame": round(float(counters[1]) / total_frames, 2),
                   "frames_with_error_status": int(counters[2]), "accepted_matches_last_frame": accepted, "accepted_match_rate_last_step": accepted_rate,
                   "accepted_note": "best <= 50 && best < 0.6 * second over every frame of the last timed step.  Consecutive S-blocks (noise, lowtex, "
                                    "midtex) frames are "
                                    "independent images, so few matches pass; S-warp (--family 5, `also.vga_warp`) is a correlated stream: "
                                    "consecutive frames show the "
                                    "same corners a pixel or two apart.  What the match leg computes is checked against the oracle in the parity leg "
                                    "(integer-equal top-2)",
                   "library_build_id": capi.build_id()},
        "per_rank": [{"rank": r, "device": int(row[7]), "frames": int(row[0]), "elapsed_s": round(row[3], 4), "frames_per_s": round(row[0] / row[3], 1),
                      "host_submit_ms": round(row[6], 4), "parity_checked_frames": int(row[4]),
                      "pci_bus": int(row[8]), "numa_node": int(row[9]), "host_threads": int(row[10]), "numa_bound": int(row[11])}
                     for r, row in enumerate(rows)],
        "roofline": roofline,
        "roofline_pipeline": {"bound": "hbm", "achieved": round(pipe_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(pipe_gbs / HBM_PEAK_GBS, 5), "frac_of_achievable": round(pipe_gbs / HBM_ACHIEVABLE_GBS, 5),
                              "algorithmic_bytes_per_step": pipe_bytes, "ms_per_step": round(step_ms, 4), "kernel_ms_per_step_serial": round(kernel_ms, 4)},
        "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
    }
    if valu_insts and mix and all(k in mix for k in valu_insts):
        n_inst = sum(valu_insts.values())
        ach = value / world * n_inst
        lo = sum(v * mix[k]["cycles_per_inst_lo"] for k, v in valu_insts.items())
        hi = sum(v * mix[k]["cycles_per_inst_hi"] for k, v in valu_insts.items())
        out["roofline_valu"] = {
            "bound": "valu_issue", "achieved": round(ach / 1e9, 2), "unit": "G wave-insts/s", "wave_insts_per_frame": round(n_inst),
            "peak": round(SIMD_CYCLES_PER_S / (lo / n_inst) / 1e9, 2), "frac": round(value / world * lo / SIMD_CYCLES_PER_S, 4),
            "frac_range": [round(value / world * lo / SIMD_CYCLES_PER_S, 4), round(value / world * hi / SIMD_CYCLES_PER_S, 4)],
            "cycles_per_inst_range": [round(lo / n_inst, 3), round(hi / n_inst, 3)],
            "frac_at_clock": (round(value / world * sum(v * mix[k]["cycles_per_inst_lo"] / (256 * 4 * clock_ghz(k) * 1e9)
                                                           for k, v in valu_insts.items()), 4) if clocks else None),
            "clock_note": "frac prices every kernel's instructions at the nominal 2.4 GHz; frac_at_clock at the clock measured under that kernel "
                          "(profiles/clock.json = tools/run_pmc_clock.sh on this build)" if clocks else None,
            "source": "whole step in the timed region: SQ_INSTS_VALU of every kernel per frame (profiles/traffic.json, same source hash as the library) "
                      "x measured frames/s per GPU; the match kernel's MFMA work is not VALU and not counted"}
    elif rep["note"]:
        out["roofline_valu"] = None
        out["roofline_note"] = rep["note"]
    if a.region_timing:
        out["stage_ms_per_launch_timed_region"] = {k: round(v, 4) for k, v in region_ms.items()}
    pipe.close()
    del d_img
    torch.cuda.empty_cache()
    if world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(w, h, nfeat, do_match, a.cpu_seconds)
        if do_match:
            out["cpu_baseline"]["match_variants"] = cpu_match_variants()
        if a.cpu_reference_seconds > 0 and os.path.exists(os.path.join(ROOT, "oracle", "_ref", "libref_orbextractor.so")):
            # the north_star's literal wording ("next to the reference src/ORBextractor.cc"): the reference's own translation unit, prebuilt
            # by oracle/Makefile where /root/reference exists and shipped with the tree; same arithmetic as the port
            out["cpu_baseline_reference_source"] = cpu_baseline(w, h, nfeat, do_match, a.cpu_reference_seconds, kind="reference")
        if a.cpu_allcores_seconds > 0:
            out["cpu_baseline_allcores"] = cpu_baseline_allcores(w, h, nfeat, do_match, a.cpu_allcores_seconds)
    return out


# ------------------------------------------------------------------------------------------------ config 5: 100k x 100k top-2
def run_match(a, cfg, world, rank, local_rank, dist, torch):
    """BASELINE.json configs[4].  N > 1: queries sharded by rank, train set replicated — every rank matches its
    own n/world queries against all n train descriptors; no exchange of partial results."""
    from orb_slam_amd import capi, dist_util, synth
    dev = torch.device("cuda", local_rank)
    n = cfg["n"]
    nq = (n + world - 1) // world
    q0 = rank * nq
    nq = max(0, min(nq, n - q0))
    # -1: the process default (the FP4 MFMA kernels); 0: the xor + popcount kernels north_star names; 1: int8 MFMA
    capi.set_match_path(getattr(a, "match_path", -1))
    path = capi.get_match_path()                               # the kernels in effect (environment default or the forced path): 0 / 1 / 2
    Qall = synth.descriptors(n, 1)
    Q = torch.from_numpy(Qall[q0:q0 + nq].copy()).to(dev)
    T = torch.from_numpy(synth.descriptors(n, 2)).to(dev)
    out3 = torch.zeros((3, max(nq, 1)), dtype=torch.int32, device=dev)
    s = torch.cuda.current_stream(dev)

    def step():
        capi.match_top2_device(Q.data_ptr(), nq, T.data_ptr(), n, out3[0].data_ptr(), out3[1].data_ptr(), out3[2].data_ptr(), s.cuda_stream)

    for _ in range(a.warmup):
        step()
    torch.cuda.synchronize(dev)
    repeats = 1
    if a.min_seconds > 0:
        tc = time.perf_counter()
        for _ in range(a.steps):
            step()
        torch.cuda.synchronize(dev)
        tc = time.perf_counter() - tc
        repeats = max(1, int(math.ceil(a.min_seconds / max(tc, 1e-6))))
    repeats = int(dist_util.agree_max(dist, repeats, dev if a.backend == "nccl" else torch.device("cpu")))
    nsteps = repeats * a.steps
    dist_util.barrier(dist, a.backend, local_rank)
    torch.cuda.synchronize(dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(s)
    for _ in range(nsteps):
        step()
    e1.record(s)
    torch.cuda.synchronize(dev)
    dist_util.barrier(dist, a.backend, local_rank)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    region_ms = e0.elapsed_time(e1) / nsteps            # split + merge kernels of one call, averaged over the timed region (calls back to back)
    # steady-state duration of ONE call: 30 further calls, each between its own pair of HIP events on the launch stream
    # (the r02 rocprof summary held 7 cold calls and disagreed with the region average; min / median / max are reported,
    # the roofline uses the MEDIAN)
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for ea, eb in evs:
        ea.record(s)
        step()
        eb.record(s)
    torch.cuda.synchronize(dev)
    per_call = sorted(ea.elapsed_time(eb) for ea, eb in evs)
    kernel_ms = per_call[len(per_call) // 2]
    checksum = int(out3[1, :nq].sum().item()) if nq else 0
    capi.set_match_path(-1)
    # parity leg (outside the timed region): sampled query rows of the last call against the oracle's sequential scan
    checked, mism = 0, 0
    if a.parity != "none" and nq:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as orc
        import numpy as np
        rows_ = np.unique(np.linspace(0, nq - 1, 64).astype(np.int64))
        got = out3[:, :nq].cpu().numpy()[:, rows_]
        Tn_ = synth.descriptors(n, 2)
        ri, rb, rs = orc.match_top2(np.ascontiguousarray(Qall[q0:q0 + nq][rows_]), Tn_)
        checked = len(rows_)
        mism = int((got[0] != ri).sum() + (got[1] != rb).sum() + (got[2] != rs).sum())
    tmax, counters, rows = dist_util.reduce_run(dist, elapsed, [float(nq) * n * nsteps, checksum, elapsed, checked, mism],
                                                dev if a.backend == "nccl" else torch.device("cpu"))
    if rank != 0:
        return {"_mismatches": int(counters[4])}
    pairs = float(counters[0])
    value = pairs / tmax
    a_match = 32 * (nq + n) + 12 * nq
    gbs = a_match / (kernel_ms * 1e-3) / 1e9
    tops = 2.0 * 256.0 * nq * n / (kernel_ms * 1e-3) / 1e12
    mfma, fp4 = path != 0, path == 2
    if not mfma:
        # the popcount form: 8 x v_xor + 8 x v_bcnt per pair (accumulating form) = 16 lane-operations -> 7.9e13 / 16 = 4.9e12 pairs/s
        pps = float(nq) * n / (kernel_ms * 1e-3)
        tops, peak_override = pps / 1e12, 4.9
    else:
        peak_override = None
    mfma_peak = FP4_MFMA_PEAK_TOPS if fp4 else I8_MFMA_PEAK_TOPS
    roofline = {"bound": "mfma" if mfma else "valu_issue", "kernel": ("k_match_split_mfma4" if fp4 else "k_match_split_mfma") if mfma else "k_match_split",
                "achieved": round(tops, 3 if peak_override else 1), "peak": peak_override or mfma_peak,
                "unit": ("TOP/s (FP4 multiply-accumulates x 2)" if fp4 else "TOP/s (int8 multiply-accumulates x 2)") if mfma
                        else "10^12 pairs/s (16 lane-operations per pair: 8 v_xor + 8 v_bcnt)",
                "frac": round(tops / (peak_override or mfma_peak), 4),
                "peak_note": ("dense FP4 through v_mfma_scale_f32_32x32x64_f8f6f4 = 2 x the int8 / FP8 dense peak (the guide measures 9.1 POP/s); the "
                              "same call "
                              "through the int8 kernels (ORBX_MATCH_MFMA=8) is priced against 5 POP/s" if fp4 else
                              "dense int8 MFMA = 2 x the 2.5 PFLOP/s bf16 dense peak; tools/microbench/valu_rate2 measures 4470 TOP/s for "
                              "v_mfma_i32_32x32x32_i8"),
                "avg_launch_ms": round(kernel_ms, 4), "pairs_per_launch": float(nq) * n,
                "per_call_ms": {"min": round(per_call[0], 4), "median": round(kernel_ms, 4), "max": round(per_call[-1], 4), "calls": len(per_call),
                                "timed_region_average": round(region_ms, 4)},
                "timing": "median of 30 steady-state calls after the timed region, each between its own HIP events on the launch stream (split + "
                          "merge kernel per call); timed_region_average = one event pair around the %d back-to-back calls of the timed region" % nsteps,
                "hbm": {"bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 6),
                        "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 6), "algorithmic_bytes_per_launch": a_match, "traffic": None},
                "traffic": None}
    if not mfma:
        # what the xor + popcount form can issue at best on THIS chip: v_xor_b32 costs 2.61 and v_bcnt_u32_b32 4.08 cycles per wave64
        # instruction per SIMD (profiles/r01_valu_issue_rates.txt), so the 8 + 8 instructions of 64 pairs take 53.5 cycles before the
        # top-2 bookkeeping: 1024 SIMDs x 64 / 53.5 x 2.4 GHz.  (SURVEY 8d's 4.9e12 assumes one lane-operation per lane and clock.)
        ceil = 1024 * 64 / (8 * 2.61 + 8 * 4.08) * 2.4e9 / 1e12
        roofline["measured_issue_ceiling"] = {"peak": round(ceil, 3), "unit": "10^12 pairs/s", "frac": round(tops / ceil, 4),
                                              "note": "8 v_xor (2.61 cycles) + 8 v_bcnt (4.08 cycles) per 64 pairs and SIMD at 2.4 GHz, top-2 "
                              "bookkeeping not counted"}
    out = {
        "metric": "pairs/s Hamming top-2, %d x %d 256-bit descriptors" % (n, n),
        "value": round(value, 1), "unit": "pairs/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "repeats": repeats, "timed_steps": nsteps,
        "timed_seconds": round(tmax, 3), "ms_per_step": round(tmax / nsteps * 1e3, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None,
        "dtype": ("fp4 e2m1 (+-1 encoded bits, f32 accumulate, exact)" if fp4 else "i8 (+-1 encoded bits, i32 accumulate)") if mfma else "u32 xor + popcount",
        "data": "synthetic",
        "config": {"workload": "match100k: batched N-to-M descriptor match, %d x %d random 256-bit descriptors, dense top-2 (BASELINE.json "
                               "configs[4])" % (n, n),
                   "queries_per_gpu": nq, "train_descriptors": n, "parallelism": "queries sharded by rank, train set replicated, no exchange",
                   "best_distance_checksum": int(counters[1]), "parity_checked_rows": int(counters[3]), "parity_mismatches": int(counters[4]),
                   "parity_note": "64 evenly spaced query rows per rank of the last call vs the oracle's sequential scan (index, best, second)",
                   "library_build_id": capi.build_id()},
        "per_rank": [{"rank": r, "pairs": row[0], "elapsed_s": round(row[2], 4)} for r, row in enumerate(rows)],
        "roofline": roofline,
    }
    if world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as orc
        Tn = synth.descriptors(n, 2)
        done, t = 0, time.perf_counter()
        blk = 64
        while time.perf_counter() - t < a.cpu_seconds:
            orc.match_top2(Qall[done:done + blk], Tn)
            done += blk
        el = time.perf_counter() - t
        out["cpu_baseline"] = {"value": round(done * float(n) / el, 1), "unit": "pairs/s", "cores": 1, "kind": "port",
                               "sample": "%d queries x %d train descriptors, oracle scalar top-2 scan (popcount per word), %.1f s" % (done, n, el)}
    return out


# ------------------------------------------------------------------------------------------------ the line the driver parses
DETAIL_PREFIX = "#detail "      # full reports: one stdout line each, in front of the final line, never starting with "{"


def _pick(d, keys):
    return {k: d[k] for k in keys if isinstance(d, dict) and k in d}


def compact_line(full, also=None):
    """The LAST stdout line: the contract's headline keys + roofline + cpu_baseline, and one short row per other configuration
    (`also_summary`).  Everything else (notes, per-call timings, parity detail, the full `also` reports) is printed in front of it as
    `#detail <name> <json>` lines.  Bounded: <= 4 KB at one rank, <= 6 KB at eight (the 23 KB line of round 4 did not
    parse in the driver's record); tests/test_bench_line.py asserts it on CPU, tests/test_gpu_bench.py on the GPU."""
    out = _pick(full, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "repeats", "timed_steps", "timed_seconds", "ms_per_step",
                       "higher_is_better", "scaling", "vs_baseline", "dtype", "data"))
    cfg = full.get("config", {})
    c = _pick(cfg, ("workload", "frames_per_step_per_gpu", "resident_frames_per_gpu", "lanes", "queries_per_gpu", "train_descriptors",
                    "parity_checked_frames", "parity_checked_rows", "parity_mismatches", "mean_keypoints_per_frame", "frames_with_error_status",
                    "host_submit_ms_per_step", "best_distance_checksum", "library_build_id"))
    if isinstance(cfg.get("lane_placement"), dict):
        c["lane_placement"] = cfg["lane_placement"].get("chosen")
    out["config"] = c
    r = full.get("roofline") or {}
    ro = _pick(r, ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
                   "frames_per_launch", "pairs_per_launch", "frac_of_achievable"))
    ro.setdefault("traffic", None)
    if "traffic_source" in r:
        ro["traffic_measured_in_this_run"] = r["traffic_source"] == "measured in this run"
    if isinstance(r.get("valu_issue"), dict):
        ro["valu_issue"] = _pick(r["valu_issue"], ("achieved", "peak", "unit", "frac", "clock_ghz", "frac_at_clock"))
    if isinstance(r.get("hbm"), dict):
        ro["hbm"] = _pick(r["hbm"], ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch"))
    out["roofline"] = ro
    if isinstance(full.get("roofline_pipeline"), dict):
        out["roofline_pipeline"] = _pick(full["roofline_pipeline"], ("bound", "achieved", "peak", "unit", "frac", "kernel_ms_per_step_serial", "traffic"))
    if isinstance(full.get("roofline_valu"), dict):
        out["roofline_valu"] = _pick(full["roofline_valu"], ("bound", "achieved", "peak", "unit", "frac", "frac_at_clock"))
    if "stage_ms_per_step" in full:
        out["stage_ms_per_step"] = full["stage_ms_per_step"]
    for k in ("cpu_baseline", "cpu_baseline_reference_source", "cpu_baseline_allcores"):
        if isinstance(full.get(k), dict):
            out[k] = _pick(full[k], ("value", "unit", "cores", "kind", "sample", "median_ms") if k == "cpu_baseline" else ("value", "unit", "cores", "kind"))
    if full.get("per_rank"):
        keys = [k for k in ("rank", "device", "pci_bus", "numa_node", "host_threads", "numa_bound", "frames", "pairs", "elapsed_s", "parity_checked_frames")
                if k in full["per_rank"][0]]
        out["per_rank"] = {"columns": keys, "rows": [[row.get(k) for k in keys] for row in full["per_rank"]]}
    if also:
        summ = {}
        for name, rpt in also.items():
            row = _pick(rpt, ("value", "unit", "ms_per_step"))          # (timed_seconds is in the #detail report: the line stays well under 4 KB)
            row["parity_mismatches"] = rpt.get("config", {}).get("parity_mismatches")
            if rpt.get("config", {}).get("accepted_match_rate_last_step") is not None and name == "vga_warp":
                row["accepted_match_rate"] = rpt["config"]["accepted_match_rate_last_step"]
            rr = rpt.get("roofline") or {}
            row["roofline_frac"] = rr.get("frac")
            if rr.get("bound") != "hbm":            # (rows without the key: bound "hbm", like the headline's roofline)
                row["roofline_bound"] = rr.get("bound")
            if isinstance(rpt.get("cpu_baseline"), dict):
                row["cpu_baseline"] = rpt["cpu_baseline"].get("value")
            summ[name] = row
        out["also_summary"] = summ
    out["detail"] = "full reports: the '%s<name> <json>' stdout lines in front of this line" % DETAIL_PREFIX
    return out


def print_detail(name, report):
    print(DETAIL_PREFIX + name + " " + json.dumps(report), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="vga", choices=sorted(CONFIGS), help="BASELINE.json workload (see the module docstring)")
    ap.add_argument("--batch", type=int, default=None, help="frames per step per GPU (default: the config's)")
    ap.add_argument("--ring", type=int, default=None, help="distinct frames resident per GPU (>= 1024 VGA frames exceeds the 256 MiB Infinity Cache)")
    ap.add_argument("--width", type=int, default=None)
    ap.add_argument("--height", type=int, default=None)
    ap.add_argument("--nfeatures", type=int, default=None)
    ap.add_argument("--family", type=int, default=1,
                    help="0 noise, 1 blocks (default), 3 lowtex, 4 midtex, 5 warp = correlated stream (orb_slam_amd/csrc/synth_frames.c)")
    ap.add_argument("--no-match", action="store_true", help="extract only (same as --config vga_extract for the VGA stream)")
    ap.add_argument("--lanes", type=int, default=4,
                    help="a step's frames go through this many concurrent lanes (own extractor handle + HIP stream each); 1 = one stream")
    ap.add_argument("--region-timing", action="store_true",
                    help="also time every kernel inside the timed region (HIP events between the kernels of every lane: costs a few percent)")
    ap.add_argument("--min-seconds", type=float, default=6.0,
                    help="repeat the --steps block until the timed region lasts at least this long (0: exactly --steps steps)")
    ap.add_argument("--parity", default=None, choices=["all", "sample", "none"],
                    help="oracle comparison of the LAST timed step's outputs: every frame (default at one rank), lane / step borders + interior "
                         "frames (default at N > 1: the ranks of a node share its host cores), or none")
    ap.add_argument("--no-parity", action="store_true", help="same as --parity none")
    ap.add_argument("--cpu-reference-seconds", type=float, default=4.0,
                    help="CPU sample of oracle/_ref/libref_orbextractor.so (the reference's own ORBextractor.cc), when that file is present; 0 disables")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed even at world size 1 (the RCCL communicator, barrier and collectives then run as at N > 1)")
    ap.add_argument("--no-also", action="store_true",
                    help="headline configuration only (the default run also measures vga_extract, hd1080, match100k and the other frame families)")
    ap.add_argument("--also-min-seconds", type=float, default=6.0,
                    help="timed region of each other frame configuration (as long as the headline's: sustained clocks)")
    ap.add_argument("--also-match-min-seconds", type=float, default=2.0, help="timed region of the 100k x 100k configurations")
    ap.add_argument("--live-traffic", default="auto", choices=["auto", "on", "off"],
                    help="measure roofline.traffic in this run (two rocprofv3 --pmc child runs, ~40 s); auto = only in the full default command "
                         "(one rank, --config vga with its `also` entries)")
    ap.add_argument("--detail-file", default=None, help="also write the full reports (headline + also + the compact line) to this JSON file")
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="bare --gpus N command: kill the ranks when the run exceeds this many seconds")
    ap.add_argument("--also-cpu-seconds", type=float, default=5.0, help="CPU baseline sample of each embedded configuration")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--cpu-allcores-seconds", type=float, default=8.0, help="0 disables the all-core CPU baseline")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo")
    ap.add_argument("--share-device", action="store_true",
                    help="functional smoke of the N>1 path on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    a = ap.parse_args()
    if a.no_parity:
        a.parity = "none"
    if a.parity is None:
        a.parity = "all" if max(a.gpus, int(os.environ.get("WORLD_SIZE", "1"))) == 1 else "sample"

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(a.gpus, a.share_device, a.rank_timeout))

    import torch
    from orb_slam_amd import dist_util, synth
    world, rank, local_rank = dist_util.env_ranks()
    if world == 1 or a.share_device:
        local_rank = 0
    if torch.cuda.device_count() <= local_rank:
        raise SystemExit("rank %d wants cuda:%d but this process sees %d device(s) (one process per GPU; --share-device for a 1-GPU functional run)"
                         % (rank, local_rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    numa = dist_util.bind_to_gpu_numa(local_rank, bind=world > 1 and not a.share_device)      # (at one rank: reported, not changed)
    a.numa_info = numa
    dist = dist_util.init(a.backend, world, rank, local_rank, force=a.force_dist)    # "nccl" is RCCL on ROCm
    if a.gpus != world:
        raise SystemExit("--gpus %d but WORLD_SIZE %d" % (a.gpus, world))

    def run(a_, name):
        cfg = dict(CONFIGS[name])
        for k in ("batch", "ring", "width", "height", "nfeatures"):
            if getattr(a_, k) is not None and k in cfg:
                cfg[k] = getattr(a_, k)
        if a_.no_match and "match" in cfg:
            cfg["match"] = False
        return (run_match if name == "match100k" else run_frontend)(a_, cfg, world, rank, local_rank, dist, torch)

    out = run(a, a.config)
    bad = int(out.get("_mismatches", 0)) if rank != 0 else int(out["config"].get("parity_mismatches", 0))
    if rank == 0 and numa.get("desc"):
        out["config"]["host_numa_binding"] = numa["desc"]
    also = {}
    if a.config == "vga" and a.family == synth.BLOCKS and not a.no_also:
        # BASELINE.json's other GPU configurations and the headline configuration on the other frame families, in the same run (the
        # driver runs this command once): same definitions, same parity leg, a timed region as long as the headline's for the frame
        # configurations (--also-min-seconds, 6 s: the chip clocks down over the first seconds of a region).  At
        # N > 1 `hd1080` is configs[3] (one 1080p stream per GPU); the frame families add nothing to a scaling run and are skipped there.
        entries = [("vga_extract", "vga_extract", synth.BLOCKS, False), ("hd1080", "hd1080", synth.BLOCKS, True),
                   ("match100k", "match100k", synth.BLOCKS, True), ("match100k_int8", "match100k", synth.BLOCKS, False),
                   ("match100k_popcount", "match100k", synth.BLOCKS, False),
                   ("vga_noise", "vga", synth.NOISE, False), ("vga_midtex", "vga", synth.MIDTEX, False), ("vga_lowtex", "vga", synth.LOWTEX, False),
                   ("vga_warp", "vga", synth.WARP, False)]
        if world > 1:
            entries = [e for e in entries if e[0] in ("hd1080", "match100k")]
        for key, name, family, cpu in entries:
            a2 = argparse.Namespace(**vars(a))
            # beside the default (FP4 MFMA) form: the int8 MFMA kernels of rounds 2-4 and the uint64 x 4 xor + popcount kernels north_star names
            a2.match_path = {"match100k_popcount": 0, "match100k_int8": 1}.get(key, -1)
            a2.config, a2.cpu_seconds, a2.cpu_allcores_seconds, a2.cpu_reference_seconds = name, a.also_cpu_seconds, 0.0, 0.0
            a2.min_seconds = a.also_match_min_seconds if name == "match100k" else a.also_min_seconds
            a2.batch = a2.ring = a2.width = a2.height = a2.nfeatures = None
            a2.region_timing = False
            a2.family = family
            a2.no_cpu_baseline = a.no_cpu_baseline or not cpu
            r = run(a2, name)
            if rank != 0:
                bad += int(r.get("_mismatches", 0))
            else:
                bad += int(r["config"].get("parity_mismatches", 0))
                also[key] = r
    # roofline.traffic measured in THIS run (rank 0 of a one-rank default command; everything timed is behind us)
    live = a.live_traffic == "on" or (a.live_traffic == "auto" and world == 1 and a.config == "vga" and a.family == synth.BLOCKS and not a.no_also
                                      and a.batch is None and a.width is None and a.nfeatures is None)
    if live and rank == 0 and world == 1 and isinstance(out.get("roofline"), dict) and out["roofline"].get("kernel"):
        lt, lt_note = measure_live_traffic(int(out["config"]["frames_per_step_per_gpu"]))
        rf = out["roofline"]
        if lt and rf["kernel"] in lt:
            rf["traffic_replayed"] = rf.get("traffic")
            rf["traffic"] = lt[rf["kernel"]]
            rf["traffic_source"] = "measured in this run"
            rf["traffic_note"] = lt_note
            out["traffic_per_stage_this_run"] = lt
            if isinstance(out.get("roofline_pipeline"), dict):      # all kernels of a step: HBM-side bytes per step against the algorithmic ones
                out["roofline_pipeline"]["traffic"] = int(sum(lt.values()))
        else:
            rf["traffic_source"] = "replayed from profiles/ (live pass unavailable: %s)" % lt_note
    # The JSON line must be the LAST line of the job's stdout.  RCCL writes to the C-level stdout ("Librccl path ...": buffered by libc, it
    # surfaced BEHIND the line when the process exited — tests/test_gpu_bench.py::test_rccl_path_at_world_size_one), and under
    # torch.distributed.run every rank shares rank 0's stdout.  So: every rank flushes libc's buffers, the other ranks then close their
    # stdout for good, a barrier, the communicator is destroyed, and only then rank 0 prints — and closes its stdout behind the line too.
    def c_flush():
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()

    def close_stdout():
        c_flush()
        os.dup2(os.open(os.devnull, os.O_WRONLY), 1)

    c_flush()
    if rank != 0:
        close_stdout()
    if dist is not None:
        dist_util.barrier(dist, a.backend, local_rank)
        dist.destroy_process_group()
    if rank == 0:
        # full reports first (lines that do not start with "{"), the compact line LAST: it is the one JSON line of this run
        print_detail("headline", out)
        for key, r in also.items():
            print_detail("also." + key, r)
        line = json.dumps(compact_line(out, also))
        if a.detail_file:
            with open(a.detail_file, "w") as f:
                json.dump(dict(out, also=also, line=json.loads(line)), f)
        c_flush()
        print(line, flush=True)
        close_stdout()
    if bad:                 # every rank: the mismatch counters were summed over the ranks
        sys.stderr.write("bench.py: %d outputs differ from the oracle (config.parity_detail)\n" % bad)
        sys.exit(1)


if __name__ == "__main__":
    main()
