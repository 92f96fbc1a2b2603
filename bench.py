#!/usr/bin/env python3
"""bench.py — throughput of the ORB front-end hot path on MI355X.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
  extract (pyramid -> FAST/NMS -> selection -> blur -> orientation + rBRIEF) for `batch` 640x480 frames,
  then brute-force Hamming top-2 matching of every frame's descriptors against the previous frame's.
Workload = BASELINE.json `metric` ("frames/s ORB extract+match @640x480, 1000 kp"): configs[1] (single MI355X,
640x480 stream, 8 levels, nFeatures 1000) plus the frame-to-frame match of the metric.

  python bench.py --gpus N --steps K --warmup W
For N>1 launch under torch.distributed.run (one rank per GPU); every rank extracts its own stream
(weak scaling, no data-path collective), RCCL only reduces the timing / counters.
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def level_sizes(w, h, nlevels=8, sf=1.2):
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


def cpu_baseline(w, h, nfeat, seconds=15.0):
    """Oracle (scalar CPU restatement of the reference algorithm) timed on this host, 1 core, bounded sample.
    The reference runs its extractor on the single Tracking thread (src/Tracking.cc:199-202), hence cores=1."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as orc
    from orb_slam_amd import synth
    o = orc.OracleExtractor(nfeatures=nfeat)
    imgs = synth.frames(w, h, synth.BLOCKS, 6000, 32)      # synthesis is outside the timed loop
    prev = o(imgs[0])[1]                                    # warm-up frame (page faults), not timed
    done = 0
    t = time.perf_counter()
    while time.perf_counter() - t < seconds:
        _, d = o(imgs[(done + 1) % len(imgs)])
        if len(d) and len(prev):
            orc.match_top2(d, prev)
        prev = d
        done += 1
    el = time.perf_counter() - t
    return {"value": round(done / el, 2), "unit": "frames/s", "cores": 1, "kind": "port",
            "sample": "%d S-blocks %dx%d frames, oracle extract (nFeatures %d) + scalar top-2 match vs previous frame, %.1f s"
                      % (done, w, h, nfeat, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=512, help="frames per step per GPU")
    ap.add_argument("--ring", type=int, default=1024, help="distinct frames resident per GPU (>= 1024 VGA frames exceeds the 256 MiB Infinity Cache)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--family", type=int, default=1, help="0 noise, 1 blocks (default), 3 lowtex")
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE configs[1] verbatim)")
    ap.add_argument("--match-stream", choices=["side", "same"], default="side",
                    help="side: the match of step i runs on a second stream while step i+1 is extracted (default); same: one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend: nccl (= RCCL, default) or gloo")
    ap.add_argument("--share-device", action="store_true", help="functional smoke of the N>1 path on a 1-GPU box: every rank uses cuda:0 (use with --backend gloo)")
    a = ap.parse_args()

    from orb_slam_amd import dist_util
    world, rank, local_rank = dist_util.env_ranks()
    if world == 1 or a.share_device:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = dist_util.init(a.backend, world, rank, local_rank)    # "nccl" is RCCL on ROCm
    assert a.gpus == world, "--gpus %d but WORLD_SIZE %d (launch with torch.distributed.run for N>1)" % (a.gpus, world)
    dev = torch.device("cuda", local_rank)

    from orb_slam_amd import capi, synth
    w, h, B = a.width, a.height, a.batch
    ring = max(a.ring // B, 1) * B
    frames = synth.frames(w, h, a.family, dist_util.stream_first_index(rank, ring), ring)          # one independent image stream per rank
    d_img = torch.from_numpy(frames).to(dev)
    del frames
    ex = capi.ORBextractor(nfeatures=a.nfeatures, device=local_rank, max_batch=B)
    cap = ex.max_keypoints
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device=dev)
    # two generations of outputs: the match of step i (side stream) runs while step i+1 is being extracted (main stream)
    d_desc = torch.zeros((2, B + 1, cap, 32), dtype=torch.uint8, device=dev)   # slot 0 = last frame of the previous step
    d_n = torch.zeros((2, B + 1), dtype=torch.int32, device=dev)
    d_status = torch.zeros(B, dtype=torch.int32, device=dev)
    d_match = torch.zeros((3, B, cap), dtype=torch.int32, device=dev)
    main = torch.cuda.current_stream(dev)
    stream = main.cuda_stream
    side = torch.cuda.Stream(dev) if a.match_stream == "side" else main
    do_match = not a.no_match
    match_events = []
    ev_extract = [None, None]      # extraction of the generation finished
    ev_match = [None, None]        # its match finished (the generation may be overwritten)

    def step(i, timed):
        # generation g holds this step's outputs in slots 1..B and the previous step's last frame in slot 0.
        # main stream: extract(i) -> [wait match(i-1)] -> hand slot B over to the other generation's slot 0.
        # side stream: match(i) as soon as extract(i) is done, i.e. concurrently with extract(i+1).
        f0 = (i * B) % ring
        g = i & 1
        ex.extract_batch_device(d_img.data_ptr() + f0 * w * h, B, w, h, w, w * h, d_kps.data_ptr(),
                                d_desc[g, 1].data_ptr(), d_n[g, 1:].data_ptr(), cap, d_status.data_ptr(), stream)
        if do_match:
            ev_extract[g] = torch.cuda.Event()
            ev_extract[g].record(main)
            with torch.cuda.stream(side):
                side.wait_event(ev_extract[g])
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(side)
                capi.match_top2_batch_device(d_desc[g, 1].data_ptr(), d_n[g, 1:].data_ptr(), d_desc[g, 0].data_ptr(), d_n[g].data_ptr(),
                                             B, cap, d_match[0].data_ptr(), d_match[1].data_ptr(), d_match[2].data_ptr(), side.cuda_stream)
                if timed:
                    e1.record(side)
                    match_events.append((e0, e1))
                ev_match[g] = torch.cuda.Event()
                ev_match[g].record(side)
            if ev_match[g ^ 1] is not None:
                main.wait_event(ev_match[g ^ 1])          # the other generation is free again (its match has finished)
            d_desc[g ^ 1, 0].copy_(d_desc[g, B], non_blocking=True)
            d_n[g ^ 1, 0:1].copy_(d_n[g, B:B + 1], non_blocking=True)

    for i in range(a.warmup):
        step(i, False)
    torch.cuda.synchronize(dev)
    ex.stage_timing(2)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i, True)
    torch.cuda.synchronize(dev)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    stage = ex.stage_times()
    ex.stage_timing(0)
    match_ms = sum(e0.elapsed_time(e1) for e0, e1 in match_events) / max(len(match_events), 1)
    last = (a.warmup + a.steps - 1) & 1
    kp_mean = float(d_n[last, 1:].float().mean().item())
    bad_status = int((d_status != 0).sum().item())
    accepted = -1
    if do_match:
        best = d_match[1, B - 1, :cap].cpu().numpy()
        sec = d_match[2, B - 1, :cap].cpu().numpy()
        nq = int(d_n[last, B].item())
        accepted = capi.count_accepted(best[:nq], sec[:nq], 50, 0.6)

    # RCCL: the only collectives of the run (MAX of the timing, all-gather of the counters)
    tmax, counters, _ = dist_util.reduce_run(dist, elapsed, [a.steps * B, kp_mean * a.steps * B, bad_status],
                                             dev if a.backend == "nccl" else torch.device("cpu"))
    total_frames = float(counters[0])

    if rank == 0:
        a_extract, a_match, per_stage = algorithmic_bytes(w, h, a.nfeatures)
        stage_ms = {k: (ms / n if n else 0.0) for k, (ms, n) in stage.items()}
        if do_match:
            stage_ms["match"] = match_ms
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        dom_bytes = (per_stage.get(dom, a_match if dom == "match" else 0)) * B
        dom_gbs = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        kernel_ms = sum(stage_ms.values())
        pipe_bytes = (a_extract + (a_match if do_match else 0)) * B
        pipe_gbs = pipe_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
        traffic, valu_busy = None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")      # written by tools/pmc_traffic.py from rocprofv3 --pmc passes
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("workload") == "vga_640x480_nf1000" and tj.get("batch") == B:
                    traffic = tj.get("per_launch_bytes", {}).get(dom)
                    valu_busy = tj.get("sq_activity", {}).get(dom, {}).get("valu_busy")
            except Exception:
                traffic = None
        out = {
            "metric": "frames/s ORB %s @%dx%d, %d kp" % ("extract+match" if do_match else "extract", w, h, a.nfeatures),
            "value": round(total_frames / tmax, 1),
            "unit": "frames/s",
            "n_gpus": world,
            "steps": a.steps,
            "warmup": a.warmup,
            "ms_per_step": round(tmax / a.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": "synthetic",
            "config": {"workload": "%dx%d grayscale stream, 8 levels, nFeatures %d, %s frames, extract%s" % (
                           w, h, a.nfeatures, {0: "S-noise", 1: "S-blocks", 3: "S-lowtex"}.get(a.family, str(a.family)),
                           " + Hamming top-2 match vs previous frame" if do_match else " only"),
                       "frames_per_step_per_gpu": B, "resident_frames_per_gpu": ring, "parallelism": "one image stream per GPU",
                       "mean_keypoints_per_frame": round(float(counters[1]) / total_frames, 2),
                       "frames_with_error_status": int(counters[2]), "accepted_matches_last_frame": accepted},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(dom_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom_gbs / HBM_PEAK_GBS, 5), "traffic": traffic, "valu_busy": valu_busy,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(stage_ms[dom], 4)},
            "roofline_pipeline": {"bound": "hbm", "achieved": round(pipe_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(pipe_gbs / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": pipe_bytes,
                                  "kernel_ms_per_step": round(kernel_ms, 4)},
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, h, a.nfeatures, a.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
