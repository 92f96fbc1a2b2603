#!/usr/bin/env python3
"""bench.py — throughput of the ORB front-end hot path on MI355X.

A step = one pass of the hot path over one batch of synthetic frames already resident in HBM:
  extract (pyramid -> FAST/NMS -> selection -> blur -> orientation + rBRIEF) for `batch` 640x480 frames,
  then brute-force Hamming top-2 matching of every frame's descriptors against the previous frame's.
Workload = BASELINE.json `metric` ("frames/s ORB extract+match @640x480, 1000 kp"): configs[1] (single MI355X,
640x480 stream, 8 levels, nFeatures 1000) plus the frame-to-frame match of the metric.

A step's `batch` consecutive frames go through `--lanes` (default 4) lanes of batch/lanes consecutive frames, each lane with its
own extractor handle and HIP stream; a lane matches its own frames, and lanes meet only where the frame-to-frame match crosses
a lane border (event-ordered hand-off of one frame's descriptors).  So the latency-bound kernels of one lane run next to the
VALU-bound kernels of another, and consecutive steps overlap.
The per-kernel roofline numbers come from a short serial pass after the timed region (every kernel alone on the chip).

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
    ap.add_argument("--batch", type=int, default=1024, help="frames per step per GPU")
    ap.add_argument("--ring", type=int, default=2048, help="distinct frames resident per GPU (>= 1024 VGA frames exceeds the 256 MiB Infinity Cache)")
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--family", type=int, default=1, help="0 noise, 1 blocks (default), 3 lowtex")
    ap.add_argument("--no-match", action="store_true", help="extract only (BASELINE configs[1] verbatim)")
    ap.add_argument("--lanes", type=int, default=4,
                    help="a step's frames go through this many concurrent lanes (own extractor handle + HIP stream each); 1 = one stream")
    ap.add_argument("--region-timing", action="store_true",
                    help="also time every kernel inside the timed region (HIP events between the kernels of every lane: costs a few percent)")
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
    do_match = not a.no_match
    from orb_slam_amd.pipeline import LanePipeline
    pipe = LanePipeline(w, h, B, lanes=a.lanes, nfeatures=a.nfeatures, device=local_rank, do_match=do_match)   # orb_slam_amd/pipeline.py
    G, b, cap = pipe.G, pipe.b, pipe.cap

    def step(i, timed):
        pipe.step(d_img.data_ptr() + ((i * B) % ring) * w * h, timed=timed)

    for i in range(a.warmup):
        step(i, False)
    torch.cuda.synchronize(dev)
    pipe.stage_timing(2 if a.region_timing else 0)
    dist_util.barrier(dist, a.backend, local_rank)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(a.steps):
        step(a.warmup + i, a.region_timing)
    torch.cuda.synchronize(dev)
    dist_util.barrier(dist, a.backend, local_rank)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0

    stage = pipe.stage_times()
    pipe.stage_timing(0)
    kp_mean = float(pipe.counts().float().mean().item())
    bad_status = int((pipe.status() != 0).sum().item())
    accepted = -1
    if do_match:
        last = pipe.lanes[G - 1]
        best = last.match[1, b - 1, :cap].cpu().numpy()
        sec = last.match[2, b - 1, :cap].cpu().numpy()
        nq = int(last.n[b].item())
        accepted = capi.count_accepted(best[:nq], sec[:nq], 50, 0.6)

    def serial_pass(nsteps):
        """The same step with every kernel alone on the chip: one extractor over all B frames, one launch per kernel, the match
        on the same stream.  Per-kernel roofline numbers come from here; in the timed region above the lanes co-run, so a
        kernel's duration there includes sharing the CUs with the kernels of the other lanes."""
        keep = os.environ.get("ORBX_OVERLAP")
        os.environ["ORBX_OVERLAP"] = "0"          # read by orbx_create: no blur side stream either, every kernel alone on the chip
        try:
            ex1 = capi.ORBextractor(nfeatures=a.nfeatures, device=local_rank, max_batch=B)
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
        for i in range(2 + nsteps):
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
    tmax, counters, _ = dist_util.reduce_run(dist, elapsed, [a.steps * B, kp_mean * a.steps * B, bad_status],
                                             dev if a.backend == "nccl" else torch.device("cpu"))
    total_frames = float(counters[0])

    if rank == 0:
        a_extract, a_match, per_stage = algorithmic_bytes(w, h, a.nfeatures)
        region_ms = {k: (ms / n if n else 0.0) for k, (ms, n) in stage.items()}      # per LAUNCH: one lane's slice of b frames
        concurrent = G > 1 or not a.region_timing        # without in-region timing the serial pass is the only per-kernel timing
        stage_ms = serial_pass(min(a.steps, 10)) if concurrent else dict(region_ms)
        dom = max(stage_ms, key=lambda k: stage_ms[k])
        dom_frame_bytes = per_stage.get(dom, a_match if dom == "match" else 0)
        dom_bytes = dom_frame_bytes * B
        dom_gbs = dom_bytes / (stage_ms[dom] * 1e-3) / 1e9 if stage_ms[dom] > 0 else 0.0
        region_frames = b
        region_gbs = dom_frame_bytes * region_frames / (region_ms[dom] * 1e-3) / 1e9 if region_ms.get(dom, 0) > 0 else 0.0
        kernel_ms = sum(stage_ms.values())
        pipe_bytes = (a_extract + (a_match if do_match else 0)) * B
        step_ms = tmax / a.steps * 1e3
        pipe_gbs = pipe_bytes / (step_ms * 1e-3) / 1e9
        traffic, valu_busy, valu_insts = None, None, None
        tfile = os.path.join(ROOT, "profiles", "traffic.json")      # written by tools/pmc_traffic.py from rocprofv3 --pmc passes
        if os.path.exists(tfile):
            try:
                tj = json.load(open(tfile))
                if tj.get("workload") == "vga_640x480_nf1000" and (w, h, a.nfeatures, a.family) == (640, 480, 1000, 1):
                    if tj.get("batch") == B:
                        traffic = tj.get("per_launch_bytes", {}).get(dom)
                    valu_busy = tj.get("sq_activity", {}).get(dom, {}).get("valu_busy")
                    pf = tj.get("valu_wave_insts_per_frame", {})
                    if pf and (not do_match or "match" in pf):
                        valu_insts = {k: v for k, v in pf.items() if do_match or k != "match"}
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
                       "frames_per_step_per_gpu": B, "resident_frames_per_gpu": ring,
                       "parallelism": "one image stream per GPU; a step's %d frames go through %d lanes of %d consecutive frames "
                                      "(own extractor handle + HIP stream each), frame-to-frame matches across lane borders via event-ordered hand-off" % (B, G, b),
                       "lanes": G,
                       "mean_keypoints_per_frame": round(float(counters[1]) / total_frames, 2),
                       "frames_with_error_status": int(counters[2]), "accepted_matches_last_frame": accepted},
            "roofline": {"bound": "hbm", "kernel": dom, "achieved": round(dom_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(dom_gbs / HBM_PEAK_GBS, 5), "traffic": traffic, "valu_busy": valu_busy,
                         "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_ms": round(stage_ms[dom], 4), "frames_per_launch": B,
                         "timing": ("serial pass after the timed region: %d steps, one launch per kernel over all %d frames, nothing else on the chip "
                                    "(HIP events on the launch stream)" % (min(a.steps, 10), B)) if concurrent else "timed region (one stream)",
                         "timed_region": {"lanes": G, "frames_per_launch": region_frames, "avg_launch_ms": round(region_ms[dom], 4) if a.region_timing else None,
                                          "achieved": round(region_gbs, 2) if a.region_timing else None,
                                          "note": "the lanes co-run, a kernel shares the CUs with the kernels of the other lanes; --region-timing measures the "
                                                  "per-launch durations there (event pairs between all kernels cost 2-6 % of the throughput)"}},
            "roofline_pipeline": {"bound": "hbm", "achieved": round(pipe_gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                  "frac": round(pipe_gbs / HBM_PEAK_GBS, 5), "algorithmic_bytes_per_step": pipe_bytes,
                                  "ms_per_step": round(step_ms, 4), "kernel_ms_per_step_serial": round(kernel_ms, 4)},
            "stage_ms_per_step": {k: round(v, 4) for k, v in stage_ms.items()},
        }
        if valu_insts:
            # the resource that actually binds this integer path: VALU issue.  A wave64 op occupies one of the 1024 SIMDs for 4 cycles.
            # Measured on gfx950 (tools/microbench/valu_rate): 2 cycles for mov/add/sub/and/or/xor/ashr, 4 for the rest; the kernels'
            # mix is priced from their static opcode histograms (tools/valu_mix.py -> profiles/valu_mix.json), unmeasured opcodes at 2 (lo) / 4 (hi).
            simd_cycles = 256 * 4 * 2.4e9
            n_inst = sum(valu_insts.values())
            ach = out["value"] / world * n_inst
            rv = {"bound": "valu_issue", "achieved": round(ach / 1e9, 2), "unit": "G wave-insts/s", "wave_insts_per_frame": round(n_inst),
                  "peak_if_every_inst_took_4_cycles": round(simd_cycles / 4 / 1e9, 2), "frac_if_every_inst_took_4_cycles": round(ach / (simd_cycles / 4), 4),
                  "source": "SQ_INSTS_VALU of every kernel of the step per frame (rocprofv3 --pmc pass, profiles/traffic.json) x measured frames/s per GPU"}
            try:
                mix = json.load(open(os.path.join(ROOT, "profiles", "valu_mix.json")))["kernels"]
                lo = sum(v * mix[k]["cycles_per_inst_lo"] for k, v in valu_insts.items())
                hi = sum(v * mix[k]["cycles_per_inst_hi"] for k, v in valu_insts.items())
                rv.update({"peak": round(simd_cycles / (lo / n_inst) / 1e9, 2), "frac": round(out["value"] / world * lo / simd_cycles, 4),
                           "frac_range": [round(out["value"] / world * lo / simd_cycles, 4), round(out["value"] / world * hi / simd_cycles, 4)],
                           "cycles_per_inst_range": [round(lo / n_inst, 3), round(hi / n_inst, 3)],
                           "pricing": "2 cycles per wave64 inst for mov/add/sub/and/or/xor/ashr/fma_f32, 4 for every other measured opcode "
                                      "(profiles/r01_valu_issue_rates.txt), static opcode mix per kernel (profiles/valu_mix.json); peak and frac use the low end"})
            except Exception:
                pass
            out["roofline_valu"] = rv
        if a.region_timing:
            out["stage_ms_per_launch_timed_region"] = {k: round(v, 4) for k, v in region_ms.items()}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(w, h, a.nfeatures, a.cpu_seconds)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
