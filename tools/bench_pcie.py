#!/usr/bin/env python3
"""PCIe-inclusive rate of the throughput configuration (never `value`, noted beside it).

The step of bench.py (1024 VGA frames in 4 lanes, extract + match vs the previous frame), but every frame starts in PINNED HOST memory and
every output (counts, keypoints, descriptors, match triples — the full capacity arrays, as a caller without a compaction pass would take
them) ends in pinned host memory:

  H2D of step i+1 (one copy stream per lane, double-buffered device frames)  ||  lane kernels of step i  ->  D2H of step i on the lane stream

Prints ONE JSON line: frames/s with both copies inside the timed region, the bytes per step each way and the PCIe rates they imply.
The outputs of the last step are compared with a device-resident run of the same frames (same library, no copies): byte-identical or exit 1."""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam_amd import capi, synth
from orb_slam_amd.pipeline import LanePipeline

ap = argparse.ArgumentParser()
ap.add_argument("--width", type=int, default=640)
ap.add_argument("--height", type=int, default=480)
ap.add_argument("--nfeatures", type=int, default=1000)
ap.add_argument("--batch", type=int, default=1024)
ap.add_argument("--ring", type=int, default=2048)
ap.add_argument("--lanes", type=int, default=4)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=4)
ap.add_argument("--copy-streams", type=int, default=0, help="0: one upload stream per lane (a lane starts as soon as ITS slice is there); 1: one stream, one copy per step")
ap.add_argument("--outputs", default="full", choices=["full", "counts"], help="full: every output array returns to the host; counts: only n per frame")
ap.add_argument("--numa-bind", action="store_true", help="bind the process to the GPU's NUMA node BEFORE the pinned buffers are allocated (first touch decides where they live)")
ap.add_argument("--chunks", type=int, default=1, help="split every lane's upload into this many copies, alternating over --streams-per-lane upload streams (more SDMA engines in flight)")
ap.add_argument("--streams-per-lane", type=int, default=1)
a = ap.parse_args()
w, h, B, ring = a.width, a.height, a.batch, max(a.ring // a.batch, 1) * a.batch
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
numa = None
if a.numa_bind:
    from orb_slam_amd import dist_util
    numa = dist_util.bind_to_gpu_numa(0)

host = torch.from_numpy(synth.frames(w, h, synth.BLOCKS, 0, ring)).pin_memory()            # the camera side: frames in pinned host memory
pipe = LanePipeline(w, h, B, lanes=a.lanes, nfeatures=a.nfeatures, device=0)
G, b, cap = pipe.G, pipe.b, pipe.cap
d_frames = [torch.empty((B, h, w), dtype=torch.uint8, device=dev) for _ in range(2)]
d_frames[0].copy_(host[:B])
torch.cuda.synchronize()
pipe.tune(d_frames[0].data_ptr())

copy_streams = [torch.cuda.Stream(dev) for _ in range(G)]
extra_streams = [[copy_streams[g]] + [torch.cuda.Stream(dev) for _ in range(a.streams_per_lane - 1)] for g in range(G)]
out = [dict(n=torch.empty(b, dtype=torch.int32).pin_memory(), kps=torch.empty((b, cap, 7), dtype=torch.float32).pin_memory(),
            desc=torch.empty((b, cap, 32), dtype=torch.uint8).pin_memory(), match=torch.empty((3, b, cap), dtype=torch.int32).pin_memory())
       for _ in range(G)]
uploaded = [[None, None] for _ in range(G)]       # event: slice g of device buffer p holds its frames
released = [[None, None] for _ in range(G)]       # event: the kernels that read slice g of buffer p are done


def upload(i):
    p, f0 = i & 1, (i * B) % ring
    if a.copy_streams == 1:
        cs = copy_streams[0]
        for g in range(G):
            if released[g][p] is not None:
                cs.wait_event(released[g][p])
        with torch.cuda.stream(cs):
            d_frames[p].copy_(host[f0:f0 + B], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(cs)
        for g in range(G):
            uploaded[g][p] = ev
        return
    for g in range(G):
        cs = copy_streams[g]
        if released[g][p] is not None:
            cs.wait_event(released[g][p])
        if a.chunks <= 1 and a.streams_per_lane <= 1:
            with torch.cuda.stream(cs):
                d_frames[p][g * b:(g + 1) * b].copy_(host[f0 + g * b:f0 + (g + 1) * b], non_blocking=True)
                uploaded[g][p] = torch.cuda.Event()
                uploaded[g][p].record(cs)
            continue
        # the lane's slice in `chunks` copies over its upload streams; the lane's first stream joins them and records the event
        sts = extra_streams[g]
        for st in sts[1:]:
            if released[g][p] is not None:
                st.wait_event(released[g][p])
        per = (b + a.chunks - 1) // a.chunks
        evs = []
        for c in range(a.chunks):
            lo, hi = g * b + c * per, min(g * b + (c + 1) * per, (g + 1) * b)
            if lo >= hi:
                break
            st = sts[c % len(sts)]
            with torch.cuda.stream(st):
                d_frames[p][lo:hi].copy_(host[f0 + lo:f0 + hi], non_blocking=True)
                if st is not cs:
                    e = torch.cuda.Event(); e.record(st); evs.append(e)
        for e in evs:
            cs.wait_event(e)
        with torch.cuda.stream(cs):
            uploaded[g][p] = torch.cuda.Event()
            uploaded[g][p].record(cs)


def run(i):
    p = i & 1
    for g, ln in enumerate(pipe.lanes):
        ln.stream.wait_event(uploaded[g][p])
    pipe.step(d_frames[p].data_ptr())
    for g, ln in enumerate(pipe.lanes):
        with torch.cuda.stream(ln.stream):
            released[g][p] = torch.cuda.Event()
            released[g][p].record(ln.stream)
            out[g]["n"].copy_(ln.n[1:], non_blocking=True)
            if a.outputs == "full":
                out[g]["kps"].copy_(ln.kps, non_blocking=True)
                out[g]["desc"].copy_(ln.desc[1:], non_blocking=True)
                out[g]["match"].copy_(ln.match, non_blocking=True)


upload(0)
for i in range(a.warmup):
    upload(i + 1)
    run(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(a.warmup, a.warmup + a.steps):
    upload(i + 1)
    run(i)
torch.cuda.synchronize()
el = time.perf_counter() - t0
last = a.warmup + a.steps - 1
got = {k: torch.cat([o[k] for o in out], dim=1 if k == "match" else 0).clone() for k in out[0]}

# the same frames through a device-resident pipeline state: replay the last two steps without copies and compare
d_all = host.to(dev)
pipe2 = LanePipeline(w, h, B, lanes=a.lanes, nfeatures=a.nfeatures, device=0, placement=pipe.placement["chosen"])
for i in range(max(last - 1, 0), last + 1):
    pipe2.step(d_all.data_ptr() + ((i * B) % ring) * w * h)
torch.cuda.synchronize()
bad = 0
n_ref = pipe2.counts().cpu()
bad += int((n_ref != got["n"]).sum())
if a.outputs == "full":
    kp, de, ma = pipe2.keypoints().cpu(), pipe2.descriptors().cpu(), pipe2.matches().cpu()
    for f in range(B):
        n = int(n_ref[f])
        if not (torch.equal(kp[f, :n].view(torch.int32), got["kps"][f, :n].view(torch.int32)) and     # class_id = -1 is a NaN pattern
                torch.equal(de[f, :n], got["desc"][f, :n])):
            bad += 1
        elif (last > 0 or f > 0) and not torch.equal(ma[:, f, :n], got["match"][:, f, :n]):
            bad += 1
h2d = B * w * h
d2h = B * 4 + (B * cap * (28 + 32 + 12) if a.outputs == "full" else 0)
ms = el * 1e3 / a.steps
print(json.dumps({"metric": "frames_per_s_pcie_inclusive", "value": round(B * a.steps / el, 1), "unit": "frames/s", "ms_per_step": round(ms, 4),
                  "steps": a.steps, "warmup": a.warmup, "workload": "%dx%d nf=%d, %d frames per step in %d lanes, extract + match" % (w, h, a.nfeatures, B, G),
                  "outputs_to_host": a.outputs, "upload_streams": 1 if a.copy_streams == 1 else G, "h2d_bytes_per_step": h2d, "d2h_bytes_per_step": d2h,
                  "h2d_GBps": round(h2d / ms / 1e6, 2), "d2h_GBps": round(d2h / ms / 1e6, 2),
                  "frames_compared_with_device_resident_run": B, "mismatching_frames": bad, "placement": pipe.placement["chosen"],
                  "upload_chunks_per_lane": a.chunks, "upload_streams_per_lane": a.streams_per_lane, "numa_binding": numa}))
sys.stdout.flush()
uploaded = released = None                 # events recorded on the lane streams go before the streams do
del out, copy_streams, extra_streams
torch.cuda.synchronize()
pipe.close()
pipe2.close()
sys.exit(1 if bad else 0)
