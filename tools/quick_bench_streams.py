#!/usr/bin/env python3
"""Development aid: S extractor handles on S streams, each taking batch/S frames of every step — does co-running the
latency-bound and the VALU-bound kernels of different sub-batches beat one big batch on one stream?"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--ring", type=int, default=1024)
ap.add_argument("--steps", type=int, default=20); ap.add_argument("--streams", default="1,2,4")
ap.add_argument("--match", action="store_true", help="each lane also matches its frames against the previous one (same stream)")
a = ap.parse_args()
w, h = 640, 480
d_img = torch.from_numpy(synth.frames(w, h, 1, 0, a.ring)).cuda()
for S in [int(x) for x in a.streams.split(",")]:
    b = a.batch // S
    exs = [capi.ORBextractor(nfeatures=1000, max_batch=b) for _ in range(S)]
    cap = exs[0].max_keypoints
    streams = [torch.cuda.Stream() for _ in range(S)]
    d_kps = torch.zeros((a.batch, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((a.batch + 1, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(a.batch + 1, dtype=torch.int32, device="cuda")
    d_match = torch.zeros((3, a.batch, cap), dtype=torch.int32, device="cuda")
    def step(i):
        f0 = (i * a.batch) % a.ring
        for s in range(S):
            exs[s].extract_batch_device(d_img.data_ptr() + (f0 + s * b) * w * h, b, w, h, w, w * h, d_kps[s * b].data_ptr(), d_desc[s * b + 1].data_ptr(),
                                        d_n[s * b + 1:].data_ptr(), cap, 0, streams[s].cuda_stream)
            if a.match:         # (approximation for timing: the train side of a lane's first frame is whatever sits in the slot before it)
                capi.match_top2_batch_device(d_desc[s * b + 1].data_ptr(), d_n[s * b + 1:].data_ptr(), d_desc[s * b].data_ptr(), d_n[s * b:].data_ptr(), b, cap,
                                             d_match[0, s * b].data_ptr(), d_match[1, s * b].data_ptr(), d_match[2, s * b].data_ptr(), streams[s].cuda_stream)
    for i in range(3): step(i)
    torch.cuda.synchronize()
    t = time.time()
    for i in range(a.steps): step(i)
    torch.cuda.synchronize()
    dt = time.time() - t
    print("streams %d x batch %d: %.0f frames/s (%.3f ms per %d frames), mean N=%.1f" % (S, b, a.steps * a.batch / dt, dt / a.steps * 1e3, a.batch, d_n.float().mean().item()))
    del exs
