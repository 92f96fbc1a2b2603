#!/usr/bin/env python3
"""Quick single-GPU timing of the extractor stages (development aid; bench.py is the contract)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=640); ap.add_argument("--h", type=int, default=480)
ap.add_argument("--nfeatures", type=int, default=1000)
ap.add_argument("--batch", type=int, default=128); ap.add_argument("--ring", type=int, default=256)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--family", type=int, default=1)
a = ap.parse_args()
t = time.time()
frames = synth.frames(a.w, a.h, a.family, 0, a.ring)
print("synth %.2fs" % (time.time() - t))
d_img = torch.from_numpy(frames).cuda()
ex = capi.ORBextractor(nfeatures=a.nfeatures, max_batch=a.batch)
cap = ex.max_keypoints
d_kps = torch.zeros((a.batch, cap, 7), dtype=torch.float32, device="cuda")
d_desc = torch.zeros((a.batch, cap, 32), dtype=torch.uint8, device="cuda")
d_n = torch.zeros(a.batch, dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def step(i):
    f0 = (i * a.batch) % (a.ring - a.batch + 1)
    ex.extract_batch_device(d_img.data_ptr() + f0 * a.w * a.h, a.batch, a.w, a.h, a.w, a.w * a.h, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, 0, s)
for i in range(2): step(i)
torch.cuda.synchronize()
ex.stage_timing(2)
t = time.time()
for i in range(a.steps): step(i)
torch.cuda.synchronize()
dt = time.time() - t
print("frames/s %.0f  (%.3f ms/batch of %d)  mean N=%.1f" % (a.steps * a.batch / dt, dt / a.steps * 1e3, a.batch, d_n.float().mean().item()))
tot = 0
for k, (ms, n) in ex.stage_times().items():
    print("  %-13s %8.3f ms/batch  (%d groups)" % (k, ms / max(n, 1), n)); tot += ms / max(n, 1)
print("  sum %.3f ms/batch -> %.0f frames/s kernel-only" % (tot, a.batch / tot * 1e3))
