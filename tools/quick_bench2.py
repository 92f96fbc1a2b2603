#!/usr/bin/env python3
"""Two extractor handles on two streams (inter-batch overlap experiment)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam_amd import capi, synth
B, ring, w, h, steps = 256, 1024, 640, 480, 40
nh = int(sys.argv[1]) if len(sys.argv) > 1 else 2
frames = synth.frames(w, h, 1, 0, ring); d_img = torch.from_numpy(frames).cuda()
exs = [capi.ORBextractor(max_batch=B) for _ in range(nh)]
cap = exs[0].max_keypoints
bufs = [(torch.zeros((B, cap, 7), device="cuda"), torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"), torch.zeros(B, dtype=torch.int32, device="cuda")) for _ in range(nh)]
streams = [torch.cuda.Stream() for _ in range(nh)]
def step(i):
    k = i % nh; f0 = (i * B) % ring
    kp, de, n = bufs[k]
    exs[k].extract_batch_device(d_img.data_ptr() + f0 * w * h, B, w, h, w, w * h, kp.data_ptr(), de.data_ptr(), n.data_ptr(), cap, 0, streams[k].cuda_stream)
for i in range(4): step(i)
torch.cuda.synchronize(); t = time.time()
for i in range(steps): step(i)
torch.cuda.synchronize(); dt = time.time() - t
print("handles %d: frames/s %.0f (%.3f ms per batch)" % (nh, steps * B / dt, dt / steps * 1e3))
