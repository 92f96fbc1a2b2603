#!/usr/bin/env python3
"""Per-kernel stage times of one launch group of 640x480 frames stored with different row pitches (is k_blur_mfma's 1080p
deficit a matter of the row pitch?).  usage: pitch_probe.py  -> one line per pitch"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam_amd import capi, synth
w, h, B = 640, 480, 256
frames = synth.frames(w, h, synth.BLOCKS, 0, B, threads=8)
for rs in (640, 1024, 1920, 2048, 4096):
    buf = np.zeros((B, h, rs), np.uint8); buf[:, :, :w] = frames
    d = torch.from_numpy(buf).cuda()
    os.environ["ORBX_OVERLAP"] = "0"
    ex = capi.ORBextractor(nfeatures=1000, max_batch=B)
    cap = ex.max_keypoints
    k = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); de = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda"); n = torch.zeros(B, dtype=torch.int32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for i in range(8):
        if i == 3:
            torch.cuda.synchronize(); ex.stage_timing(2)
        ex.extract_batch_device(d.data_ptr(), B, w, h, rs, rs * h, k.data_ptr(), de.data_ptr(), n.data_ptr(), cap, 0, st)
    torch.cuda.synchronize()
    t = ex.stage_times()
    print("pitch %4d:" % rs, " ".join("%s %.4f" % (s[:6], ms / max(c, 1)) for s, (ms, c) in t.items()), flush=True)
    ex.close()
