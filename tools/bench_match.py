#!/usr/bin/env python3
"""BASELINE config 5: batched N-to-M descriptor match, 100k x 100k 256-bit descriptors (dense top-2)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam_amd import capi, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
Q = torch.from_numpy(synth.descriptors(n, 1)).cuda(); T = torch.from_numpy(synth.descriptors(n, 2)).cuda()
out = torch.zeros((3, n), dtype=torch.int32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2): capi.match_top2_device(Q.data_ptr(), n, T.data_ptr(), n, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), s)
torch.cuda.synchronize(); t = time.time(); K = 5
for _ in range(K): capi.match_top2_device(Q.data_ptr(), n, T.data_ptr(), n, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), s)
torch.cuda.synchronize(); dt = (time.time() - t) / K
pairs = float(n) * n
print("match %dx%d: %.3f ms  %.3e pairs/s  (%.1f%% of the 16 lane-op/pair VALU model at 3.93e13 lane-ops/s)  A_match GB/s %.2f" % (
    n, n, dt * 1e3, pairs / dt, 100 * pairs * 16 / dt / 3.93e13, (32 * 2 * n + 12 * n) / dt / 1e9))
