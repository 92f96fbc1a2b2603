#!/usr/bin/env python3
"""Latency / rate of the host-buffer drop-in call orbx_extract (one frame per call, PCIe copies included)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from orb_slam_amd import capi, synth
w, h, nf = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480, 1000)
frames = synth.frames(w, h, synth.BLOCKS, 0, 64)
ex = capi.ORBextractor(nfeatures=nf)
for i in range(10): ex(frames[i])
ts = []
for i in range(300):
    t = time.perf_counter(); k, d = ex(frames[i % 64]); ts.append(time.perf_counter() - t)
ts = np.array(ts) * 1e6
print("orbx_extract %dx%d nf=%d: median %.0f us  p10 %.0f  p90 %.0f  -> %.0f frames/s single stream (host buffers, H2D+D2H included), N=%d" % (
    w, h, nf, np.median(ts), np.percentile(ts, 10), np.percentile(ts, 90), 1e6 / np.median(ts), len(k)))
