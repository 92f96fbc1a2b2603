#!/usr/bin/env python3
"""Timing of the greedy grid-window search kernel alone on synthetic frames (uniform keypoints, per-level counts of the
VGA/1000 extractor), queries = the train features displaced by a few pixels with a few descriptor bits flipped."""
import argparse, json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from orb_slam_amd import capi

ap = argparse.ArgumentParser()
ap.add_argument("--problems", type=int, default=512); ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--window", type=float, default=15.0); ap.add_argument("--rule", type=int, default=1)
ap.add_argument("--iters", type=int, default=10); ap.add_argument("--levels", default="same")
a = ap.parse_args()
P, n = a.problems, a.n
rng = np.random.default_rng(1)
cam = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.0, 0.0, 0.0, 0.0), 640, 480)
b = capi.image_bounds(cam)
quota = np.array([217, 181, 151, 126, 105, 87, 73, 60]) * n // 1000
octv = np.repeat(np.arange(8), quota)
octv = np.concatenate([octv, np.zeros(n - len(octv), np.int64)])[:n]
K = np.zeros((P, n), dtype=capi.KP_DTYPE)
K["x"] = rng.random((P, n)).astype(np.float32) * 640
K["y"] = rng.random((P, n)).astype(np.float32) * 480
K["angle"] = rng.random((P, n)).astype(np.float32) * 360
K["octave"] = octv[None, :]
D = rng.integers(0, 256, (P, n, 32), dtype=np.uint8)
perm = np.stack([rng.permutation(n) for _ in range(P)])
QX = np.zeros((P, n, 3), np.float32)
QX[:, :, 0] = np.take_along_axis(K["x"], perm, 1) + rng.normal(0, 3, (P, n))
QX[:, :, 1] = np.take_along_axis(K["y"], perm, 1) + rng.normal(0, 3, (P, n))
QX[:, :, 2] = a.window
lv = np.take_along_axis(K["octave"], perm, 1)
QL = np.stack([lv, lv], -1).astype(np.int32) if a.levels == "same" else np.stack([lv - 1, lv + 1], -1).astype(np.int32)
QD = np.take_along_axis(D, perm[:, :, None], 1).copy()
QD[:, :, 0] ^= rng.integers(0, 256, (P, n), dtype=np.uint8)
QA = (np.take_along_axis(K["angle"], perm, 1) + rng.normal(0, 5, (P, n))).astype(np.float32) % 360
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
dK = t(K.view(np.uint8).reshape(P, n, 28)); dn = torch.full((P,), n, dtype=torch.int32, device="cuda")
dUn = torch.zeros((P, n, 28), dtype=torch.uint8, device="cuda"); dOff = torch.zeros((P, capi.GRID_CELLS + 1), dtype=torch.int32, device="cuda")
dFeat = torch.zeros((P, n), dtype=torch.int32, device="cuda")
st = torch.cuda.current_stream().cuda_stream
capi.undistort_grid_batch_device(cam, b, dK.data_ptr(), dn.data_ptr(), P, n, dUn.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), st)
dD, dQX, dQL, dQD, dQA = t(D), t(QX), t(QL), t(QD), t(QA)
q2t = torch.zeros((P, n), dtype=torch.int32, device="cuda"); t2q = torch.zeros((P, n), dtype=torch.int32, device="cuda")
best = torch.zeros((P, n), dtype=torch.int32, device="cuda"); sec = torch.zeros((P, n), dtype=torch.int32, device="cuda")
nm = torch.zeros(P, dtype=torch.int32, device="cuda")
def run():
    capi.window_search_batch_device(b, a.rule, 100, 0.8, True, dUn.data_ptr(), dD.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), dn.data_ptr(), n, 0,
                                    dQX.data_ptr(), dQL.data_ptr(), dQD.data_ptr(), dQA.data_ptr(), 0, dn.data_ptr(), n, P, q2t.data_ptr(), t2q.data_ptr(),
                                    best.data_ptr(), sec.data_ptr(), nm.data_ptr(), st)
for _ in range(2): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(a.iters): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / a.iters
print(json.dumps({"ms": round(ms, 4), "problems": P, "n": n, "window": a.window, "rule": a.rule, "queries_per_s": round(P * n / ms * 1e3),
                  "mean_matches": round(float(nm.float().mean()), 1)}))
