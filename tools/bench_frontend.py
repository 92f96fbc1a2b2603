#!/usr/bin/env python3
"""The whole per-frame front-end on device-resident data, one batch of VGA frames per step (SURVEY.md §8 + the §8f rows):
extract (orbx) -> undistort + grid (orbf) -> bag-of-words transform (orbv) -> greedy window search of frame t against
frame t-1 (orbs, WindowSearch rule with rotation check) -> dense top-2 match (orbm).  Prints one JSON line with the
per-stage times measured with events on the launch stream."""
import argparse
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from orb_slam_amd import capi, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--w", type=int, default=640); ap.add_argument("--h", type=int, default=480)
ap.add_argument("--nfeatures", type=int, default=1000)
ap.add_argument("--batch", type=int, default=512); ap.add_argument("--ring", type=int, default=1024)
ap.add_argument("--steps", type=int, default=10); ap.add_argument("--window", type=float, default=100.0)
ap.add_argument("--voc-k", type=int, default=10); ap.add_argument("--voc-l", type=int, default=6)
a = ap.parse_args()
B, w, h = a.batch, a.w, a.h
frames = synth.frames(w, h, synth.BLOCKS, 0, a.ring)
d_img = torch.from_numpy(frames).cuda()
ex = capi.ORBextractor(nfeatures=a.nfeatures, max_batch=B)
cap = ex.max_keypoints
voc = synth.vocabulary(a.voc_k, a.voc_l, seed=1)
V = capi.ORBVocabulary.from_nodes(a.voc_k, a.voc_l, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
cam = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), w, h)
bounds = capi.image_bounds(cam)
dev = "cuda"
i32, u8, f32, f64 = torch.int32, torch.uint8, torch.float32, torch.float64
# two generations of frame state (t and t-1)
S = [dict(kps=torch.zeros((B, cap, 7), dtype=f32, device=dev), desc=torch.zeros((B, cap, 32), dtype=u8, device=dev),
          n=torch.zeros(B, dtype=i32, device=dev), un=torch.zeros((B, cap, 7), dtype=f32, device=dev),
          off=torch.zeros((B, capi.GRID_CELLS + 1), dtype=i32, device=dev), feat=torch.zeros((B, cap), dtype=i32, device=dev)) for _ in range(2)]
bow_id = torch.zeros((B, cap), dtype=i32, device=dev); bow_val = torch.zeros((B, cap), dtype=f64, device=dev)
fv_node = torch.zeros((B, cap), dtype=i32, device=dev); fv_off = torch.zeros((B, cap + 1), dtype=i32, device=dev)
fv_feat = torch.zeros((B, cap), dtype=i32, device=dev); cnt = torch.zeros((2, B), dtype=i32, device=dev)
qxyr = torch.zeros((B, cap, 3), dtype=f32, device=dev); qlev = torch.zeros((B, cap, 2), dtype=i32, device=dev)
qang = torch.zeros((B, cap), dtype=f32, device=dev)
q2t = torch.zeros((B, cap), dtype=i32, device=dev); t2q = torch.zeros((B, cap), dtype=i32, device=dev)
best = torch.zeros((B, cap), dtype=i32, device=dev); second = torch.zeros((B, cap), dtype=i32, device=dev)
nm = torch.zeros(B, dtype=i32, device=dev)
m_idx = torch.zeros((B, cap), dtype=i32, device=dev); m_best = torch.zeros((B, cap), dtype=i32, device=dev); m_sec = torch.zeros((B, cap), dtype=i32, device=dev)
st = torch.cuda.current_stream().cuda_stream
names = ["extract", "undistort_grid", "bow", "query_setup", "window_search", "dense_match"]
acc = {k: 0.0 for k in names}


def step(i, timed):
    cur, prev = S[i & 1], S[(i + 1) & 1]
    f0 = (i * B) % (a.ring - B + 1)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)]
    ev[0].record()
    ex.extract_batch_device(d_img.data_ptr() + f0 * w * h, B, w, h, w, w * h, cur["kps"].data_ptr(), cur["desc"].data_ptr(), cur["n"].data_ptr(), cap, 0, st)
    ev[1].record()
    capi.undistort_grid_batch_device(cam, bounds, cur["kps"].data_ptr(), cur["n"].data_ptr(), B, cap, cur["un"].data_ptr(), cur["off"].data_ptr(),
                                     cur["feat"].data_ptr(), st)
    ev[2].record()
    V.transform_batch_device(cur["desc"].data_ptr(), cur["n"].data_ptr(), B, cap, 4, bow_id.data_ptr(), bow_val.data_ptr(), cnt[0].data_ptr(),
                             fv_node.data_ptr(), fv_off.data_ptr(), fv_feat.data_ptr(), cnt[1].data_ptr(), st)
    ev[3].record()
    # WindowSearch(last, current, window, ..): queries = the previous frame's undistorted keypoints at their own level
    qxyr[:, :, 0:2] = prev["un"][:, :, 0:2]
    qxyr[:, :, 2] = a.window
    oct_prev = prev["un"][:, :, 5].view(i32)
    qlev[:, :, 0] = oct_prev
    qlev[:, :, 1] = oct_prev
    qang.copy_(prev["un"][:, :, 3])
    ev[4].record()
    capi.window_search_batch_device(bounds, capi.RULE_WINDOW, capi.TH_HIGH, 0.8, True, cur["un"].data_ptr(), cur["desc"].data_ptr(), cur["off"].data_ptr(),
                                    cur["feat"].data_ptr(), cur["n"].data_ptr(), cap, 0, qxyr.data_ptr(), qlev.data_ptr(), prev["desc"].data_ptr(),
                                    qang.data_ptr(), 0, prev["n"].data_ptr(), cap, B, q2t.data_ptr(), t2q.data_ptr(), best.data_ptr(), second.data_ptr(),
                                    nm.data_ptr(), st)
    ev[5].record()
    capi.match_top2_batch_device(cur["desc"].data_ptr(), cur["n"].data_ptr(), prev["desc"].data_ptr(), prev["n"].data_ptr(), B, cap, m_idx.data_ptr(),
                                 m_best.data_ptr(), m_sec.data_ptr(), st)
    ev[6].record()
    if timed:
        torch.cuda.synchronize()
        for j, k in enumerate(names):
            acc[k] += ev[j].elapsed_time(ev[j + 1])


for i in range(3):
    step(i, False)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(3, 3 + a.steps):
    step(i, False)
e1.record()
torch.cuda.synchronize()
total_ms = e0.elapsed_time(e1) / a.steps
for i in range(3 + a.steps, 3 + 2 * a.steps):
    step(i, True)
out = {"metric": "frontend_frames_per_s", "value": round(B / (total_ms * 1e-3), 1), "unit": "frames/s", "ms_per_step": round(total_ms, 4),
       "config": {"workload": "%dx%d, %d kp, batch %d: extract + undistort/grid + BoW(k=%d,L=%d) + WindowSearch(r=%g, rot) + dense top-2" %
                  (w, h, a.nfeatures, B, a.voc_k, a.voc_l, a.window)},
       "stage_ms_per_step": {k: round(v / a.steps, 4) for k, v in acc.items()},
       "mean_keypoints": round(float(S[0]["n"].float().mean().item()), 1), "mean_window_matches": round(float(nm.float().mean().item()), 1),
       "mean_bow_words": round(float(cnt[0].float().mean().item()), 1)}
# the same chain on one host core through the CPU oracle (test infrastructure), a bounded sample of frames
if os.environ.get("FRONTEND_CPU", "1") != "0":
    import time
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import oracle_lib as ol
    oe = ol.OracleExtractor(a.nfeatures)
    ov = ol.OracleVocabulary(voc=voc)
    ns = 4
    t = {k: 0.0 for k in ("extract", "undistort_grid", "bow", "window_search", "dense_match")}
    prev = None
    for f in range(ns + 1):
        t0 = time.perf_counter(); k, d = oe(frames[f]); t1 = time.perf_counter()
        un = ol.frame_undistort(cam, k); off, feat = ol.frame_grid(bounds, un); t2 = time.perf_counter()
        ov.transform(d, 4); t3 = time.perf_counter()
        if prev is not None:
            pk, pd, pun = prev
            qx = np.stack([pun["x"], pun["y"], np.full(len(pun), a.window, np.float32)], -1)
            ql = np.stack([pun["octave"], pun["octave"]], -1).astype(np.int32)
            t4 = time.perf_counter()
            ol.window_search(bounds, capi.RULE_WINDOW, capi.TH_HIGH, 0.8, True, un, d, off, feat, None, qx, ql, pd, pun["angle"], None)
            t5 = time.perf_counter()
            ol.match_top2(d, pd)
            t6 = time.perf_counter()
            t["extract"] += t1 - t0; t["undistort_grid"] += t2 - t1; t["bow"] += t3 - t2; t["window_search"] += t5 - t4; t["dense_match"] += t6 - t5
        prev = (k, d, un)
    out["cpu_oracle"] = {"frames_per_s": round(ns / sum(t.values()), 2), "cores": 1, "sample": "%d frames" % ns,
                         "stage_ms_per_frame": {k: round(v / ns * 1e3, 3) for k, v in t.items()}}
print(json.dumps(out))
