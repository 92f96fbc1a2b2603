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
ap.add_argument("--family", type=int, default=1, help="synthetic frame family: 1 S-blocks (independent frames), 5 S-warp (a correlated stream: the window search finds its partners)")
ap.add_argument("--lanes", type=int, default=4, help="the step's frames go through this many free-running lanes (own handles + stream each, NOTES.md §4.5)")
a = ap.parse_args()
B, w, h = a.batch, a.w, a.h
G = max(1, min(a.lanes, B))
while B % G:
    G -= 1
b = B // G
frames = synth.frames(w, h, a.family, 0, a.ring)
d_img = torch.from_numpy(frames).cuda()
voc = synth.vocabulary(a.voc_k, a.voc_l, seed=1)
cam = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), w, h)
bounds = capi.image_bounds(cam)
dev = "cuda"
i32, u8, f32, f64 = torch.int32, torch.uint8, torch.float32, torch.float64
names = ["extract", "undistort_grid", "bow", "query_setup", "window_search", "dense_match"]
acc = {k: 0.0 for k in names}
# handles first, then the lane streams back to back (stream -> hardware queue placement, NOTES.md §4.5)
exs = [capi.ORBextractor(nfeatures=a.nfeatures, max_batch=b) for _ in range(G)]
vocs = [capi.ORBVocabulary.from_nodes(a.voc_k, a.voc_l, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"]) for _ in range(G)]   # a handle owns scratch
raw = [capi.stream_create(0) for _ in range(G)]
cap = exs[0].max_keypoints


class Lane:
    """b consecutive frames of every step: frame t is searched / matched against the same slot of the previous step"""
    def __init__(self, g):
        self.g, self.ex, self.V = g, exs[g], vocs[g]
        self.stream = torch.cuda.ExternalStream(raw[g], device=torch.device("cuda", 0))
        z = lambda *shape, dt=i32: torch.zeros(shape, dtype=dt, device=dev)
        self.S = [dict(kps=z(b, cap, 7, dt=f32), desc=z(b, cap, 32, dt=u8), n=z(b), un=z(b, cap, 7, dt=f32), off=z(b, capi.GRID_CELLS + 1), feat=z(b, cap))
                  for _ in range(2)]
        self.bow_id, self.bow_val = z(b, cap), z(b, cap, dt=f64)
        self.fv_node, self.fv_off, self.fv_feat, self.cnt = z(b, cap), z(b, cap + 1), z(b, cap), z(2, b)
        self.qxyr, self.qlev, self.qang = z(b, cap, 3, dt=f32), z(b, cap, 2), z(b, cap, dt=f32)
        self.q2t, self.t2q, self.best, self.second, self.nm = z(b, cap), z(b, cap), z(b, cap), z(b, cap), z(b)
        self.m_idx, self.m_best, self.m_sec = z(b, cap), z(b, cap), z(b, cap)

    def step(self, i, timed):
        cur, prev = self.S[i & 1], self.S[(i + 1) & 1]
        f0 = (i * B) % (a.ring - B + 1) + self.g * b
        s = self.stream
        st = s.cuda_stream
        with torch.cuda.stream(s):
            ev = [torch.cuda.Event(enable_timing=True) for _ in range(len(names) + 1)] if timed else None
            mark = (lambda j: ev[j].record(s)) if timed else (lambda j: None)
            mark(0)
            self.ex.extract_batch_device(d_img.data_ptr() + f0 * w * h, b, w, h, w, w * h, cur["kps"].data_ptr(), cur["desc"].data_ptr(), cur["n"].data_ptr(), cap, 0, st)
            mark(1)
            capi.undistort_grid_batch_device(cam, bounds, cur["kps"].data_ptr(), cur["n"].data_ptr(), b, cap, cur["un"].data_ptr(), cur["off"].data_ptr(),
                                             cur["feat"].data_ptr(), st)
            mark(2)
            self.V.transform_batch_device(cur["desc"].data_ptr(), cur["n"].data_ptr(), b, cap, 4, self.bow_id.data_ptr(), self.bow_val.data_ptr(), self.cnt[0].data_ptr(),
                                          self.fv_node.data_ptr(), self.fv_off.data_ptr(), self.fv_feat.data_ptr(), self.cnt[1].data_ptr(), st)
            mark(3)
            # WindowSearch(last, current, window, ..): queries = the previous frame's undistorted keypoints at their own level
            self.qxyr[:, :, 0:2] = prev["un"][:, :, 0:2]
            self.qxyr[:, :, 2] = a.window
            oct_prev = prev["un"][:, :, 5].view(i32)
            self.qlev[:, :, 0] = oct_prev
            self.qlev[:, :, 1] = oct_prev
            self.qang.copy_(prev["un"][:, :, 3])
            mark(4)
            capi.window_search_batch_device(bounds, capi.RULE_WINDOW, capi.TH_HIGH, 0.8, True, cur["un"].data_ptr(), cur["desc"].data_ptr(), cur["off"].data_ptr(),
                                            cur["feat"].data_ptr(), cur["n"].data_ptr(), cap, 0, self.qxyr.data_ptr(), self.qlev.data_ptr(), prev["desc"].data_ptr(),
                                            self.qang.data_ptr(), 0, prev["n"].data_ptr(), cap, b, self.q2t.data_ptr(), self.t2q.data_ptr(), self.best.data_ptr(),
                                            self.second.data_ptr(), self.nm.data_ptr(), st)
            mark(5)
            capi.match_top2_batch_device(cur["desc"].data_ptr(), cur["n"].data_ptr(), prev["desc"].data_ptr(), prev["n"].data_ptr(), b, cap, self.m_idx.data_ptr(),
                                         self.m_best.data_ptr(), self.m_sec.data_ptr(), st)
            mark(6)
        if timed:
            torch.cuda.synchronize()      # stage times: this lane alone on the chip
            for j, k in enumerate(names):
                acc[k] += ev[j].elapsed_time(ev[j + 1])


lanes = [Lane(g) for g in range(G)]
torch.cuda.synchronize()


def step(i, timed=False):
    for ln in lanes:
        ln.step(i, timed)


for i in range(3):
    step(i)
torch.cuda.synchronize()
import time  # noqa: E402
t0 = time.perf_counter()
for i in range(3, 3 + a.steps):
    step(i)
torch.cuda.synchronize()
total_ms = (time.perf_counter() - t0) * 1e3 / a.steps
for i in range(3 + a.steps, 3 + 2 * a.steps):
    step(i, True)
cat = lambda f: torch.cat([f(ln) for ln in lanes])
out = {"metric": "frontend_frames_per_s", "value": round(B / (total_ms * 1e-3), 1), "unit": "frames/s", "ms_per_step": round(total_ms, 4),
       "config": {"family": synth.FAMILY_NAMES.get(a.family, str(a.family)), "workload": "%dx%d, %d kp, %d frames per step in %d lanes: extract + undistort/grid + BoW(k=%d,L=%d) + WindowSearch(r=%g, rot) + dense top-2" %
                  (w, h, a.nfeatures, B, G, a.voc_k, a.voc_l, a.window), "lanes": G},
       "stage_ms_per_step": {k: round(v / a.steps, 4) for k, v in acc.items()},
       "stage_timing": "every lane's stages alone on the chip (a serial pass after the timed loop), summed over the lanes",
       "mean_keypoints": round(float(cat(lambda ln: ln.S[0]["n"]).float().mean().item()), 1),
       "mean_window_matches": round(float(cat(lambda ln: ln.nm).float().mean().item()), 1),
       "mean_bow_words": round(float(cat(lambda ln: ln.cnt[0]).float().mean().item()), 1)}
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
