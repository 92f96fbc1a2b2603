"""Seeded inputs of the ORBmatcher golden fixtures (tests/golden/golden_matcher.npz): the problems are regenerated from seeds, only the
outputs of the REFERENCE's own src/ORBmatcher.cc are committed (tests/golden/make_golden_matcher.py).  Test infrastructure."""
import numpy as np

import kf_pairs
from orb_slam_amd import capi, synth

CAM = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.0, 0.0, 0.0, 0.0), 640, 480)
SCALE = np.array([np.float32(1.2) ** i for i in range(8)], np.float32)


def frame(rng, n, crowd):
    k = np.zeros(n, dtype=capi.KP_DTYPE)
    if crowd:
        cx, cy = rng.random(30) * 600 + 20, rng.random(30) * 440 + 20
        c = rng.integers(0, 30, n)
        k["x"] = (cx[c] + rng.normal(0, 8, n)).astype(np.float32); k["y"] = (cy[c] + rng.normal(0, 8, n)).astype(np.float32)
    else:
        k["x"] = (rng.random(n) * 640).astype(np.float32); k["y"] = (rng.random(n) * 480).astype(np.float32)
    k["angle"] = (rng.random(n) * 360).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    k["size"], k["class_id"] = 31, -1
    return k


def noisy_copies(rng, desc, src, flips, keep=0.85):
    q = synth.descriptors(len(src), int(rng.integers(1, 10**6)))
    d = desc[src].copy()
    for _ in range(flips):
        bit = rng.integers(0, 256, len(src))
        d[np.arange(len(src)), bit // 8] ^= (1 << (bit % 8)).astype(np.uint8)
    m = rng.random(len(src)) < keep
    q[m] = d[m]
    return q


def window_problem(seed, n1=800, n2=900, win=40, crowd=True):
    """two frames for WindowSearch / SearchForInitialization: F1's features are noisy copies of F2's"""
    rng = np.random.default_rng(seed)
    k2 = frame(rng, n2, crowd)
    k2["octave"] = rng.choice([0, 0, 1, 2, 3], n2)
    d2 = synth.descriptors(n2, seed + 200)
    src = rng.integers(0, n2, n1)
    k1 = k2[src].copy()
    k1["x"] += rng.normal(0, win / 4, n1).astype(np.float32); k1["y"] += rng.normal(0, win / 4, n1).astype(np.float32)
    k1["angle"] = ((k2["angle"][src] + rng.normal(12, 8, n1)) % 360).astype(np.float32)
    d1 = noisy_copies(rng, d2, src, 8)
    state1 = rng.choice([0, 1, 1, 1, 2], n1).astype(np.uint8)
    prev = np.stack([k2["x"][src] + rng.normal(0, win / 4, n1), k2["y"][src] + rng.normal(0, win / 4, n1)], -1).astype(np.float32)
    return dict(k1=k1, d1=d1, state1=state1, k2=k2, d2=d2, win=win, prev=prev)


def mappoint_problem(seed, nt=900, nq=1000, th=3.0):
    """a frame and projected map points for SearchByProjection(F, vpMapPoints, th)"""
    rng = np.random.default_rng(seed)
    k = frame(rng, nt, True)
    desc = synth.descriptors(nt, seed + 100)
    claimed = (rng.random(nt) < 0.2).astype(np.uint8)
    src = rng.integers(0, nt, nq)
    qlevel = np.clip(k["octave"][src] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
    qcos = np.where(rng.random(nq) < 0.5, np.float32(0.9995), np.float32(0.9)).astype(np.float32)
    r = np.where(qcos > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)
    r = (r * np.float32(th)).astype(np.float32)
    R = (r * SCALE[qlevel]).astype(np.float32)
    qxy = np.stack([k["x"][src] + rng.normal(0, 1, nq) * R / 3, k["y"][src] + rng.normal(0, 1, nq) * R / 3], -1).astype(np.float32)
    qdesc = noisy_copies(rng, desc, src, 6)
    qstate = rng.choice([0, 1, 1, 1, 1, 2], nq).astype(np.uint8)
    return dict(k=k, desc=desc, claimed=claimed, qlevel=qlevel, qcos=qcos, R=R, qxy=qxy, qdesc=qdesc, qstate=qstate, th=th)


def keyframe_pair(seed):
    pr = kf_pairs.pair(seed, 900, 1000, max_flips=30)
    rng = np.random.default_rng(seed + 7)
    pr["s1"] = rng.choice([0, 1, 1, 1, 2], 900).astype(np.uint8)          # 0 none / 1 good / 2 bad map point
    pr["s2"] = rng.choice([0, 1, 1, 1, 2], 1000).astype(np.uint8)
    return pr


VOC_ARGS = dict(k=10, L=4, seed=6)
LEVELSUP = 2
