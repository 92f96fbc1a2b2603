"""Pins the oracle's ORBmatcher restatements (oracle/search_oracle.cpp) to the reference's OWN src/ORBmatcher.cc, compiled where it
lies against plain-data stand-ins of Frame / KeyFrame / MapPoint (oracle/matcherstub, oracle/ref_orbmatcher_wrap.cpp ->
oracle/_ref/libref_orbmatcher.so).  Same seeded problems through both, every output compared:
SearchByProjection(Frame&, vector<MapPoint*>&, th), WindowSearch, SearchForInitialization, SearchByBoW (KeyFrame-Frame and
KeyFrame-KeyFrame), SearchForTriangulation with CheckDistEpipolarLine, ComputeThreeMaxima, DescriptorDistance.  No GPU."""
import ctypes
import os

import numpy as np
import pytest

import kf_pairs
import oracle_lib as ol
from orb_slam_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libref_orbmatcher.so")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="oracle/_ref/libref_orbmatcher.so is built only where /root/reference exists")

CAM = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.0, 0.0, 0.0, 0.0), 640, 480)
SCALE = np.array([np.float32(1.2) ** i for i in range(8)], np.float32)
_REF = None


def load(path):
    """the harness of oracle/ref_orbmatcher_wrap.cpp with its prototypes (libref_orbmatcher.so = in front of the reference's src/ORBmatcher.cc;
    libprod_orbmatcher.so = in front of the product's orb_slam_amd/cpp/ORBmatcher.cc: tests/test_gpu_orbmatcher_dropin.py)"""
    if True:
        L = ctypes.CDLL(path)
        f, i, vp = ctypes.c_float, ctypes.c_int, ctypes.c_void_p
        L.ref_matcher_three_maxima.argtypes = [vp, i, vp]
        L.ref_matcher_descriptor_distance.argtypes = [vp, vp]
        L.ref_matcher_check_epipolar.argtypes = [f, f, f, f, i, vp, vp, i]
        L.ref_search_by_projection_mappoints.argtypes = [vp, f, f, vp, vp, vp, vp, i, vp, vp, i, vp, vp, vp, vp, vp, i, vp]
        L.ref_window_search.argtypes = [vp, f, i, vp, vp, vp, i, vp, vp, vp, vp, i, i, i, i, vp]
        L.ref_search_for_initialization.argtypes = [vp, f, i, vp, vp, i, vp, vp, vp, vp, i, vp, i, vp]
        L.ref_search_by_projection_last_frame.argtypes = [vp, f, i, f, vp, vp, vp, vp, vp, i, vp, vp, i, vp, vp, vp, vp, vp, i, vp]
        L.ref_search_by_projection_two_frames.argtypes = [vp, f, vp, vp, vp, vp, vp, i, vp, vp, vp, vp, i, vp, i, vp]
        L.ref_search_by_projection_keyframe.argtypes = [vp, i, f, i, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, vp, i, vp]
        L.ref_search_by_projection_scw.argtypes = [vp, i, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, i, vp]
        L.ref_fuse.argtypes = [i, vp, f, vp, vp, i, vp, vp, vp, vp, i, vp, vp, vp, vp, vp, i, vp, vp]
        L.ref_search_by_sim3.argtypes = [vp, f, vp, vp, i] + [vp, vp, vp, vp, vp, vp, vp, i] * 2 + [vp]
        L.ref_search_by_bow.argtypes = [f, i] + [vp, vp, vp, i, vp, vp, vp, i] + [vp, vp, vp, i, vp, vp, i] + [vp]
        L.ref_search_by_bow_kf.argtypes = [f, i] + [vp, vp, vp, i, vp, vp, vp, i] * 2 + [vp]
        L.ref_search_for_triangulation.argtypes = [f, i, vp, vp, i] + [vp, vp, vp, i, vp, vp, vp, i] * 2 + [vp]
        L.ref_set_pose.argtypes = [vp, f]
        L.ref_set_sim3.argtypes = [vp, f]
    return L


def ref():
    global _REF
    if _REF is None:
        _REF = load(PATH)
    return _REF


def P(a):
    return a.ctypes.data


def _frame(rng, n, crowd):
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


def _noisy_copies(rng, desc, src, flips, keep=0.85):
    q = synth.descriptors(len(src), int(rng.integers(1, 10**6)))
    d = desc[src].copy()
    for _ in range(flips):
        bit = rng.integers(0, 256, len(src))
        d[np.arange(len(src)), bit // 8] ^= (1 << (bit % 8)).astype(np.uint8)
    m = rng.random(len(src)) < keep
    q[m] = d[m]
    return q


def test_three_maxima_and_descriptor_distance():
    rng = np.random.default_rng(1)
    for _ in range(300):
        h = rng.integers(0, int(rng.choice([2, 5, 40, 400])), 30).astype(np.int32)
        if rng.random() < 0.3:
            h[rng.integers(0, 30, 25)] = 0
        out = np.zeros(3, np.int32)
        ref().ref_matcher_three_maxima(P(h), 30, P(out))
        assert tuple(out) == ol.three_maxima(h), h
    d = synth.descriptors(400, 5)
    for i in range(0, 400, 2):
        assert ref().ref_matcher_descriptor_distance(P(d[i]), P(d[i + 1])) == ol.hamming256(d[i], d[i + 1])


def test_check_dist_epipolar_line():
    pr = kf_pairs.pair(9, 400, 400, line_noise=1.5)
    F = np.ascontiguousarray(pr["F"].reshape(9))
    s2 = kf_pairs.LEVEL_SIGMA2
    rng = np.random.default_rng(2)
    hits = 0
    for i in range(400):
        j = int(rng.integers(0, 400))
        a = ref().ref_matcher_check_epipolar(pr["k1"]["x"][j], pr["k1"]["y"][j], pr["k2"]["x"][i], pr["k2"]["y"][i], int(pr["k2"]["octave"][i]), P(F), P(s2), 8)
        assert bool(a) == ol.check_dist_epipolar_line(pr["k1"]["x"][j], pr["k1"]["y"][j], pr["k2"]["x"][i], pr["k2"]["y"][i], F, s2[pr["k2"]["octave"][i]])
        hits += a
    assert 5 < hits < 395


@pytest.mark.parametrize("seed,nt,nq,th,crowd", [(1, 1000, 1000, 1.0, False), (2, 800, 1200, 3.0, True), (3, 300, 50, 5.0, True), (4, 1, 30, 1.0, False), (5, 500, 0, 3.0, False)])
def test_search_by_projection_of_map_points(seed, nt, nq, th, crowd):
    rng = np.random.default_rng(seed)
    b = capi.image_bounds(CAM)
    k = _frame(rng, nt, crowd)
    desc = synth.descriptors(nt, seed + 100)
    if crowd:
        desc[rng.integers(0, nt, nt // 3)] = desc[0]
    off, feat = ol.frame_grid(b, k)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    claimed = (rng.random(nt) < 0.2).astype(np.uint8)
    src = rng.integers(0, nt, nq)
    qlevel = np.clip(k["octave"][src] + rng.integers(0, 2, nq), 0, 7).astype(np.int32)
    qcos = np.where(rng.random(nq) < 0.5, np.float32(0.9995), np.float32(0.9)).astype(np.float32)
    r = np.where(qcos > 0.998, np.float32(2.5), np.float32(4.0)).astype(np.float32)
    if th != 1.0:
        r = (r * np.float32(th)).astype(np.float32)
    R = (r * SCALE[qlevel]).astype(np.float32)
    qxy = np.stack([k["x"][src] + rng.normal(0, 1, nq) * R / 3, k["y"][src] + rng.normal(0, 1, nq) * R / 3], -1).astype(np.float32)
    qdesc = _noisy_copies(rng, desc, src, 6)
    qstate = rng.choice([0, 1, 1, 1, 1, 2], nq).astype(np.uint8)
    t2q = np.zeros(max(nt, 1), np.int32)
    n = ref().ref_search_by_projection_mappoints(ctypes.addressof(b), 0.8, th, P(k), P(desc), P(off), P(featp), nt,
                                                 P(claimed), P(SCALE), 8, P(qxy), P(qlevel), P(qcos), P(qdesc), P(qstate), nq, P(t2q))
    qxyr = np.concatenate([qxy, R[:, None]], 1)
    qlev = np.stack([qlevel - 1, qlevel], -1)
    w = ol.window_search(b, capi.RULE_MAPPOINTS, capi.TH_HIGH, 0.8, False, k, desc, off, feat, claimed, qxyr, qlev, qdesc, None, (qstate == 1).astype(np.uint8))
    got = t2q[:nt].copy(); got[got == -2] = -1
    assert n == w[0]
    np.testing.assert_array_equal(got, w[2])
    if nt > 100 and nq > 100:
        assert n > 50


@pytest.mark.parametrize("seed,n1,n2,win,check,lo,hi,crowd", [(11, 1000, 1000, 100, True, -1, 2**31 - 1, False), (12, 700, 1000, 15, True, 1, 5, True),
                                                               (13, 1000, 400, 50, False, 0, 3, True), (14, 2, 1, 100, True, -1, 2**31 - 1, False)])
def test_window_search(seed, n1, n2, win, check, lo, hi, crowd):
    rng = np.random.default_rng(seed)
    b = capi.image_bounds(CAM)
    k2 = _frame(rng, n2, crowd)
    d2 = synth.descriptors(n2, seed + 200)
    off, feat = ol.frame_grid(b, k2)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    src = rng.integers(0, n2, n1)
    k1 = k2[src].copy()
    k1["x"] += rng.normal(0, win / 4, n1).astype(np.float32); k1["y"] += rng.normal(0, win / 4, n1).astype(np.float32)
    k1["angle"] = ((k2["angle"][src] + rng.normal(12, 8, n1)) % 360).astype(np.float32)
    d1 = _noisy_copies(rng, d2, src, 8)
    state1 = rng.choice([0, 1, 1, 1, 2], n1).astype(np.uint8)
    t2q = np.zeros(max(n2, 1), np.int32)
    n = ref().ref_window_search(ctypes.addressof(b), 0.8, int(check), P(k1), P(d1), P(state1), n1, P(k2), P(d2), P(off), P(featp),
                                n2, win, lo, hi, P(t2q))
    lvl = k1["octave"]
    valid = (state1 == 1) & ~((lo > 0) & (lvl < lo)) & ~((hi < 2**31 - 1) & (lvl > hi))
    qxyr = np.stack([k1["x"], k1["y"], np.full(n1, win, np.float32)], -1)
    w = ol.window_search(b, capi.RULE_WINDOW, capi.TH_HIGH, 0.8, check, k2, d2, off, feat, None, qxyr, np.stack([lvl, lvl], -1), d1, k1["angle"], valid.astype(np.uint8))
    assert n == w[0]
    np.testing.assert_array_equal(t2q[:n2], w[2])
    if n1 > 500:
        assert n > 40


@pytest.mark.parametrize("seed,n1,n2,win,check,crowd", [(21, 1000, 1000, 100, True, False), (22, 900, 700, 30, False, True), (23, 1000, 1000, 200, True, True)])
def test_search_for_initialization(seed, n1, n2, win, check, crowd):
    rng = np.random.default_rng(seed)
    b = capi.image_bounds(CAM)
    k2 = _frame(rng, n2, crowd)
    k2["octave"] = rng.choice([0, 0, 0, 1, 2], n2)
    d2 = synth.descriptors(n2, seed + 300)
    off, feat = ol.frame_grid(b, k2)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    src = rng.integers(0, n2, n1)
    k1 = k2[src].copy()
    k1["angle"] = ((k2["angle"][src] + rng.normal(12, 8, n1)) % 360).astype(np.float32)
    k1["octave"] = rng.choice([0, 0, 0, 1], n1)
    d1 = _noisy_copies(rng, d2, src, 5)
    prev = np.stack([k2["x"][src] + rng.normal(0, win / 4, n1), k2["y"][src] + rng.normal(0, win / 4, n1)], -1).astype(np.float32)
    prev_in = prev.copy()
    q2t = np.zeros(max(n1, 1), np.int32)
    n = ref().ref_search_for_initialization(ctypes.addressof(b), 0.9, int(check), P(k1), P(d1), n1, P(k2), P(d2), P(off), P(featp),
                                            n2, P(prev), win, P(q2t))
    qxyr = np.concatenate([prev_in, np.full((n1, 1), win, np.float32)], 1)
    w = ol.window_search(b, capi.RULE_INIT, capi.TH_LOW, 0.9, check, k2, d2, off, feat, None, qxyr, np.zeros((n1, 2), np.int32), d1, k1["angle"],
                         (k1["octave"] == 0).astype(np.uint8))
    assert n == w[0] and n > 60
    np.testing.assert_array_equal(q2t[:n1], w[1])
    # vbPrevMatched is moved to the matched key point, untouched elsewhere
    m = q2t[:n1] >= 0
    np.testing.assert_array_equal(prev[m, 0], k2["x"][q2t[:n1][m]])
    np.testing.assert_array_equal(prev[~m], prev_in[~m])


@pytest.mark.parametrize("seed,n1,n2,th,check,crowd", [(61, 1000, 1000, 15.0, True, False), (62, 1000, 800, 7.0, True, True), (63, 600, 1000, 15.0, False, True)])
def test_search_by_projection_from_last_frame(seed, n1, n2, th, check, crowd):
    """SearchByProjection(CurrentFrame, LastFrame, th) through the reference's own projection code (identity pose, depth 1)"""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    b = capi.image_bounds(CAM)
    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    k2 = _frame(rng, n2, crowd)
    d2 = synth.descriptors(n2, seed + 400)
    off, feat = ol.frame_grid(b, k2)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    claimed = (rng.random(n2) < 0.15).astype(np.uint8)
    src = rng.integers(0, n2, n1)
    k1 = k2[src].copy()
    k1["angle"] = ((k2["angle"][src] + rng.normal(10, 8, n1)) % 360).astype(np.float32)
    k1["octave"] = np.clip(k2["octave"][src] + rng.integers(-1, 2, n1), 0, 7)
    d1 = _noisy_copies(rng, d2, src, 6)
    tu = k2["x"][src] + rng.normal(0, th / 2, n1); tv = k2["y"][src] + rng.normal(0, th / 2, n1)
    tu[:20] = rng.choice([-30.0, 700.0], 20)                                  # some projections outside the image bounds
    world = np.stack([((tu - cx) / fx), ((tv - cy) / fy), np.ones(n1)], -1).astype(np.float32)
    state1 = (rng.random(n1) < 0.85).astype(np.uint8)
    outlier = (rng.random(n1) < 0.1).astype(np.uint8)
    cam = np.array([fx, fy, cx, cy], np.float32)
    t2q = np.zeros(max(n2, 1), np.int32)
    n = ref().ref_search_by_projection_last_frame(ctypes.addressof(b), 0.9, int(check), th, P(cam), P(k2), P(d2), P(off), P(featp), n2, P(claimed), P(SCALE), 8,
                                                  P(k1), P(d1), P(world), P(state1), P(outlier), n1, P(t2q))
    # the reference's float expressions: u = fx*xc*invzc + cx with invzc = (float)(1.0 / 1.0f)
    u = (fx * world[:, 0] * f32(1.0) + cx).astype(np.float32); v = (fy * world[:, 1] * f32(1.0) + cy).astype(np.float32)
    inb = ~((u < b.min_x) | (u > b.max_x)) & ~((v < b.min_y) | (v > b.max_y))
    valid = (state1 == 1) & (outlier == 0) & inb
    oct1 = k1["octave"]
    qxyr = np.stack([u, v, (f32(th) * SCALE[oct1]).astype(np.float32)], -1)
    w = ol.window_search(b, capi.RULE_BEST, capi.TH_HIGH, 0.9, check, k2, d2, off, feat, claimed, qxyr, np.stack([oct1 - 1, oct1 + 1], -1), d1, k1["angle"],
                         valid.astype(np.uint8))
    got = t2q[:n2].copy(); got[got == -2] = -1
    assert n == w[0] and n > 100 and (~inb).sum() >= 10
    np.testing.assert_array_equal(got, w[2])


@pytest.mark.parametrize("seed,n1,n2,th,crowd", [(71, 1000, 1000, 7.5, False), (72, 800, 1000, 10.0, True), (73, 1000, 500, 4.0, True)])
def test_search_by_sim3(seed, n1, n2, th, crowd):
    """SearchBySim3 through the reference's own code (identity poses / similarity): its two scans are two searches of the oracle's
    rule 5 (window + levels [predicted-1, predicted], best <= TH_HIGH), its tail the agreement check"""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    b = capi.image_bounds(CAM)
    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    cam = np.array([fx, fy, cx, cy], np.float32)
    k1 = _frame(rng, n1, crowd)
    d1 = synth.descriptors(n1, seed + 500)
    src = rng.integers(0, n1, n2)
    k2 = k1[src].copy()
    k2["x"] += rng.normal(0, 1.5, n2).astype(np.float32); k2["y"] += rng.normal(0, 1.5, n2).astype(np.float32)
    d2 = _noisy_copies(rng, d1, src, 6)

    def side(k, n):
        lv = np.clip(k["octave"] + rng.integers(0, 2, n), 1, 9)                  # predicted level (9 = beyond the last factor: clipped to 7)
        tu, tv = k["x"] + rng.normal(0, th / 3, n), k["y"] + rng.normal(0, th / 3, n)
        world = np.stack([(tu - cx) / fx, (tv - cy) / fy, np.ones(n)], -1).astype(np.float32)
        dist = np.sqrt((world.astype(np.float64) ** 2).sum(1))
        sc = np.concatenate([SCALE.astype(np.float64), [SCALE[7] * 1.2, SCALE[7] * 1.44, SCALE[7] * 1.7]])
        mind = (dist / np.sqrt(sc[lv - 1] * sc[lv])).astype(np.float32)            # ratio = dist / mind lies between two scale factors
        state = rng.choice([0, 1, 1, 1, 1, 2], n).astype(np.uint8)
        u = (fx * (world[:, 0] * f32(1.0)) + cx).astype(np.float32); v = (fy * (world[:, 1] * f32(1.0)) + cy).astype(np.float32)
        inimg = (u >= b.min_x) & (u < b.max_x) & (v >= b.min_y) & (v < b.max_y)
        lvl = np.minimum(lv, 7)
        return dict(world=world, mind=mind, state=state, u=u, v=v, valid=(state == 1) & inimg, lvl=lvl, rad=(f32(th) * SCALE[lvl]).astype(np.float32))

    s1, s2 = side(k1, n1), side(k2, n2)
    off1, feat1 = ol.frame_grid(b, k1); off2, feat2 = ol.frame_grid(b, k2)
    fp1 = np.ascontiguousarray(np.append(feat1, 0).astype(np.int32)); fp2 = np.ascontiguousarray(np.append(feat2, 0).astype(np.int32))
    m12 = np.zeros(max(n1, 1), np.int32)
    n = ref().ref_search_by_sim3(ctypes.addressof(b), th, P(cam), P(SCALE), 8, P(k1), P(d1), P(off1), P(fp1), P(s1["state"]), P(s1["world"]), P(s1["mind"]), n1,
                                 P(k2), P(d2), P(off2), P(fp2), P(s2["state"]), P(s2["world"]), P(s2["mind"]), n2, P(m12))

    def scan(q, qd, kt, dt, offt, featt):
        qxyr = np.stack([q["u"], q["v"], q["rad"]], -1)
        return ol.window_search(b, capi.RULE_FREE, capi.TH_HIGH, 0.0, False, kt, dt, offt, featt, None, qxyr, np.stack([q["lvl"] - 1, q["lvl"]], -1), qd, None,
                                q["valid"].astype(np.uint8))[1]
    a12 = scan(s1, d1, k2, d2, off2, feat2)
    a21 = scan(s2, d2, k1, d1, off1, feat1)
    nf, want = ol.sim3_agreement(a12, a21)
    # the reference reports the pKF2 MAP POINT; features of pKF2 without one cannot be reported
    want = np.where((want >= 0) & (s2["state"][np.maximum(want, 0)] != 0), want, -1)
    got = m12[:n1]
    assert (a12 >= 0).sum() > 150 and (a21 >= 0).sum() > 100
    np.testing.assert_array_equal(got, want)
    assert n == nf and n > 40


@pytest.mark.parametrize("seed,n1,n2,win,crowd", [(81, 1000, 1000, 30, False), (82, 700, 1000, 10, True), (83, 1000, 500, 60, True)])
def test_search_by_projection_between_two_frames(seed, n1, n2, win, crowd):
    """SearchByProjection(F1, F2, windowSize, vpMapPointMatches2): WindowSearch's rule behind the reference's own projection (identity pose)"""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    b = capi.image_bounds(CAM)
    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    cam = np.array([fx, fy, cx, cy], np.float32)
    k2 = _frame(rng, n2, crowd)
    d2 = synth.descriptors(n2, seed + 600)
    off, feat = ol.frame_grid(b, k2)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    claimed = (rng.random(n2) < 0.15).astype(np.uint8)
    src = rng.integers(0, n2, n1)
    k1 = k2[src].copy()
    d1 = _noisy_copies(rng, d2, src, 8)
    tu, tv = k2["x"][src] + rng.normal(0, win / 4, n1), k2["y"][src] + rng.normal(0, win / 4, n1)
    world = np.stack([(tu - cx) / fx, (tv - cy) / fy, np.ones(n1)], -1).astype(np.float32)
    state1 = rng.choice([0, 1, 1, 1, 2], n1).astype(np.uint8)
    t2q = np.zeros(max(n2, 1), np.int32)
    n = ref().ref_search_by_projection_two_frames(ctypes.addressof(b), 0.8, P(cam), P(k1), P(d1), P(state1), P(world), n1, P(k2), P(d2), P(off), P(featp), n2,
                                                  P(claimed), win, P(t2q))
    u = (fx * world[:, 0] * f32(1.0) + cx).astype(np.float32); v = (fy * world[:, 1] * f32(1.0) + cy).astype(np.float32)
    lvl = k1["octave"]
    w = ol.window_search(b, capi.RULE_WINDOW, capi.TH_HIGH, 0.8, False, k2, d2, off, feat, claimed, np.stack([u, v, np.full(n1, win, np.float32)], -1),
                         np.stack([lvl, lvl], -1), d1, None, (state1 == 1).astype(np.uint8))
    got = t2q[:n2].copy(); got[got == -2] = -1
    assert n == w[0] and n > 40
    np.testing.assert_array_equal(got, w[2])


@pytest.mark.parametrize("seed,nkf,n2,th,orbdist,check,crowd", [(91, 1000, 1000, 10.0, 100, True, False), (92, 800, 1000, 3.0, 64, True, True), (93, 1000, 600, 10.0, 100, False, True)])
def test_search_by_projection_from_keyframe(seed, nkf, n2, th, orbdist, check, crowd):
    """SearchByProjection(CurrentFrame, pKF, sAlreadyFound, th, ORBdist) (relocalisation): rule 2 with the level predicted from the distance"""
    rng = np.random.default_rng(seed)
    f32 = np.float32
    b = capi.image_bounds(CAM)
    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    cam = np.array([fx, fy, cx, cy], np.float32)
    k2 = _frame(rng, n2, crowd)
    d2 = synth.descriptors(n2, seed + 700)
    off, feat = ol.frame_grid(b, k2)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    claimed = (rng.random(n2) < 0.15).astype(np.uint8)
    src = rng.integers(0, n2, nkf)
    kk = k2[src].copy()
    kk["angle"] = ((k2["angle"][src] + rng.normal(10, 8, nkf)) % 360).astype(np.float32)
    dk = _noisy_copies(rng, d2, src, 6)
    lv = np.clip(k2["octave"][src] + rng.integers(-1, 2, nkf), 1, 9)
    tu, tv = k2["x"][src] + rng.normal(0, th, nkf), k2["y"][src] + rng.normal(0, th, nkf)
    tu[:15] = rng.choice([-40.0, 720.0], 15)
    world = np.stack([(tu - cx) / fx, (tv - cy) / fy, np.ones(nkf)], -1).astype(np.float32)
    dist = np.sqrt((world.astype(np.float64) ** 2).sum(1))
    sc = np.concatenate([SCALE.astype(np.float64), [SCALE[7] * 1.2, SCALE[7] * 1.44, SCALE[7] * 1.7]])
    mind = (dist / np.sqrt(sc[lv - 1] * sc[lv])).astype(np.float32)
    state = rng.choice([0, 1, 1, 1, 1, 2, 3], nkf).astype(np.uint8)
    t2q = np.zeros(max(n2, 1), np.int32)
    n = ref().ref_search_by_projection_keyframe(ctypes.addressof(b), int(check), th, orbdist, P(cam), P(SCALE), 8, P(k2), P(d2), P(off), P(featp), n2, P(claimed),
                                                P(kk), P(dk), P(state), P(world), P(mind), nkf, P(t2q))
    u = (fx * world[:, 0] * f32(1.0) + cx).astype(np.float32); v = (fy * world[:, 1] * f32(1.0) + cy).astype(np.float32)
    inb = ~((u < b.min_x) | (u > b.max_x)) & ~((v < b.min_y) | (v > b.max_y))
    lvl = np.minimum(lv, 7)
    qxyr = np.stack([u, v, (f32(th) * SCALE[lvl]).astype(np.float32)], -1)
    w = ol.window_search(b, capi.RULE_BEST, orbdist, 0.0, check, k2, d2, off, feat, claimed, qxyr, np.stack([lvl - 1, lvl + 1], -1), dk, kk["angle"],
                         ((state == 1) & inb).astype(np.uint8))
    got = t2q[:n2].copy(); got[got == -2] = -1
    assert n == w[0] and n > 60
    np.testing.assert_array_equal(got, w[2])


def _projected_queries(rng, k, src, nq, th, b, zero_state=False):
    """map points at depth 1 in front of an identity-pose key frame, predicted level chosen through their minimum distance"""
    f32 = np.float32
    fx, fy, cx, cy = f32(517.3), f32(516.5), f32(318.6), f32(255.3)
    lv = np.clip(k["octave"][src] + rng.integers(0, 2, nq), 1, 9)
    tu, tv = k["x"][src] + rng.normal(0, th / 2, nq), k["y"][src] + rng.normal(0, th / 2, nq)
    tu[:12] = rng.choice([-40.0, 720.0], 12)
    world = np.stack([(tu - cx) / fx, (tv - cy) / fy, np.ones(nq)], -1).astype(np.float32)
    dist = np.sqrt((world.astype(np.float64) ** 2).sum(1))
    sc = np.concatenate([SCALE.astype(np.float64), [SCALE[7] * 1.2, SCALE[7] * 1.44, SCALE[7] * 1.7]])
    mind = (dist / np.sqrt(sc[lv - 1] * sc[lv])).astype(np.float32)
    u = (fx * (world[:, 0] * f32(1.0)) + cx).astype(np.float32); v = (fy * (world[:, 1] * f32(1.0)) + cy).astype(np.float32)
    inimg = (u >= b.min_x) & (u < b.max_x) & (v >= b.min_y) & (v < b.max_y)
    lvl = np.minimum(lv, 7)
    qxyr = np.stack([u, v, (f32(th) * SCALE[lvl]).astype(np.float32)], -1)
    return world, mind, inimg, qxyr, np.stack([lvl - 1, lvl], -1), np.array([fx, fy, cx, cy], np.float32)


@pytest.mark.parametrize("seed,nkf,nq,th,crowd", [(101, 1000, 1000, 10, False), (102, 1000, 600, 4, True), (103, 500, 1000, 10, True)])
def test_search_by_projection_with_sim3_pose(seed, nkf, nq, th, crowd):
    """SearchByProjection(pKF, Scw, vpPoints, vpMatched, th) (loop closing): features matched earlier are skipped, a match claims its
    feature -> the in-order rule 2 without rotation check, TH_LOW"""
    rng = np.random.default_rng(seed)
    b = capi.image_bounds(CAM)
    k = _frame(rng, nkf, crowd)
    d = synth.descriptors(nkf, seed + 800)
    off, feat = ol.frame_grid(b, k)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    claimed = (rng.random(nkf) < 0.15).astype(np.uint8)
    src = rng.integers(0, nkf, nq)
    qd = _noisy_copies(rng, d, src, 5)
    world, mind, inimg, qxyr, qlev, cam = _projected_queries(rng, k, src, nq, float(th), b)
    state = rng.choice([1, 1, 1, 1, 2, 3], nq).astype(np.uint8)
    state[(state == 3) & (np.cumsum(state == 3) > claimed.sum())] = 1           # at most as many "already found" points as claimed slots
    t2q = np.zeros(max(nkf, 1), np.int32)
    n = ref().ref_search_by_projection_scw(ctypes.addressof(b), th, P(cam), P(SCALE), 8, P(k), P(d), P(off), P(featp), nkf, P(claimed), P(state), P(world), P(mind),
                                           P(qd), nq, P(t2q))
    w = ol.window_search(b, capi.RULE_BEST, capi.TH_LOW, 0.0, False, k, d, off, feat, claimed, qxyr, qlev, qd, None, ((state == 1) & inimg).astype(np.uint8))
    got = t2q[:nkf].copy(); got[got == -2] = -1
    assert n == w[0] and n > 60
    np.testing.assert_array_equal(got, w[2])


@pytest.mark.parametrize("which", [0, 1], ids=["Fuse(pKF, vpMapPoints, th)", "Fuse(pKF, Scw, vpPoints, th)"])
@pytest.mark.parametrize("seed,nkf,nq,th,crowd", [(111, 1000, 1000, 2.5, False), (112, 1000, 700, 4.0, True), (113, 400, 1000, 2.5, True)])
def test_fuse(which, seed, nkf, nq, th, crowd):
    """both Fuse overloads: every map point scans its window on its own (nothing is claimed) -> rule 5 with TH_LOW; the sequence of
    feature indices the reference fuses into, in query order, is the oracle's"""
    rng = np.random.default_rng(seed)
    b = capi.image_bounds(CAM)
    k = _frame(rng, nkf, crowd)
    d = synth.descriptors(nkf, seed + 900)
    off, feat = ol.frame_grid(b, k)
    featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    kf_state = rng.choice([0, 0, 1, 2], nkf).astype(np.uint8)
    src = rng.integers(0, nkf, nq)
    qd = _noisy_copies(rng, d, src, 5)
    world, mind, inimg, qxyr, qlev, cam = _projected_queries(rng, k, src, nq, th, b)
    state = rng.choice([0, 1, 1, 1, 1, 2] if which == 0 else [1, 1, 1, 1, 2], nq).astype(np.uint8)
    log = np.zeros(nq + 1, np.int32); nlog = ctypes.c_int()
    n = ref().ref_fuse(which, ctypes.addressof(b), th, P(cam), P(SCALE), 8, P(k), P(d), P(off), P(featp), nkf, P(kf_state), P(state), P(world), P(mind), P(qd), nq,
                       P(log), ctypes.addressof(nlog))
    w = ol.window_search(b, capi.RULE_FREE, capi.TH_LOW, 0.0, False, k, d, off, feat, None, qxyr, qlev, qd, None, ((state == 1) & inimg).astype(np.uint8))
    want = w[1][w[1] >= 0]
    assert n == len(want) == nlog.value and n > 60
    np.testing.assert_array_equal(log[:n], want)


def _fvs(pr, levelsup=2):
    OV = ol.OracleVocabulary(voc=synth.vocabulary(8, 3, seed=4))
    t1, t2 = OV.transform(pr["d1"], levelsup), OV.transform(pr["d2"], levelsup)
    c = lambda t: (np.ascontiguousarray(t[2], np.uint32), np.ascontiguousarray(t[3], np.int32), np.ascontiguousarray(t[4], np.uint32))
    return c(t1), c(t2)


@pytest.mark.parametrize("seed,n1,n2,check", [(31, 1000, 1000, True), (32, 600, 1000, False), (33, 1000, 300, True), (34, 1, 50, True)])
def test_search_by_bow_keyframe_frame(seed, n1, n2, check):
    pr = kf_pairs.pair(seed, n1, n2, max_flips=40)
    fv1, fv2 = _fvs(pr)
    rng = np.random.default_rng(seed)
    state = rng.choice([0, 1, 1, 1, 2], n1).astype(np.uint8)
    a1, a2 = np.ascontiguousarray(pr["k1"]["angle"]), np.ascontiguousarray(pr["k2"]["angle"])
    t2q = np.zeros(max(n2, 1), np.int32)
    n = ref().ref_search_by_bow(0.7, int(check), P(fv1[0]), P(fv1[1]), P(fv1[2]), len(fv1[0]), P(pr["d1"]), P(a1), P(state), n1,
                                P(fv2[0]), P(fv2[1]), P(fv2[2]), len(fv2[0]), P(pr["d2"]), P(a2), n2, P(t2q))
    w = ol.search_by_bow(capi.TH_LOW, 0.7, check, fv1, pr["d1"], a1, (state == 1).astype(np.uint8), fv2, pr["d2"], a2)
    assert n == w[0]
    np.testing.assert_array_equal(t2q[:n2], w[2])
    if n1 >= 600:
        assert n > 30


@pytest.mark.parametrize("seed,n1,n2,check", [(41, 1000, 1000, True), (42, 1000, 500, False), (43, 400, 1000, True)])
def test_search_by_bow_keyframe_keyframe(seed, n1, n2, check):
    pr = kf_pairs.pair(seed, n1, n2, max_flips=40)
    fv1, fv2 = _fvs(pr)
    rng = np.random.default_rng(seed)
    s1, s2 = rng.choice([0, 1, 1, 1, 2], n1).astype(np.uint8), rng.choice([0, 1, 1, 1, 2], n2).astype(np.uint8)
    a1, a2 = np.ascontiguousarray(pr["k1"]["angle"]), np.ascontiguousarray(pr["k2"]["angle"])
    q2t = np.zeros(max(n1, 1), np.int32)
    n = ref().ref_search_by_bow_kf(0.75, int(check), P(fv1[0]), P(fv1[1]), P(fv1[2]), len(fv1[0]), P(pr["d1"]), P(a1), P(s1), n1,
                                   P(fv2[0]), P(fv2[1]), P(fv2[2]), len(fv2[0]), P(pr["d2"]), P(a2), P(s2), n2, P(q2t))
    w = ol.search_by_bow_kf(capi.TH_LOW, 0.75, check, fv1, pr["d1"], a1, (s1 == 1).astype(np.uint8), fv2, pr["d2"], a2, (s2 == 1).astype(np.uint8))
    assert n == w[0] and n > 15
    np.testing.assert_array_equal(q2t[:n1], w[1])


@pytest.mark.parametrize("seed,n1,n2,check,kw", [(51, 1000, 1000, True, {}), (52, 1000, 700, False, dict(line_noise=1.0)), (53, 500, 1000, True, dict(max_flips=40)),
                                                  (54, 1000, 1000, True, dict(p_mp1=0.0, p_mp2=0.0, line_noise=3.0)), (55, 1, 300, True, {})])
def test_search_for_triangulation(seed, n1, n2, check, kw):
    pr = kf_pairs.pair(seed, n1, n2, **kw)
    fv1, fv2 = _fvs(pr)
    F = np.ascontiguousarray(pr["F"].reshape(9))
    q2t = np.zeros(max(n1, 1), np.int32)
    n = ref().ref_search_for_triangulation(0.6, int(check), P(F), P(kf_pairs.LEVEL_SIGMA2), 8, P(fv1[0]), P(fv1[1]), P(fv1[2]), len(fv1[0]), P(pr["k1"]), P(pr["d1"]),
                                           P(pr["mp1"]), n1, P(fv2[0]), P(fv2[1]), P(fv2[2]), len(fv2[0]), P(pr["k2"]), P(pr["d2"]), P(pr["mp2"]), n2, P(q2t))
    w = ol.search_for_triangulation(capi.TH_LOW, check, F, kf_pairs.LEVEL_SIGMA2, fv1, pr["k1"], pr["d1"], pr["mp1"], fv2, pr["k2"], pr["d2"], pr["mp2"])
    assert n == w[0]
    np.testing.assert_array_equal(q2t[:n1], w[1])
    if n1 >= 500:
        assert n > 30


def test_distinctive_descriptor():
    """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:185-250) through the reference's own MapPoint.cc: the descriptor it
    keeps is the row orc_distinctive picks (first row with the smallest median on ties; bad key frames skipped)"""
    path = os.path.join(ROOT, "oracle", "_ref", "libref_mappoint.so")
    if not os.path.exists(path):
        pytest.skip("oracle/_ref/libref_mappoint.so is built only where /root/reference exists")
    L = ctypes.CDLL(path)
    L.ref_distinctive.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    rng = np.random.default_rng(4)
    checked = 0
    for n in [1, 2, 3, 4, 5, 8, 17, 40, 64]:
        for rep in range(6):
            d = synth.descriptors(n, 900 + 10 * n + rep)
            if n > 3 and rep % 2:
                d[rng.integers(0, n, n // 2)] = d[0]                   # duplicates: ties between rows
            bad = (rng.random(n) < (0.3 if rep >= 4 else 0.0)).astype(np.uint8)
            out = np.zeros(32, np.uint8)
            ok = L.ref_distinctive(P(d), n, P(bad), P(out))
            good = d[bad == 0]
            if len(good) == 0:
                assert ok == 0
                continue
            idx, med = ol.distinctive(good)
            assert ok == 1 and np.array_equal(out, good[idx]), (n, rep, idx, med)
            checked += 1
    assert checked > 40
