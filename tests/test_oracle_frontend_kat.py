"""Known-answer tests of the UNPINNED parts of the front-end oracle (oracle/frame_oracle.cpp, oracle/search_oracle.cpp) against
independent pure-Python / numpy statements written from the reference text, on small inputs.  (The bag-of-words oracle needs
none: it is pinned to the reference's own DBoW2, tests/test_ref_pin.py.)"""
import math

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import capi, synth

CAM = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), 640, 480)


def _undistort_np(cam, pts):
    """cvUndistortPoints (OpenCV 2.4 undistort.cpp) with R = I, P = K, vectorised in float64"""
    K = np.array(list(cam.K), np.float32).astype(np.float64).reshape(3, 3)
    k = np.zeros(8)
    k[: cam.ndist] = np.array(list(cam.dist)[: cam.ndist], np.float32).astype(np.float64)
    fx, fy, cx, cy = K[0, 0], K[1, 1], K[0, 2], K[1, 2]
    ifx, ify = 1.0 / fx, 1.0 / fy
    x = (pts[:, 0].astype(np.float64) - cx) * ifx
    y = (pts[:, 1].astype(np.float64) - cy) * ify
    x0, y0 = x.copy(), y.copy()
    for _ in range(5):
        r2 = x * x + y * y
        icdist = (1 + ((k[7] * r2 + k[6]) * r2 + k[5]) * r2) / (1 + ((k[4] * r2 + k[1]) * r2 + k[0]) * r2)
        dx = 2 * k[2] * x * y + k[3] * (r2 + 2 * x * x)
        dy = k[2] * (r2 + 2 * y * y) + 2 * k[3] * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    xx = K[0, 0] * x + K[0, 1] * y + K[0, 2]
    yy = K[1, 0] * x + K[1, 1] * y + K[1, 2]
    ww = 1.0 / (K[2, 0] * x + K[2, 1] * y + K[2, 2])
    return np.stack([(xx * ww).astype(np.float32), (yy * ww).astype(np.float32)], -1)


def test_undistort_matches_numpy_statement():
    rng = np.random.default_rng(2)
    k = np.zeros(5000, dtype=ol.KP_DTYPE)
    k["x"] = (rng.random(5000) * 700 - 30).astype(np.float32)
    k["y"] = (rng.random(5000) * 540 - 30).astype(np.float32)
    un = ol.frame_undistort(CAM, k)
    want = _undistort_np(CAM, np.stack([k["x"], k["y"]], -1))
    assert un["x"].tobytes() == want[:, 0].tobytes() and un["y"].tobytes() == want[:, 1].tobytes()
    b = ol.frame_bounds(CAM, capi.Bounds)
    corners = _undistort_np(CAM, np.array([[0, 0], [640, 0], [0, 480], [640, 480]], np.float32))
    assert b.min_x == min(math.floor(corners[0, 0]), math.floor(corners[2, 0])) and b.max_x == max(math.ceil(corners[1, 0]), math.ceil(corners[3, 0]))
    assert b.min_y == min(math.floor(corners[0, 1]), math.floor(corners[1, 1])) and b.max_y == max(math.ceil(corners[2, 1]), math.ceil(corners[3, 1]))


def _grid_py(b, kps):
    """Frame.cc:108-123 / :267-277 in plain Python"""
    grid = [[[] for _ in range(48)] for _ in range(64)]
    f32 = np.float32
    for i, kp in enumerate(kps):
        vx = float(f32(f32(kp["x"] - f32(b.min_x)) * f32(b.inv_w)))
        px = int(math.floor(vx + 0.5)) if vx >= 0 else -int(math.floor(-vx + 0.5))          # round(): half away from zero
        vy = float(f32(f32(kp["y"] - f32(b.min_y)) * f32(b.inv_h)))
        py = int(math.floor(vy + 0.5)) if vy >= 0 else -int(math.floor(-vy + 0.5))          # round(): half away from zero
        if 0 <= px < 64 and 0 <= py < 48:
            grid[px][py].append(i)
    return grid


def _area_py(b, kps, grid, x, y, r, lo, hi):
    """Frame.cc:200-265 in plain Python (float32 arithmetic as in the reference's float expressions)"""
    f32 = np.float32
    x, y, r = f32(x), f32(y), f32(r)
    out = []
    c0 = max(0, math.floor(float(f32(f32(x - f32(b.min_x)) - r) * f32(b.inv_w))))
    if c0 >= 64:
        return out
    c1 = min(63, math.ceil(float(f32(f32(x - f32(b.min_x)) + r) * f32(b.inv_w))))
    if c1 < 0:
        return out
    r0 = max(0, math.floor(float(f32(f32(y - f32(b.min_y)) - r) * f32(b.inv_h))))
    if r0 >= 48:
        return out
    r1 = min(47, math.ceil(float(f32(f32(y - f32(b.min_y)) + r) * f32(b.inv_h))))
    if r1 < 0:
        return out
    check = not (lo == -1 and hi == -1)
    same = check and lo == hi
    for ix in range(c0, c1 + 1):
        for iy in range(r0, r1 + 1):
            for i in grid[ix][iy]:
                o = int(kps[i]["octave"])
                if check and not same and (o < lo or o > hi):
                    continue
                if same and o != lo:
                    continue
                if abs(f32(kps[i]["x"]) - x) > r or abs(f32(kps[i]["y"]) - y) > r:
                    continue
                out.append(i)
    return out


def _ham(a, b):
    return int(np.unpackbits(a ^ b).sum())


def _search_py(b, rule, th, ratio, check, kps, desc, grid, claimed_in, qxyr, qlev, qdesc, qangle, qvalid):
    """the greedy searches of src/ORBmatcher.cc in plain Python (rule numbering of include/orbs.h)"""
    f32 = np.float32
    nt, nq = len(kps), len(qxyr)
    claimed = [bool(c) for c in claimed_in] if claimed_in is not None else [False] * nt
    matched_dist = [2**31 - 1] * nt
    q2t, t2q = [-1] * nq, [-1] * nt
    best_o, second_o = [-1] * nq, [-1] * nq
    hist = [[] for _ in range(30)]
    n = 0

    def rot_bin(a1, a2):
        rot = f32(a1) - f32(a2)
        if rot < 0:
            rot = f32(rot + f32(360.0))
        v = float(f32(rot * f32(f32(1.0) / f32(30))))
        bn = int(math.floor(v + 0.5))
        return 0 if bn == 30 else bn

    for q in range(nq):
        if qvalid is not None and not qvalid[q]:
            continue
        near = _area_py(b, kps, grid, qxyr[q][0], qxyr[q][1], qxyr[q][2], int(qlev[q][0]), int(qlev[q][1]))
        if not near:
            continue
        bd, bd2, bi, bl, bl2 = 2**31 - 1, 2**31 - 1, -1, -1, -1
        for idx in near:
            if rule not in (3, 5) and claimed[idx]:
                continue
            d = _ham(qdesc[q], desc[idx])
            if rule == 3 and matched_dist[idx] <= d:
                continue
            if d < bd:
                bd2, bd, bl2, bl, bi = bd, d, bl, int(kps[idx]["octave"]), idx
            elif d < bd2:
                bl2, bd2 = int(kps[idx]["octave"]), d
        best_o[q], second_o[q] = bd, bd2
        if rule == 0:
            acc = bd <= th and not (bl == bl2 and f32(bd) > f32(ratio) * f32(bd2))
        elif rule == 1:
            acc = f32(bd) <= f32(bd2) * f32(ratio) and bd <= th
        elif rule in (2, 5):
            acc = bd <= th
        else:
            acc = bd <= th and f32(bd) < f32(bd2) * f32(ratio)
        if not acc:
            continue
        if rule == 3:
            if t2q[bi] >= 0:
                q2t[t2q[bi]] = -1
                n -= 1
            q2t[q], t2q[bi], matched_dist[bi] = bi, q, bd
            n += 1
            if check:
                hist[rot_bin(qangle[q], kps[bi]["angle"])].append(q)
        elif rule == 5:
            q2t[q] = bi
            n += 1
        else:
            claimed[bi] = True
            q2t[q], t2q[bi] = bi, q
            n += 1
            if check and rule != 0:
                hist[rot_bin(qangle[q], kps[bi]["angle"])].append(bi)
    if check and rule not in (0, 5):
        sizes = [len(h) for h in hist]
        keep = set(ol.three_maxima(sizes))          # ComputeThreeMaxima has its own KAT (tests/test_abi.py)
        for i in range(30):
            if i in keep:
                continue
            for e in hist[i]:
                if rule == 3:
                    if q2t[e] >= 0:
                        t2q[q2t[e]] = -1
                        q2t[e] = -1
                        n -= 1
                else:
                    q2t[t2q[e]] = -1
                    t2q[e] = -1
                    n -= 1
    return n, q2t, t2q, best_o, second_o


@pytest.mark.parametrize("rule,th,ratio,check", [(0, 100, 0.8, False), (1, 100, 0.8, True), (2, 100, 0.9, True), (3, 50, 0.9, True), (5, 50, 0.9, False), (1, 100, 0.9, False)])
def test_grid_area_and_searches_match_plain_python(rule, th, ratio, check):
    rng = np.random.default_rng(40 + rule)
    nt, nq = 260, 300
    k = np.zeros(nt, dtype=ol.KP_DTYPE)
    k["x"] = (rng.random(nt) * 660 - 10).astype(np.float32)
    k["y"] = (rng.random(nt) * 500 - 10).astype(np.float32)
    k["x"][:120] = 300 + rng.normal(0, 12, 120)            # a crowd: contention for the same features
    k["y"][:120] = 220 + rng.normal(0, 12, 120)
    k["angle"] = (rng.random(nt) * 360).astype(np.float32)
    k["octave"] = rng.integers(0, 4, nt)
    desc = synth.descriptors(nt, 5)
    desc[rng.integers(0, nt, 60)] = desc[0]
    un = ol.frame_undistort(CAM, k)
    b = ol.frame_bounds(CAM, capi.Bounds)
    off, feat = ol.frame_grid(b, un)
    grid = _grid_py(b, un)
    for c in range(64 * 48):                                # the CSR grid is the plain-Python mGrid
        assert list(feat[off[c]:off[c + 1]]) == grid[c // 48][c % 48]
    src = rng.integers(0, nt, nq)
    qxyr = np.stack([un["x"][src] + rng.normal(0, 4, nq), un["y"][src] + rng.normal(0, 4, nq), rng.choice([6.0, 15.0, 40.0], nq)], -1).astype(np.float32)
    lv = un["octave"][src]
    qlev = np.stack([lv - 1, lv + (rule != 0)], -1).astype(np.int32)
    qlev[::7] = -1
    qdesc = desc[src].copy()
    qdesc[np.arange(nq), rng.integers(0, 32, nq)] ^= np.uint8(4)
    qdesc[::5] = synth.descriptors(nq, 9)[::5]
    qangle = ((un["angle"][src] + rng.normal(15, 10, nq)) % 360).astype(np.float32)
    qvalid = (rng.random(nq) < 0.9).astype(np.uint8)
    claimed = (rng.random(nt) < 0.15).astype(np.uint8)
    for q in range(0, nq, 11):
        got = ol.frame_features_in_area(b, un, off, feat, float(qxyr[q, 0]), float(qxyr[q, 1]), float(qxyr[q, 2]), int(qlev[q, 0]), int(qlev[q, 1]))
        assert list(got) == _area_py(b, un, grid, qxyr[q, 0], qxyr[q, 1], qxyr[q, 2], int(qlev[q, 0]), int(qlev[q, 1]))
    want = _search_py(b, rule, th, ratio, check, un, desc, grid, claimed if rule == 0 else None, qxyr, qlev, qdesc, qangle, qvalid)
    got = ol.window_search(b, rule, th, ratio, check, un, desc, off, feat, claimed if rule == 0 else None, qxyr, qlev, qdesc, qangle, qvalid)
    assert got[0] == want[0] and list(got[1]) == want[1] and list(got[2]) == want[2] and list(got[3]) == want[3] and list(got[4]) == want[4]
    assert want[0] > 30


def test_distinctive_matches_plain_python():
    rng = np.random.default_rng(8)
    for n in (1, 2, 3, 4, 7, 20, 33):
        d = synth.descriptors(n, 30 + n)
        if n > 3:
            d[2] = d[0]
        D = [[_ham(d[i], d[j]) for j in range(n)] for i in range(n)]
        best, bi = 2**31 - 1, 0
        for i in range(n):
            med = sorted(D[i])[int(0.5 * (n - 1))]
            if med < best:
                best, bi = med, i
        assert ol.distinctive(d) == (bi, best)
