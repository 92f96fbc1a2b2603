"""Seeded key-frame pairs for the KeyFrame-to-KeyFrame searches (SearchForTriangulation, SearchByBoW(KF, KF), SearchBySim3):
two views of the same points under a real fundamental matrix, descriptors of view 2 = noisy copies of view 1's (several per
feature, so later queries meet claimed candidates), some features already holding map points.  Test infrastructure."""
import numpy as np

from orb_slam_amd import capi, synth

# mvLevelSigma2 the way ORBextractor builds it: mvScaleFactor[i] = mvScaleFactor[i-1]*scaleFactor, sigma2 = factor*factor (floats)
_sf = [np.float32(1.0)]
for _ in range(7):
    _sf.append(np.float32(_sf[-1] * np.float32(1.2)))
LEVEL_SIGMA2 = np.array([s * s for s in _sf], np.float32)


def fundamental(seed):
    """F12 with x1' F12 x2 = 0 for a small rotation + translation between two 640x480 views (float32 3x3)"""
    rng = np.random.default_rng(seed)
    K = np.array([[517.3, 0, 318.6], [0, 516.5, 255.3], [0, 0, 1]])
    w = rng.normal(0, 0.03, 3)
    th = np.linalg.norm(w) + 1e-12
    k = w / th
    Kx = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    R = np.eye(3) + np.sin(th) * Kx + (1 - np.cos(th)) * Kx @ Kx
    t = rng.normal(0, 1, 3); t /= np.linalg.norm(t)
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    Ki = np.linalg.inv(K)
    F = Ki.T @ tx @ R @ Ki
    F /= np.abs(F).max()
    return F.astype(np.float32)


def pair(seed, n1, n2, max_flips=24, line_noise=2.0, p_mp1=0.3, p_mp2=0.3):
    rng = np.random.default_rng(seed)
    F = fundamental(seed + 17)
    k1 = np.zeros(n1, dtype=capi.KP_DTYPE); k2 = np.zeros(n2, dtype=capi.KP_DTYPE)
    k1["x"] = (rng.random(n1) * 640).astype(np.float32); k1["y"] = (rng.random(n1) * 480).astype(np.float32)
    k1["angle"] = (rng.random(n1) * 360).astype(np.float32); k1["octave"] = rng.integers(0, 8, n1)
    k1["size"], k1["class_id"] = 31, -1
    d1 = synth.descriptors(max(n1, 1), seed + 1)[:n1]
    d2 = synth.descriptors(max(n2, 1), seed + 2)[:n2]
    k2["x"] = (rng.random(n2) * 640).astype(np.float32); k2["y"] = (rng.random(n2) * 480).astype(np.float32)
    k2["angle"] = (rng.random(n2) * 360).astype(np.float32); k2["octave"] = rng.integers(0, 8, n2)
    k2["size"], k2["class_id"] = 31, -1
    if n1 and n2:
        src = rng.integers(0, n1, n2)
        copy = rng.random(n2) < 0.85
        nd = d1[src].copy()
        nfl = rng.integers(0, max_flips + 1, n2)
        for j in range(max_flips):
            bit = rng.integers(0, 256, n2)
            m = j < nfl
            nd[np.arange(n2)[m], bit[m] // 8] ^= (1 << (bit[m] % 8)).astype(np.uint8)
        d2[copy] = nd[copy]
        # view-2 position: a point of the epipolar line of the source feature + noise across the line (in units of the level sigma)
        Fd = F.astype(np.float64)
        x1 = np.stack([k1["x"][src], k1["y"][src], np.ones(n2)], 1).astype(np.float64)
        l = x1 @ Fd                                  # rows [a b c]
        nrm = np.hypot(l[:, 0], l[:, 1]) + 1e-30
        px = rng.random(n2) * 640
        py = -(l[:, 0] * px + l[:, 2]) / np.where(np.abs(l[:, 1]) < 1e-12, 1e-12, l[:, 1])
        off = rng.normal(0, line_noise, n2) * np.sqrt(LEVEL_SIGMA2[k2["octave"]])
        px = px + off * l[:, 0] / nrm; py = py + off * l[:, 1] / nrm
        k2["x"][copy] = px.astype(np.float32)[copy]; k2["y"][copy] = py.astype(np.float32)[copy]
        near = copy & (rng.random(n2) < 0.8)
        k2["angle"][near] = ((k1["angle"][src] - rng.normal(15, 8, n2).astype(np.float32)) % np.float32(360))[near]
    mp1 = (rng.random(n1) < p_mp1).astype(np.uint8)
    mp2 = (rng.random(n2) < p_mp2).astype(np.uint8)
    return dict(F=F, k1=k1, d1=d1, mp1=mp1, k2=k2, d2=d2, mp2=mp2)


# ---- plain-Python restatements (known-answer side of the CPU tests; written from the reference text, independent of oracle/)
def _popcount_dist(a, b):
    return int(np.unpackbits(np.bitwise_xor(a, b)).sum())


def _three_maxima(h):
    max1 = max2 = max3 = 0
    i1 = i2 = i3 = -1
    for i, b in enumerate(h):
        s = len(b)
        if s > max1:
            max3, max2, max1 = max2, max1, s; i3, i2, i1 = i2, i1, i
        elif s > max2:
            max3, max2 = max2, s; i3, i2 = i2, i
        elif s > max3:
            max3 = s; i3 = i
    if max2 < np.float32(0.1) * np.float32(max1):
        i2 = i3 = -1
    elif max3 < np.float32(0.1) * np.float32(max1):
        i3 = -1
    return i1, i2, i3


def _rot_bin(a1, a2):
    rot = np.float32(a1) - np.float32(a2)
    if rot < 0:
        rot = np.float32(rot + np.float32(360))
    v = np.float32(rot * np.float32(1.0 / 30))
    b = int(np.floor(abs(float(v)) + 0.5) * (1 if v >= 0 else -1))      # C round(): half away from zero
    return 0 if b == 30 else b


def py_epipolar(kp1, kp2, F, sigma2):
    f = np.float32
    F = np.asarray(F, np.float32).reshape(3, 3)
    a = f(f(f(kp1["x"]) * F[0, 0] + f(kp1["y"]) * F[1, 0]) + F[2, 0])
    b = f(f(f(kp1["x"]) * F[0, 1] + f(kp1["y"]) * F[1, 1]) + F[2, 1])
    c = f(f(f(kp1["x"]) * F[0, 2] + f(kp1["y"]) * F[1, 2]) + F[2, 2])
    num = f(f(a * f(kp2["x"]) + b * f(kp2["y"])) + c)
    den = f(a * a + b * b)
    if den == 0:
        return False
    dsqr = f(f(num * num) / den)
    return float(dsqr) < 3.84 * float(sigma2)


def _walk(fv1, fv2):
    """the merge walk over two FeatureVectors: yields (run of fv1 features, run of fv2 features) for the common nodes"""
    (n1, o1, f1), (n2, o2, f2) = fv1, fv2
    a = b = 0
    while a < len(n1) and b < len(n2):
        if n1[a] == n2[b]:
            yield f1[o1[a]:o1[a + 1]], f2[o2[b]:o2[b + 1]]
            a += 1; b += 1
        elif n1[a] < n2[b]:
            a += 1
        else:
            b += 1


def py_search_for_triangulation(th_low, check, F, sigma2, fv1, k1, d1, mp1, fv2, k2, d2, mp2):
    n1, n2 = len(d1), len(d2)
    m12 = np.full(n1, -1, np.int32)
    matched2 = np.zeros(n2, bool)
    hist = [[] for _ in range(30)]
    nm = 0
    for run1, run2 in _walk(fv1, fv2):
        for idx1 in run1:
            if mp1[idx1]:
                continue
            cand = []
            for idx2 in run2:
                if matched2[idx2] or mp2[idx2]:
                    continue
                dist = _popcount_dist(d1[idx1], d2[idx2])
                if dist > th_low:
                    continue
                cand.append((dist, int(idx2)))
            if not cand:
                continue
            cand.sort()
            dist_th = 2 * cand[0][0]
            for dist, idx2 in cand:
                if dist > dist_th:
                    break
                if py_epipolar(k1[idx1], k2[idx2], F, sigma2[k2["octave"][idx2]]):
                    matched2[idx2] = True
                    m12[idx1] = idx2
                    nm += 1
                    if check:
                        hist[_rot_bin(k1["angle"][idx1], k2["angle"][idx2])].append(int(idx1))
                    break
    if check:
        keep = _three_maxima(hist)
        for i in range(30):
            if i in keep:
                continue
            for idx1 in hist[i]:
                m12[idx1] = -1
                nm -= 1
    return nm, m12


def py_search_by_bow_kf(th_low, ratio, check, fv1, d1, a1, v1, fv2, d2, a2, v2):
    n1, n2 = len(d1), len(d2)
    m12 = np.full(n1, -1, np.int32)
    matched2 = np.zeros(n2, bool)
    hist = [[] for _ in range(30)]
    nm = 0
    for run1, run2 in _walk(fv1, fv2):
        for idx1 in run1:
            if not v1[idx1]:
                continue
            best1 = best2 = 2 ** 31 - 1
            bi = -1
            for idx2 in run2:
                if matched2[idx2] or not v2[idx2]:
                    continue
                dist = _popcount_dist(d1[idx1], d2[idx2])
                if dist < best1:
                    best2, best1, bi = best1, dist, int(idx2)
                elif dist < best2:
                    best2 = dist
            if best1 < th_low and np.float32(best1) < np.float32(ratio) * np.float32(best2):
                m12[idx1] = bi
                matched2[bi] = True
                nm += 1
                if check:
                    hist[_rot_bin(a1[idx1], a2[bi])].append(int(idx1))
    if check:
        keep = _three_maxima(hist)
        for i in range(30):
            if i in keep:
                continue
            for idx1 in hist[i]:
                m12[idx1] = -1
                nm -= 1
    return nm, m12
