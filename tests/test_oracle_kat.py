"""Known-answer tests that pin the CPU oracle's primitives (SURVEY.md §4 tier T1).  The reference ships no
tests or golden vectors for this path (parity unpinned), so every primitive is checked against an
independent numpy/pure-Python statement of its published definition."""
import ctypes

import numpy as np
import pytest

import oracle_lib as orc
from orb_slam_amd import synth

L = orc.lib()


def test_cvround_ties_to_even_and_floor_ceil():
    for v, r in [(0.5, 0), (1.5, 2), (2.5, 2), (-0.5, 0), (-1.5, -2), (2.4999, 2), (2.5001, 3), (1e6 + 0.5, 1000000), (-2.5, -2)]:
        assert L.orc_cvRound(v) == r, v
    for v in [-2.5, -2.0, -0.1, 0.0, 0.1, 2.0, 2.5, 1e6 + 0.25]:
        assert L.orc_cvFloor(v) == int(np.floor(v)) and L.orc_cvCeil(v) == int(np.ceil(v))


def test_constructor_tables_match_survey():
    o = orc.OracleExtractor(1000, 1.2, 8)
    assert o.features_per_level() == [217, 181, 151, 126, 105, 87, 73, 60]
    assert o.umax() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert sum(2 * u + 1 for u in o.umax()[1:]) * 2 + 31 == 749      # pixels of the circular patch
    assert orc.OracleExtractor(2000, 1.2, 8).features_per_level() == [434, 362, 302, 251, 209, 175, 145, 122]
    sf = o.scale_factors()
    ref = [np.float32(1.0)]
    for _ in range(7):
        ref.append(np.float32(np.float64(ref[-1]) * np.float64(np.float32(1.2))))
    assert np.array_equal(sf.view(np.uint32), np.array(ref, np.float32).view(np.uint32))


def test_level_sizes_match_survey():
    o = orc.OracleExtractor(dumps=True)
    o(synth.frame(640, 480, synth.FLAT, 0))
    assert [o.level_size(l) for l in range(8)] == [(640, 480), (533, 400), (444, 333), (370, 278), (309, 231), (257, 193), (214, 161), (179, 134)]
    o = orc.OracleExtractor(nfeatures=2000, dumps=True)
    o(synth.frame(1920, 1080, synth.FLAT, 0))
    assert [o.level_size(l) for l in range(8)] == [(1920, 1080), (1600, 900), (1333, 750), (1111, 625), (926, 521), (772, 434), (643, 362), (536, 301)]


def test_gaussian_kernel_fixed_point():
    k = (ctypes.c_int * 7)()
    L.orc_gaussian_kernel_q8(k)
    assert list(k) == [18, 34, 49, 55, 49, 34, 18] and sum(k) == 257


def test_reflect101():
    for p, n, r in [(-1, 10, 1), (-3, 10, 3), (10, 10, 8), (12, 10, 6), (0, 10, 0), (9, 10, 9), (-5, 3, 1), (7, 3, 1), (-2, 1, 0), (-16, 5, 0)]:
        assert L.orc_reflect101(p, n) == r, (p, n)


def _atan2_np(y, x):
    f = np.float32
    k = f(180.0 / np.pi)
    p1, p3, p5, p7 = f(0.9997878412794807) * k, f(-0.3258083974640975) * k, f(0.1555786518463281) * k, f(-0.04432655554792128) * k
    ax, ay = np.abs(f(x)), np.abs(f(y))
    eps = f(2.2204460492503131e-16)
    if ax >= ay:
        c = ay / (ax + eps); c2 = c * c
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    else:
        c = ax / (ay + eps); c2 = c * c
        a = f(90.0) - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c
    if x < 0: a = f(180.0) - a
    if y < 0: a = f(360.0) - a
    return f(a)


def test_fast_atan2_definition_and_accuracy():
    rng = np.random.default_rng(0)
    pts = [(0, 0), (0, 1), (1, 0), (1, 1), (-1, 1), (-1, -1), (1, -1), (0, -5), (-7, 0), (2864925, -2864925)]
    pts += [tuple(v) for v in rng.integers(-2900000, 2900000, size=(3000, 2))]
    for y, x in pts:
        got = np.float32(L.orc_fastAtan2(float(y), float(x)))
        assert got.view(np.uint32) == _atan2_np(np.float32(y), np.float32(x)).view(np.uint32), (y, x)
        if x or y:
            true = np.degrees(np.arctan2(float(y), float(x))) % 360.0
            err = abs(float(got) - true)
            assert min(err, 360 - err) < 0.3                      # OpenCV documents ~0.3 degree accuracy
    assert L.orc_fastAtan2(0.0, 0.0) == 0.0


def test_hamming_bithack_equals_popcount():
    rng = np.random.default_rng(1)
    d = rng.integers(0, 256, size=(400, 32), dtype=np.uint8)
    d[0] = 0; d[1] = 255
    for i in range(0, 399):
        ref = int(np.unpackbits(d[i] ^ d[i + 1]).sum())
        assert orc.hamming256(d[i], d[i + 1]) == ref
    assert orc.hamming256(d[0], d[1]) == 256 and orc.hamming256(d[5], d[5]) == 0


def test_match_top2_scan_semantics():
    rng = np.random.default_rng(2)
    T = rng.integers(0, 4, size=(300, 32), dtype=np.uint8)     # low entropy -> many distance ties
    Q = rng.integers(0, 4, size=(60, 32), dtype=np.uint8)
    T[17] = T[3]; T[200] = T[3]; Q[0] = T[3]
    idx, best, sec = orc.match_top2(Q, T)
    D = np.unpackbits(Q[:, None, :] ^ T[None, :, :], axis=2).sum(2)
    for q in range(len(Q)):
        order = np.sort(D[q])
        assert best[q] == order[0] and sec[q] == order[1]           # two smallest WITH multiplicity
        assert idx[q] == int(np.argmax(D[q] == order[0]))           # first index attaining the best
    assert idx[0] == 3 and best[0] == 0 and sec[0] == 0
    i0, b0, s0 = orc.match_top2(Q[:2], T[:0])
    assert (i0 == -1).all() and (b0 == 2**31 - 1).all() and (s0 == 2**31 - 1).all()


def _resize_np(src, dw, dh):
    """cv::resize INTER_LINEAR 8U (OpenCV 2.4 fixed-point path) restated with numpy."""
    sh, sw = src.shape
    f32, f64 = np.float32, np.float64

    def taps(dn, sn, clamp_weights):
        scale = f64(1.0) / (f64(dn) / f64(sn))
        d = np.arange(dn, dtype=f64)
        fx = ((d + 0.5) * scale - 0.5).astype(f32)
        s = np.floor(fx).astype(np.int64)
        fx = (fx - s.astype(f32)).astype(f32)
        if clamp_weights:
            lo = s < 0; fx[lo] = 0; s[lo] = 0
            hi = s >= sn - 1; fx[hi] = 0; s[hi] = sn - 1
        a0 = np.rint((f32(1.0) - fx) * f32(2048)).astype(np.int64)
        a1 = np.rint(fx * f32(2048)).astype(np.int64)
        return s, a0, a1

    sx, a0, a1 = taps(dw, sw, True)
    sy, b0, b1 = taps(dh, sh, False)
    sx1 = np.minimum(sx + 1, sw - 1)
    S = src.astype(np.int64)
    y0 = np.clip(sy, 0, sh - 1); y1 = np.clip(sy + 1, 0, sh - 1)
    D0 = S[y0][:, sx] * a0 + S[y0][:, sx1] * a1
    D1 = S[y1][:, sx] * a0 + S[y1][:, sx1] * a1
    out = (((b0[:, None] * (D0 >> 4)) >> 16) + ((b1[:, None] * (D1 >> 4)) >> 16) + 2) >> 2
    return out.astype(np.uint8)


@pytest.mark.parametrize("sw,sh,dw,dh", [(640, 480, 533, 400), (533, 400, 444, 333), (214, 161, 179, 134), (1920, 1080, 1600, 900), (37, 41, 31, 34), (50, 20, 49, 19)])
def test_resize_linear_definition(sw, sh, dw, dh):
    src = synth.frame(sw, sh, synth.NOISE, sw + dh)
    assert np.array_equal(orc.resize_linear(src, dw, dh), _resize_np(src, dw, dh))


def _fast_bruteforce(img, th):
    """FAST-9/16 + score + 3x3 strict NMS straight from the definition."""
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    h, w = img.shape
    I = img.astype(int)
    score = np.zeros((h, w), int)
    corner = np.zeros((h, w), bool)
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            d = [I[y, x] - I[y + dy, x + dx] for dx, dy in ring]
            best = -10**9
            for s in range(16):
                arc = [d[(s + k) % 16] for k in range(9)]
                best = max(best, min(arc), min(-a for a in arc))
            if best > th:                       # 9 contiguous all < v-th  or all > v+th
                corner[y, x] = True
                score[y, x] = best - 1
    out = []
    for y in range(3, h - 3):
        for x in range(3, w - 3):
            if corner[y, x]:
                nb = [score[y + j, x + i] for j in (-1, 0, 1) for i in (-1, 0, 1) if (i, j) != (0, 0)]
                if all(score[y, x] > v for v in nb):
                    out.append((x, y, score[y, x]))
    return out, np.where(corner, score, 0)


@pytest.mark.parametrize("family,th", [(synth.NOISE, 20), (synth.NOISE, 7), (synth.BLOCKS, 20), (synth.LOWTEX, 7), (synth.NOISE, 60)])
def test_fast_definition(family, th):
    img = synth.frame(64, 48, family, 9)[:40, :56]
    kps, sc = orc.fast(img, th, want_scores=True)
    ref, ref_sc = _fast_bruteforce(img, th)
    assert [(int(k["x"]), int(k["y"]), int(k["response"])) for k in kps] == ref       # raster order
    assert np.array_equal(sc.astype(int), ref_sc)
    assert all(k["size"] == 7 and k["angle"] == -1 and k["octave"] == 0 and k["class_id"] == -1 for k in kps)


def test_fast_corner_set_equals_scikit_image():
    """third-party voice (neither this repository nor OpenCV): scikit-image 0.18.3's corner_fast(n=9) marks exactly the pixels
    the oracle's cv::FAST restatement scores before non-maximum suppression (fixture: tests/golden/make_fast_skimage.py)"""
    import os
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fast_skimage.npz"))
    total = 0
    for name in ("blocks", "noise", "lowtex"):
        h, w, fam, idx = (int(v) for v in z[name + "_shape"])
        img = synth.frame(w, h, fam, idx)
        for t in (20, 7):
            want = np.unpackbits(z["%s_t%d" % (name, t)])[: h * w].reshape(h, w).astype(bool)
            _, sc = orc.fast(img, t, want_scores=True)
            assert np.array_equal(sc > 0, want), (name, t)
            total += int(want.sum())
    assert total > 10000


def test_fast_score_is_threshold_independent():
    """corner@t <=> score >= t, and survivors@20 = survivors@7 restricted to score >= 20 (one pass serves both)."""
    img = synth.frame(160, 120, synth.NOISE, 4)
    k7, k20 = orc.fast(img, 7), orc.fast(img, 20)
    sub = k7[k7["response"] >= 20]
    assert len(k20) > 0 and np.array_equal(sub, k20)


def _blur_np(img, mode):
    k = np.array([18, 34, 49, 55, 49, 34, 18], np.int64)
    h, w = img.shape
    P = np.pad(img.astype(np.int64), 3, mode="reflect")            # numpy 'reflect' == BORDER_REFLECT_101
    rows = sum(k[i] * P[:, i:i + w] for i in range(7))
    s = sum(k[j] * rows[j:j + h, :] for j in range(7))
    q = s >> 16
    rem = s & 0xFFFF
    up = (rem >= 0x8000).astype(np.int64)
    even = ((rem > 0x8000) | ((rem == 0x8000) & ((q & 1) == 1))).astype(np.int64)
    use_even = np.zeros((h, w), bool)
    if mode == 0:
        use_even[:, : (w & ~3)] = True
    return np.minimum(q + np.where(use_even, even, up), 255).astype(np.uint8)


@pytest.mark.parametrize("w,h,family", [(64, 48, 0), (61, 35, 0), (130, 70, 1), (33, 33, 3)])
@pytest.mark.parametrize("mode", [0, 1])
def test_gaussian_blur_definition(w, h, family, mode):
    img = synth.frame(w, h, family, 3)
    assert np.array_equal(orc.gaussian_blur7(img, mode), _blur_np(img, mode))


def test_gaussian_blur_tie_rule():
    """a pixel whose fixed-point sum is exactly k + 0.5: SSE2 emulation rounds to even, scalar tail rounds up"""
    s = 0
    found = None
    for v in range(256):                      # constant image v -> sum = v*257*257
        s = v * 257 * 257
        if (s & 0xFFFF) == 0x8000:
            found = v
    assert found is None                      # no constant image ties: construct one by search instead
    rng = np.random.default_rng(5)
    for _ in range(200):
        img = rng.integers(0, 256, size=(16, 16), dtype=np.uint8)
        a, b = orc.gaussian_blur7(img, 0), orc.gaussian_blur7(img, 1)
        assert (np.abs(a.astype(int) - b.astype(int)) <= 1).all() and (a <= b).all()


def test_retain_best_semantics():
    rng = np.random.default_rng(3)
    for n, keep in [(10, 3), (100, 17), (500, 217), (64, 64), (5, 9), (50, 0), (300, 1)]:
        r = rng.integers(7, 60, size=n).astype(np.float32)        # FAST scores: small ints, many ties
        idx = orc.retain_best(r, keep)
        assert len(idx) == min(n, keep) and len(set(idx.tolist())) == len(idx)
        if keep >= n:
            assert idx.tolist() == list(range(n))                  # untouched, original order
        else:
            assert sorted(r[idx].tolist(), reverse=True) == sorted(r.tolist(), reverse=True)[:keep]
    r = np.arange(100, dtype=np.float32)                           # no ties: the set is unique
    assert sorted(orc.retain_best(r, 10).tolist()) == list(range(90, 100))


def test_sincos_is_glibc():
    s, c = ctypes.c_float(), ctypes.c_float()
    for a in [0.0, 0.5, 1.0, 3.1415927, 6.28]:
        L.orc_sincosf(a, ctypes.byref(s), ctypes.byref(c))
        assert abs(s.value - np.sin(np.float32(a), dtype=np.float64)) < 1e-6 and abs(c.value - np.cos(np.float32(a), dtype=np.float64)) < 1e-6
