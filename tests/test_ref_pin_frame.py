"""Pins the oracle's Frame-side restatements (oracle/frame_oracle.cpp: image bounds, inverse cell sizes, the mGrid fill with
PosInGrid, GetFeaturesInArea, the scale tables) to the reference's OWN src/Frame.cc + include/Frame.h, compiled where they lie
with a stand-in extractor that hands the constructor preset key points (oracle/ref_frame_wrap.cpp -> oracle/_ref/libref_frame.so).
cv::undistortPoints is an OpenCV primitive: behind the stand-in header it IS the oracle's restatement, so that part stays
unpinned; everything Frame.cc does around it runs from the reference's source text.  No GPU."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(ROOT, "oracle", "_ref", "libref_frame.so")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="oracle/_ref/libref_frame.so is built only where /root/reference exists")
_REF = None


def ref():
    global _REF
    if _REF is None:
        L = ctypes.CDLL(PATH)
        f, i, vp = ctypes.c_float, ctypes.c_int, ctypes.c_void_p
        L.ref_frame_build.argtypes = [vp, vp, i, vp, vp, vp, vp]
        L.ref_frame_features_in_area.argtypes = [f, f, f, i, i, vp]
        L.ref_frame_scale_tables.argtypes = [vp, vp, vp]
        _REF = L
    return _REF


CAMERAS = [
    (517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), 640, 480),       # TUM fr1 (distorted)
    (535.4, 539.2, 320.1, 247.6, (0.0, 0.0, 0.0, 0.0), 640, 480),                      # TUM fr3 (no distortion: the other branch)
    (458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05), 752, 480),
    (1400.0, 1400.0, 960.0, 540.0, (-0.1, 0.02, 0.001, -0.002), 1920, 1080),
]


def _keypoints(rng, n, w, h):
    k = np.zeros(n, dtype=capi.KP_DTYPE)
    k["x"] = (rng.random(n) * (w + 40) - 20).astype(np.float32)          # some outside the image: PosInGrid must drop them
    k["y"] = (rng.random(n) * (h + 40) - 20).astype(np.float32)
    if n > 50:
        k["x"][:n // 4] = (w / 2 + rng.normal(0, 12, n // 4)).astype(np.float32)      # a crowded spot
        k["y"][:n // 4] = (h / 2 + rng.normal(0, 12, n // 4)).astype(np.float32)
    k["angle"] = (rng.random(n) * 360).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    k["size"], k["class_id"] = 31, -1
    return k


@pytest.mark.parametrize("ci", range(len(CAMERAS)))
@pytest.mark.parametrize("n", [1, 300, 2000])
def test_frame_constructor_grid_and_window_queries(ci, n):
    fx, fy, cx, cy, dist, w, h = CAMERAS[ci]
    cam = capi.Camera.make(fx, fy, cx, cy, dist, w, h)
    rng = np.random.default_rng(100 * ci + n)
    kps = _keypoints(rng, n, w, h)
    b_ref = capi.Bounds()
    un_ref = np.zeros(n, dtype=capi.KP_DTYPE)
    off_ref = np.zeros(64 * 48 + 1, np.int32); feat_ref = np.zeros(n + 1, np.int32)
    got_n = ref().ref_frame_build(ctypes.addressof(cam), kps.ctypes.data, n, ctypes.addressof(b_ref), un_ref.ctypes.data, off_ref.ctypes.data, feat_ref.ctypes.data)
    assert got_n == n
    # bounds + inverse cell sizes, undistorted key points (bit for bit), grid
    b = ol.frame_bounds(cam, capi.Bounds)
    assert b.astuple() == b_ref.astuple()
    un = ol.frame_undistort(cam, kps)
    assert un.tobytes() == un_ref.tobytes()
    off, feat = ol.frame_grid(b, un)
    np.testing.assert_array_equal(off, off_ref)
    np.testing.assert_array_equal(feat, feat_ref[:off_ref[-1]])
    assert off[-1] <= n and (n < 300 or off[-1] < n)                     # some key points fell outside the grid
    # window queries, all the level-argument forms the callers use
    out = np.zeros(n + 1, np.int32)
    hits = 0
    for _ in range(120):
        j = int(rng.integers(0, n))
        x, y = float(un["x"][j] + rng.normal(0, 5)), float(un["y"][j] + rng.normal(0, 5))
        if rng.random() < 0.1:
            x, y = float(rng.choice([-50.0, w + 60.0])), float(rng.random() * h)          # a window off the image
        r = float(rng.choice([1.0, 4.0, 15.0, 60.0, 250.0]))
        lv = int(un["octave"][j])
        mn, mx = [(-1, -1), (lv, lv), (lv - 1, lv), (lv - 1, lv + 1), (0, 0), (-1, lv), (lv, -1), (3, 1)][int(rng.integers(0, 8))]
        m = ref().ref_frame_features_in_area(x, y, r, mn, mx, out.ctypes.data)
        want = ol.frame_features_in_area(b, un, off, feat, x, y, r, mn, mx)
        np.testing.assert_array_equal(out[:m], want, err_msg="query (%g, %g, r=%g, levels %d..%d)" % (x, y, r, mn, mx))
        hits += m
    assert hits > 0 or n < 300


def test_scale_tables_of_the_frame():
    """mvScaleFactors / mvLevelSigma2 / mvInvLevelSigma2 as Frame::Frame derives them (float chain) = what the tests and tools use"""
    import kf_pairs
    cam = capi.Camera.make(*CAMERAS[1])
    kps = _keypoints(np.random.default_rng(1), 10, 640, 480)
    tmp = capi.Bounds(); un = np.zeros(10, dtype=capi.KP_DTYPE); off = np.zeros(3073, np.int32); feat = np.zeros(11, np.int32)
    ref().ref_frame_build(ctypes.addressof(cam), kps.ctypes.data, 10, ctypes.addressof(tmp), un.ctypes.data, off.ctypes.data, feat.ctypes.data)
    f = np.zeros(8, np.float32); s2 = np.zeros(8, np.float32); inv = np.zeros(8, np.float32)
    assert ref().ref_frame_scale_tables(f.ctypes.data, s2.ctypes.data, inv.ctypes.data) == 8
    np.testing.assert_array_equal(s2, kf_pairs.LEVEL_SIGMA2)
    chain = [np.float32(1.0)]
    for _ in range(7):
        chain.append(np.float32(chain[-1] * np.float32(1.2)))
    np.testing.assert_array_equal(f, np.array(chain, np.float32))
    np.testing.assert_array_equal(inv, (np.float32(1.0) / s2).astype(np.float32))
