"""CPU checks of the PRODUCT's scalar arithmetic (orb_slam_amd/csrc/orb_math.h, compiled for the host by
tests/_probe/math_probe.cpp) against the oracle and against glibc — no GPU needed.  The same header is
what the HIP kernels compile, so kernel-side arithmetic is verified here and only data movement is left
to the -m gpu tests."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

import oracle_lib as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def probe():
    so = os.path.join(ROOT, "tests", "_probe", "libmath_probe.so")
    src = os.path.join(ROOT, "tests", "_probe", "math_probe.cpp")
    hdr = os.path.join(ROOT, "orb_slam_amd", "csrc", "orb_math.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
        subprocess.check_call(["g++", "-O2", "-ffp-contract=off", "-std=c++17", "-fPIC", "-shared",
                               "-I" + os.path.join(ROOT, "orb_slam_amd", "csrc"), src, "-o", so])
    P = ctypes.CDLL(so)
    P.probe_fast_atan2.restype = ctypes.c_float
    P.probe_fast_atan2.argtypes = [ctypes.c_float, ctypes.c_float]
    P.probe_cv_round_f.argtypes = [ctypes.c_float]
    P.probe_sincos_sweep.restype = ctypes.c_long
    P.probe_sincos_sweep.argtypes = [ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_int]
    P.probe_fast9_score.argtypes = [ctypes.c_void_p, ctypes.c_int]
    return P


def test_sincos_equals_glibc_exhaustively(probe):
    """every float in [0, 6.2832] (the whole range angle*pi/180 can take): 1.09e9 inputs, bit-for-bit"""
    hi = struct.unpack("<I", struct.pack("<f", 6.2832))[0]
    bad = (ctypes.c_uint32 * 16)()
    n = probe.probe_sincos_sweep(0, hi, 1, bad, 16)
    assert n == 0, "sincos differs from glibc at %d inputs, e.g. %s" % (n, [hex(b) for b in bad[:min(n, 16)]])


def test_atan2_equals_oracle(probe):
    L = orc.lib()
    rng = np.random.default_rng(0)
    v = rng.integers(-2900000, 2900000, size=(200000, 2))
    v[:100, 0] = 0; v[50:150, 1] = 0
    for y, x in v[:40000]:
        a = np.float32(probe.probe_fast_atan2(float(y), float(x)))
        b = np.float32(L.orc_fastAtan2(float(y), float(x)))
        assert a.view(np.uint32) == b.view(np.uint32), (y, x)


def test_cv_round_f(probe):
    L = orc.lib()
    for v in [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 17.49999, 17.5, 18.5, -13.5, 0.0, 1e-8]:
        assert probe.probe_cv_round_f(v) == L.orc_cvRound(float(np.float32(v)))


def test_fast9_score_equals_opencv_ladder(probe):
    """the product's 2x min3/max3 window-9 network vs the oracle's restatement of OpenCV's cornerScore ladder,
    on random 7x7 patches (centre pixel only)"""
    rng = np.random.default_rng(4)
    ring = [(0, 3), (1, 3), (2, 2), (3, 1), (3, 0), (3, -1), (2, -2), (1, -3), (0, -3), (-1, -3), (-2, -2), (-3, -1), (-3, 0), (-3, 1), (-2, 2), (-1, 3)]
    n_corner = 0
    for it in range(6000):
        amp = [255, 60, 24][it % 3]
        img = (128 + rng.integers(-amp // 2, amp // 2 + 1, size=(7, 7))).clip(0, 255).astype(np.uint8)
        if it % 5 == 0:                      # plant a bright/dark arc so that real corners occur
            s, ln = rng.integers(0, 16), rng.integers(8, 13)
            for k in range(ln):
                dx, dy = ring[(s + k) % 16]
                img[3 + dy, 3 + dx] = 250 if it % 2 else 5
        d = (ctypes.c_int * 16)(*[int(img[3, 3]) - int(img[3 + dy, 3 + dx]) for dx, dy in ring])
        for tmin in (7, 20, 3):
            got = probe.probe_fast9_score(d, tmin)
            _, sc = orc.fast(img, tmin, want_scores=True)
            assert got == int(sc[3, 3]), (it, tmin)
            n_corner += got > 0
    assert n_corner > 300


def test_blur_round_and_taps(probe):
    for s in [0, 1, 0x7FFF, 0x8000, 0x8001, 0x18000, 0x28000, 0x17FFF, 0x18001, 255 * 257 * 257, 254 * 65536 + 0x8000, 255 * 65536 + 0x8000]:
        q, rem = s >> 16, s & 0xFFFF
        up = min(q + (rem >= 0x8000), 255)
        ev = min(q + ((rem > 0x8000) or (rem == 0x8000 and q & 1)), 255)
        assert probe.probe_blur_round(s, 0) == up and probe.probe_blur_round(s, 1) == ev
    v = (ctypes.c_int * 7)(1, 2, 3, 4, 5, 6, 7)
    assert probe.probe_blur_taps7(v) == 18 * 8 + 34 * 8 + 49 * 8 + 55 * 4


def test_resize_px_and_reflect(probe):
    assert probe.probe_resize_px(10, 20, 30, 40, 2048, 0, 2048, 0) == 10
    assert probe.probe_resize_px(10, 20, 30, 40, 0, 2048, 0, 2048) == 40
    assert probe.probe_resize_px(255, 255, 255, 255, 1000, 1048, 300, 1748) == 255
    L = orc.lib()
    for p in range(-20, 40):
        for n in (1, 2, 5, 17):
            assert probe.probe_reflect101(p, n) == L.orc_reflect101(p, n)
