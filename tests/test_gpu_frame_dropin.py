"""The drop-in claim, executed on the GPU (missing #4): the reference's OWN Frame::Frame (src/Frame.cc:56-128, compiled
where it lies: oracle/_ref/libref_frame_product.so) with the PRODUCT's orb_slam_amd/cpp/ORBextractor.h in place of the reference's
header, linked against orb_slam_amd/liborbx.so — its call `(*mpORBextractor)(im, cv::Mat(), mvKeys, mDescriptors)` at :60 lands in
orbx_extract.  Compared with the same constructor driven by the reference's own extractor:
  mvKeys, mDescriptors        == /root/reference/src/ORBextractor.cc on the same image (oracle/_ref/libref_orbextractor.so)
  mvKeysUn, bounds, mGrid     == the reference's Frame.cc fed with that extractor's key points (oracle/_ref/libref_frame.so)
  mnScaleLevels, mfScaleFactor = GetLevels() / GetScaleFactor() of the product class."""
import ctypes
import os

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import capi, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
PRODUCT = os.path.join(REFDIR, "libref_frame_product.so")
FRAME = os.path.join(REFDIR, "libref_frame.so")
pytestmark = [pytest.mark.gpu,
              pytest.mark.skipif(not (os.path.exists(PRODUCT) and os.path.exists(FRAME) and ol.ref_available()),
                                 reason="oracle/_ref is built only where /root/reference exists")]

CAMERAS = [
    (517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), 640, 480),       # TUM fr1 (Data/Settings.yaml of the reference)
    (535.4, 539.2, 320.1, 247.6, (0.0, 0.0, 0.0, 0.0), 640, 480),                      # no distortion: the other branch of UndistortKeyPoints
    (1400.0, 1400.0, 960.0, 540.0, (-0.1, 0.02, 0.001, -0.002), 1920, 1080),
]
_L = {}


def _libs():
    if not _L:
        ol.lib()
        vp, i = ctypes.c_void_p, ctypes.c_int
        P = ctypes.CDLL(PRODUCT)
        P.ref_frame_product_build.argtypes = [vp, vp, i, i, i, vp, vp, vp, vp, vp, vp, vp, vp]
        F = ctypes.CDLL(FRAME)
        F.ref_frame_build.argtypes = [vp, vp, i, vp, vp, vp, vp]
        _L["P"], _L["F"] = P, F
    return _L["P"], _L["F"]


@pytest.mark.parametrize("ci,nf,family", [(0, 1000, synth.BLOCKS), (0, 2000, synth.BLOCKS), (1, 1000, synth.NOISE), (1, 1000, synth.LOWTEX),
                                          (2, 2000, synth.BLOCKS), (0, 1000, synth.FLAT)],
                         ids=["tum_fr1_1000", "tum_fr1_init2000", "nodist_noise", "nodist_lowtex", "hd1080_2000", "flat_no_keypoints"])
def test_reference_frame_constructor_runs_on_the_product_extractor(ci, nf, family):
    P, F = _libs()
    fx, fy, cx, cy, dist, w, h = CAMERAS[ci]
    cam = capi.Camera.make(fx, fy, cx, cy, dist, w, h)
    cap = nf + 64
    for idx in (11, 12):                                     # two frames through one extractor instance (handle reuse, as Tracking does)
        img = synth.frame(w, h, family, idx)
        keys = np.zeros(cap, dtype=capi.KP_DTYPE); keys_un = np.zeros(cap, dtype=capi.KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        b = capi.Bounds()
        off = np.zeros(64 * 48 + 1, np.int32); feat = np.zeros(cap + 1, np.int32)
        levels, scale = ctypes.c_int32(0), ctypes.c_float(0)
        n = P.ref_frame_product_build(ctypes.addressof(cam), img.ctypes.data, img.strides[0], nf, cap, keys.ctypes.data, desc.ctypes.data, keys_un.ctypes.data,
                                      ctypes.addressof(b), off.ctypes.data, feat.ctypes.data, ctypes.addressof(levels), ctypes.addressof(scale))
        assert n >= 0, "ref_frame_product_build failed (%d)" % n
        # the reference's own extractor on the same image
        rk, rd = ol.RefExtractor(nf)(img)
        assert n == len(rk)
        if family == synth.FLAT:
            assert n == 0                                    # Frame::Frame returns right after the extractor (:64-65): the scale members stay unset
            continue
        assert (levels.value, scale.value) == (8, np.float32(1.2))
        assert n > 0.8 * nf or family == synth.LOWTEX
        assert keys[:n].tobytes() == rk.tobytes()            # mvKeys: order, coordinates, size, angle bits, response, octave
        assert desc[:n].tobytes() == rd.tobytes()            # mDescriptors
        # the reference's Frame.cc fed with the reference extractor's key points
        b_ref = capi.Bounds()
        un_ref = np.zeros(n, dtype=capi.KP_DTYPE)
        off_ref = np.zeros(64 * 48 + 1, np.int32); feat_ref = np.zeros(n + 1, np.int32)
        assert F.ref_frame_build(ctypes.addressof(cam), rk.ctypes.data, n, ctypes.addressof(b_ref), un_ref.ctypes.data, off_ref.ctypes.data, feat_ref.ctypes.data) == n
        assert b.astuple() == b_ref.astuple()                # ComputeImageBounds + the two inverse cell sizes
        assert keys_un[:n].tobytes() == un_ref.tobytes()     # mvKeysUn
        np.testing.assert_array_equal(off, off_ref)          # mGrid, cell by cell in push_back order
        np.testing.assert_array_equal(feat[:off[-1]], feat_ref[:off_ref[-1]])
    P.ref_frame_product_close()
