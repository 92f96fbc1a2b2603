"""GPU parity of the Frame-side steps (include/orbf.h): undistortion, the 64x48 search grid and the window query,
against oracle/frame_oracle.cpp.  Float bit patterns and integer lists exact."""
import ctypes

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

CAMS = {
    "tum1": capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), 640, 480),
    "euroc": capi.Camera.make(458.654, 457.296, 367.215, 248.375, (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05), 752, 480),
    "nodist": capi.Camera.make(535.4, 539.2, 320.1, 247.6, (0.0, 0.0, 0.0, 0.0), 640, 480),
    "k3": capi.Camera.make(1400.0, 1400.0, 960.0, 540.0, (-0.1, 0.02, 0.001, -0.002, 0.003), 1920, 1080),
}


def _random_kps(n, w, h, seed):
    rng = np.random.default_rng(seed)
    k = np.zeros(n, dtype=capi.KP_DTYPE)
    k["x"] = (rng.random(n) * (w + 40) - 20).astype(np.float32)        # some outside the image
    k["y"] = (rng.random(n) * (h + 40) - 20).astype(np.float32)
    k["size"] = 31
    k["angle"] = (rng.random(n) * 360).astype(np.float32)
    k["response"] = rng.integers(7, 200, n).astype(np.float32)
    k["octave"] = rng.integers(0, 8, n)
    k["class_id"] = -1
    return k


def _check_frame(cam, kps):
    b = capi.image_bounds(cam)
    un, off, feat = capi.undistort_grid(cam, b, kps)
    want_un = ol.frame_undistort(cam, kps)
    assert un.tobytes() == want_un.tobytes()
    woff, wfeat = ol.frame_grid(b, want_un)
    assert np.array_equal(off, woff) and np.array_equal(feat, wfeat)
    return b, un, off, feat


@pytest.mark.parametrize("name", sorted(CAMS))
def test_undistort_and_grid(name):
    cam = CAMS[name]
    for n, seed in ((1000, 1), (2000, 2), (1, 3), (0, 4), (8192, 5), (777, 6)):
        _check_frame(cam, _random_kps(n, cam.width, cam.height, seed))
    if cam.width == 640:
        ok, od = ol.OracleExtractor(1000)(synth.frame(640, 480, synth.BLOCKS, 9))     # real extractor output (level-scaled coordinates)
        _check_frame(cam, ok)


def test_clustered_points_share_cells():
    cam = CAMS["tum1"]
    k = _random_kps(3000, 640, 480, 8)
    k["x"][:2000] = 300 + (np.arange(2000) % 7)
    k["y"][:2000] = 200 + (np.arange(2000) % 5)
    _check_frame(cam, k)


def test_batch_device_layout():
    torch = pytest.importorskip("torch")
    cam = CAMS["euroc"]
    b = capi.image_bounds(cam)
    B, cap = 7, 1000
    n = np.array([1000, 0, 999, 1, 512, 1000, 37], np.int32)
    K = np.stack([_random_kps(cap, cam.width, cam.height, 20 + i) for i in range(B)])
    dK = torch.from_numpy(K.view(np.uint8).reshape(B, cap, 28)).cuda()
    dn = torch.from_numpy(n).cuda()
    dUn = torch.zeros((B, cap, 28), dtype=torch.uint8, device="cuda")
    dOff = torch.full((B, capi.GRID_CELLS + 1), -1, dtype=torch.int32, device="cuda")
    dFeat = torch.full((B, cap), -1, dtype=torch.int32, device="cuda")
    capi.undistort_grid_batch_device(cam, b, dK.data_ptr(), dn.data_ptr(), B, cap, dUn.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    un = dUn.cpu().numpy().reshape(B, cap * 28).view(capi.KP_DTYPE).reshape(B, cap)
    off, feat = dOff.cpu().numpy(), dFeat.cpu().numpy()
    for f in range(B):
        want_un = ol.frame_undistort(cam, K[f, :n[f]])
        assert un[f, :n[f]].tobytes() == want_un.tobytes()
        woff, wfeat = ol.frame_grid(b, want_un)
        assert np.array_equal(off[f], woff) and np.array_equal(feat[f, :woff[-1]], wfeat)


@pytest.mark.parametrize("name", ["tum1", "nodist"])
def test_features_in_area(name):
    cam = CAMS[name]
    kps = _random_kps(2000, cam.width, cam.height, 31)
    b, un, off, feat = _check_frame(cam, kps)
    rng = np.random.default_rng(5)
    nq = 1500
    q = np.zeros((nq, 3), np.float32)
    q[:, 0] = rng.random(nq) * (cam.width + 100) - 50
    q[:, 1] = rng.random(nq) * (cam.height + 100) - 50
    q[:, 2] = rng.choice([2.5, 4.0, 10.0, 15.0, 50.0, 100.0, 200.0, 1000.0], nq)
    lev = np.zeros((nq, 2), np.int32)
    kind = rng.integers(0, 4, nq)
    lo = rng.integers(0, 8, nq)
    lev[:, 0] = np.where(kind == 0, -1, np.where(kind == 3, lo - 1, lo))
    lev[:, 1] = np.where(kind == 0, -1, np.where(kind == 1, lo, np.where(kind == 2, lo + 1, lo + 1)))
    q[:5] = [[-500, 100, 10], [5000, 100, 10], [100, -500, 10], [100, 5000, 10], [un["x"][0], un["y"][0], 0.0]]
    seg, cand = capi.features_in_area(b, un, off, feat, q, lev)
    assert seg[0] == 0 and seg[-1] == len(cand)
    for i in range(nq):
        want = ol.frame_features_in_area(b, un, off, feat, float(q[i, 0]), float(q[i, 1]), float(q[i, 2]), int(lev[i, 0]), int(lev[i, 1]))
        assert np.array_equal(cand[seg[i]:seg[i + 1]], want), i
    assert (np.diff(seg) > 64).any() and (np.diff(seg) == 0).any()
    # capacity protocol: too small a buffer reports the size needed
    with pytest.raises(capi.OrbxError) as e:
        capi.features_in_area(b, un, off, feat, q, lev, cand_cap=10)
    assert e.value.code == capi.ORBX_ERR_CAPACITY
