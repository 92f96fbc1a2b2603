"""GPU parity of the KeyFrame-to-KeyFrame searches (include/orbs.h): SearchForTriangulation (ORBS_RULE_TRIANGULATION with the
epipolar test inside the scan), SearchByBoW(KeyFrame*, KeyFrame*) through the list search, and SearchBySim3's agreement check,
against oracle/search_oracle.cpp.  Everything runs on the device from the descriptors on: BoW transform -> merge-walk ranges ->
in-order search."""
import numpy as np
import pytest

import kf_device as kd
import kf_pairs
import oracle_lib as ol
from orb_slam_amd import capi

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("check", [True, False], ids=["rot", "norot"])
def test_search_for_triangulation_pipeline(check):
    pytest.importorskip("torch")
    cap = 1000
    pairs = [kf_pairs.pair(100, 1000, 1000), kf_pairs.pair(101, 1000, 700, line_noise=1.0), kf_pairs.pair(102, 500, 1000, max_flips=40),
             kf_pairs.pair(103, 1000, 1000, p_mp1=0.0, p_mp2=0.0, line_noise=3.0), kf_pairs.pair(104, 1, 300), kf_pairs.pair(105, 40, 0),
             kf_pairs.pair(106, 1000, 1000, max_flips=6, line_noise=0.5)]
    S = kd.setup(pairs, cap)
    q2t, t2q, best, second, nm, nqh = kd.run_triangulation(S, cap, capi.TH_LOW, check, kf_pairs.LEVEL_SIGMA2)
    total = not_best = 0
    for i, p in enumerate(pairs):
        fv1, fv2 = kd.host_fv(S["A"], i), kd.host_fv(S["B"], i)
        w = ol.search_for_triangulation(capi.TH_LOW, check, p["F"], kf_pairs.LEVEL_SIGMA2, fv1, p["k1"], p["d1"], p["mp1"], fv2, p["k2"], p["d2"], p["mp2"])
        assert nm[i] == w[0], (i, nm[i], w[0])
        nqi = nqh[i]
        assert nqi == len(fv1[2])
        pos_feat = fv1[2].astype(np.int64)
        n1, n2 = S["n1"][i], S["n2"][i]
        np.testing.assert_array_equal(kd.by_feature(pos_feat, n1, q2t[i, :nqi]), w[1], err_msg="vMatches12 pair %d" % i)
        np.testing.assert_array_equal(kd.inverse_by_feature(pos_feat, t2q[i, :n2]), w[2], err_msg="inverse pair %d" % i)
        np.testing.assert_array_equal(kd.by_feature(pos_feat, n1, best[i, :nqi]), w[3], err_msg="BestDist pair %d" % i)
        np.testing.assert_array_equal(kd.by_feature(pos_feat, n1, second[i, :nqi]), w[4], err_msg="match distance pair %d" % i)
        total += w[0]
        not_best += int(((w[1] >= 0) & (w[4] > w[3])).sum())
    assert total > 400 and not_best > 5


@pytest.mark.parametrize("check", [True, False], ids=["rot", "norot"])
def test_search_by_bow_keyframes(check):
    """ORBmatcher::SearchByBoW(KeyFrame*, KeyFrame*, ...) = the list search with rule BOW, th = TH_LOW - 1 (that function tests
    `bestDist1 < TH_LOW`) and the pKF2 features WITHOUT a good map point marked as claimed"""
    pytest.importorskip("torch")
    cap = 1000
    pairs = [kf_pairs.pair(200, 1000, 1000, max_flips=40), kf_pairs.pair(201, 800, 1000, max_flips=60), kf_pairs.pair(202, 1000, 300, max_flips=30),
             kf_pairs.pair(203, 0, 100), kf_pairs.pair(204, 1000, 1000, max_flips=50, p_mp1=0.05, p_mp2=0.05)]
    S = kd.setup(pairs, cap)
    q2t, t2q, nm, nqh, V1, V2 = kd.run_bow_kf(S, cap, capi.TH_LOW, 0.6, check)
    total = 0
    for i, p in enumerate(pairs):
        fv1, fv2 = kd.host_fv(S["A"], i), kd.host_fv(S["B"], i)
        n1, n2 = S["n1"][i], S["n2"][i]
        w = ol.search_by_bow_kf(capi.TH_LOW, 0.6, check, fv1, p["d1"], p["k1"]["angle"], V1[i, :n1], fv2, p["d2"], p["k2"]["angle"], V2[i, :n2])
        assert nm[i] == w[0], (i, nm[i], w[0])
        pos_feat = fv1[2].astype(np.int64)
        np.testing.assert_array_equal(kd.by_feature(pos_feat, n1, q2t[i, :nqh[i]]), w[1], err_msg="vpMatches12 pair %d" % i)
        np.testing.assert_array_equal(kd.inverse_by_feature(pos_feat, t2q[i, :n2]), w[2], err_msg="inverse pair %d" % i)
        total += w[0]
    assert total > 300


def test_sim3_agreement_kernel():
    torch = pytest.importorskip("torch")
    rng = np.random.default_rng(3)
    P, cap1, cap2 = 6, 700, 900
    n1 = np.array([700, 650, 1, 0, 300, 700], np.int32); n2 = np.array([900, 100, 1, 50, 0, 900], np.int32)
    m12 = rng.integers(-1, cap2, (P, cap1)).astype(np.int32); m21 = rng.integers(-1, cap1, (P, cap2)).astype(np.int32)
    for i in range(P):
        m12[i, :n1[i]] = np.minimum(m12[i, :n1[i]], max(n2[i], 1) - 1) if n2[i] else -1
        for j in range(0, min(n1[i], n2[i]), 2):
            m12[i, j] = (j * 7) % n2[i]; m21[i, (j * 7) % n2[i]] = j
    d12, d21 = torch.from_numpy(m12).cuda(), torch.from_numpy(m21).cuda()
    out = torch.full((P, cap1), -9, dtype=torch.int32, device="cuda"); nf = torch.full((P,), -9, dtype=torch.int32, device="cuda")
    dn1, dn2 = torch.from_numpy(n1).cuda(), torch.from_numpy(n2).cuda()
    capi.agreement_batch_device(d12.data_ptr(), dn1.data_ptr(), cap1, d21.data_ptr(), dn2.data_ptr(), cap2,
                                P, out.data_ptr(), nf.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    out, nf = out.cpu().numpy(), nf.cpu().numpy()
    found = 0
    for i in range(P):
        n, want = ol.sim3_agreement(m12[i, :n1[i]], m21[i, :n2[i]])
        assert nf[i] == n
        np.testing.assert_array_equal(out[i, :n1[i]], want)
        found += n
    assert found > 100
