"""Known-answer tests of the oracle's KeyFrame-to-KeyFrame searches (oracle/search_oracle.cpp: SearchForTriangulation with
CheckDistEpipolarLine, SearchByBoW(KF, KF), the SearchBySim3 agreement check) against plain-Python restatements written from
the reference text (tests/kf_pairs.py), plus the host-side float bound the device kernel uses for the epipolar test."""
import numpy as np
import pytest

import kf_pairs
import oracle_lib as ol
from orb_slam_amd import capi, synth

VOC = synth.vocabulary(6, 3, seed=4)


def _fvs(pr, levelsup=2):
    OV = ol.OracleVocabulary(voc=VOC)
    t1, t2 = OV.transform(pr["d1"], levelsup), OV.transform(pr["d2"], levelsup)
    return (t1[2], t1[3], t1[4]), (t2[2], t2[3], t2[4])


@pytest.mark.parametrize("seed,n1,n2,check", [(1, 160, 200, True), (2, 200, 120, False), (3, 90, 300, True), (4, 1, 40, True), (5, 50, 0, False)])
def test_search_for_triangulation_matches_plain_python(seed, n1, n2, check):
    pr = kf_pairs.pair(seed, n1, n2)
    fv1, fv2 = _fvs(pr)
    got = ol.search_for_triangulation(capi.TH_LOW, check, pr["F"], kf_pairs.LEVEL_SIGMA2, fv1, pr["k1"], pr["d1"], pr["mp1"], fv2, pr["k2"], pr["d2"], pr["mp2"])
    want = kf_pairs.py_search_for_triangulation(capi.TH_LOW, check, pr["F"], kf_pairs.LEVEL_SIGMA2, fv1, pr["k1"], pr["d1"], pr["mp1"], fv2, pr["k2"], pr["d2"], pr["mp2"])
    assert got[0] == want[0]
    np.testing.assert_array_equal(got[1], want[1])
    # inverse map and the per-query distances are consistent with the forward map
    for i1, i2 in enumerate(got[1]):
        if i2 >= 0:
            assert got[2][i2] == i1 and got[4][i1] <= 2 * got[3][i1] and got[4][i1] <= capi.TH_LOW
    assert (got[2] >= 0).sum() == got[0]


def test_triangulation_cases_are_not_trivial():
    pr = kf_pairs.pair(11, 400, 500)
    fv1, fv2 = _fvs(pr)
    a = ol.search_for_triangulation(capi.TH_LOW, False, pr["F"], kf_pairs.LEVEL_SIGMA2, fv1, pr["k1"], pr["d1"], pr["mp1"], fv2, pr["k2"], pr["d2"], pr["mp2"])
    # some queries match, some have candidates but none on the epipolar line, and some matches are NOT the best-distance candidate
    assert a[0] > 30
    has_cand = (a[3] >= 0) & (a[3] < 2 ** 31 - 1)
    assert (has_cand & (a[1] < 0)).sum() > 10
    assert ((a[1] >= 0) & (a[4] > a[3])).sum() > 3


@pytest.mark.parametrize("seed,n1,n2,check", [(21, 150, 180, True), (22, 200, 100, False), (23, 60, 260, True)])
def test_search_by_bow_keyframes_matches_plain_python(seed, n1, n2, check):
    pr = kf_pairs.pair(seed, n1, n2, max_flips=40)
    fv1, fv2 = _fvs(pr)
    v1, v2 = 1 - pr["mp1"], 1 - pr["mp2"]                     # here the flags mean "holds a good map point"
    got = ol.search_by_bow_kf(capi.TH_LOW, 0.6, check, fv1, pr["d1"], pr["k1"]["angle"], v1, fv2, pr["d2"], pr["k2"]["angle"], v2)
    want = kf_pairs.py_search_by_bow_kf(capi.TH_LOW, 0.6, check, fv1, pr["d1"], pr["k1"]["angle"], v1, fv2, pr["d2"], pr["k2"]["angle"], v2)
    assert got[0] == want[0] and got[0] > 5
    np.testing.assert_array_equal(got[1], want[1])


def test_check_dist_epipolar_line_restatements_agree():
    rng = np.random.default_rng(5)
    F = kf_pairs.fundamental(3)
    pr = kf_pairs.pair(9, 300, 300, line_noise=1.5)
    hits = 0
    for i in range(300):
        s2 = kf_pairs.LEVEL_SIGMA2[pr["k2"]["octave"][i]]
        j = rng.integers(0, 300)
        a = ol.check_dist_epipolar_line(pr["k1"]["x"][j], pr["k1"]["y"][j], pr["k2"]["x"][i], pr["k2"]["y"][i], F, s2)
        assert a == kf_pairs.py_epipolar(pr["k1"][j], pr["k2"][i], F, s2)
        hits += a
    assert ol.check_dist_epipolar_line(1.0, 2.0, 3.0, 4.0, np.zeros(9, np.float32), 1.0) is False          # den == 0
    assert 0 <= hits < 300


def test_epipolar_bound_is_the_double_comparison_in_float():
    """orbs_epipolar_bound(s) = the smallest float t with t >= 3.84*(double)s: for every float d, d < t <=> (double)d < 3.84*s"""
    rng = np.random.default_rng(0)
    sig = np.concatenate([kf_pairs.LEVEL_SIGMA2, (rng.random(200) * 50 + 0.01).astype(np.float32), np.float32([0.0, 1e-30, 1e30])])
    for s in sig:
        t = np.float32(capi.epipolar_bound(float(s)))
        T = 3.84 * float(s)
        assert float(t) >= T
        below = np.nextafter(t, np.float32(-np.inf))
        assert float(below) < T or t == 0
        for d in (below, t, np.nextafter(t, np.float32(np.inf))):
            assert (d < t) == (float(d) < T)


def test_sim3_agreement_restatement():
    rng = np.random.default_rng(8)
    for n1, n2 in [(50, 70), (1, 1), (0, 5), (300, 200)]:
        m12 = rng.integers(-1, max(n2, 1), n1).astype(np.int32) if n2 else np.full(n1, -1, np.int32)
        m21 = rng.integers(-1, max(n1, 1), n2).astype(np.int32)
        for i in range(0, min(n1, n2), 3):                       # plant agreeing pairs
            m12[i] = i; m21[i] = i
        n, out = ol.sim3_agreement(m12, m21)
        want = np.array([m12[i] if m12[i] >= 0 and m21[m12[i]] == i else -1 for i in range(n1)], np.int32)
        np.testing.assert_array_equal(out, want)
        assert n == (want >= 0).sum()
