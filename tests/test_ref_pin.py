"""The oracle restatement against the REFERENCE'S OWN SOURCES, compiled where they lie (oracle/Makefile → oracle/_ref/).

* libref_orbextractor.so = /root/reference/src/ORBextractor.cc against oracle/cvstub: pins everything ORB_SLAM computes
  itself (grid, quotas, fallback, retain order, IC_Angle, rBRIEF, scale chains).  The OpenCV pixel primitives behind the
  stand-in headers are the oracle's own Appendix-A restatements, so THOSE stay unpinned.
* libref_dbow2.so = the vendored DBoW2 sources: every operation of the bag-of-words path is DBoW2's own → fully pinned.

CPU only.  The .so files are built in the container that has /root/reference and travel with the tree; without them
(and without /root/reference to build them) the module skips."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if not ol.ref_available() and os.path.exists("/root/reference/src/ORBextractor.cc"):
    subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
pytestmark = pytest.mark.skipif(not ol.ref_available(), reason="oracle/_ref not built and /root/reference absent")

EXTRACT_CASES = [
    # w, h, nfeatures, scaleFactor, nlevels, scoreType, fastTh
    (640, 480, 1000, 1.2, 8, ol.FAST_SCORE, 20),      # Data/Settings.yaml defaults (BASELINE config 2)
    (640, 480, 2000, 1.2, 8, ol.FAST_SCORE, 20),      # the initialisation extractor (src/Tracking.cc:126)
    (640, 480, 1000, 1.2, 8, ol.HARRIS_SCORE, 20),
    (752, 480, 1500, 1.3, 6, ol.FAST_SCORE, 12),
    (321, 243, 500, 1.5, 5, ol.HARRIS_SCORE, 9),
    (97, 83, 50, 1.2, 4, ol.FAST_SCORE, 20),
]


@pytest.mark.parametrize("case", EXTRACT_CASES, ids=lambda c: "%dx%d_n%d_s%s_l%d_t%d_th%d" % c)
def test_oracle_extractor_equals_reference_source(case):
    w, h, nf, sf, nl, st, th = case
    ref = ol.RefExtractor(nf, sf, nl, st, th)
    orc = ol.OracleExtractor(nf, sf, nl, st, th)
    for fam in (synth.NOISE, synth.BLOCKS, synth.FLAT, synth.LOWTEX):
        for idx in (0, 5):
            img = synth.frame(w, h, fam, idx)
            rk, rd = ref(img)
            ok, od = orc(img)
            assert len(rk) == len(ok), (fam, idx, len(rk), len(ok))
            assert rk.tobytes() == ok.tobytes(), (fam, idx)            # order, coordinates, size, angle bits, response, octave
            assert rd.tobytes() == od.tobytes(), (fam, idx)


def test_oracle_extractor_equals_reference_source_hd_and_strided():
    ref = ol.RefExtractor(2000)
    orc = ol.OracleExtractor(2000)
    img = synth.frame(1920, 1080, synth.BLOCKS, 3)
    rk, rd = ref(img)
    ok, od = orc(img)
    assert len(rk) == 2000 and rk.tobytes() == ok.tobytes() and rd.tobytes() == od.tobytes()
    # a view into a wider buffer (row stride != width), as a cv::Mat ROI would be
    big = synth.frame(800, 500, synth.BLOCKS, 1)
    view = big[10:490, 30:670]
    R = ol.ref_lib("libref_orbextractor.so")
    cap = 4000
    kps = np.zeros(cap, dtype=ol.KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = R.ref_orb_extract(ref.h, view.ctypes.data, 640, 480, view.strides[0], kps.ctypes.data, desc.ctypes.data, cap)
    ok, od = orc(np.ascontiguousarray(view))
    assert n == len(ok) and kps[:n].tobytes() == ok.tobytes() and desc[:n].tobytes() == od.tobytes()


def test_both_blur_roundings_follow_through_the_reference_source():
    img = synth.frame(640, 480, synth.BLOCKS, 2)
    for mode in (0, 1):
        rk, rd = ol.RefExtractor(1000, blur_mode=mode)(img)
        ok, od = ol.OracleExtractor(1000, blur_mode=mode)(img)
        assert rk.tobytes() == ok.tobytes() and rd.tobytes() == od.tobytes()


def test_forb_distance_is_descriptor_distance():
    R = ol.ref_lib("libref_dbow2.so")
    L = ol.lib()
    d = synth.descriptors(400, 11)
    d[7] = d[3]
    d[9] = 255 - d[3]
    for i in range(0, 399):
        a, b = d[i], d[i + 1]
        want = R.ref_forb_distance(a.ctypes.data, b.ctypes.data)
        assert want == L.orc_hamming256(a.ctypes.data, b.ctypes.data) == L.orc_forb_distance(a.ctypes.data, b.ctypes.data)
        assert want == int(np.unpackbits(a ^ b).sum())
    assert R.ref_forb_distance(d[3].ctypes.data, d[7].ctypes.data) == 0
    assert R.ref_forb_distance(d[3].ctypes.data, d[9].ctypes.data) == 256


VOC_CASES = [
    # k, L, ragged, order, scoring, weighting, min_leaf_level
    (10, 3, False, "bfs", 0, 0, 1),
    (10, 4, True, "kmeans", 0, 0, 2),
    (7, 4, True, "kmeans", 1, 1, 1),
    (5, 5, True, "bfs", 5, 0, 1),
    (9, 3, False, "bfs", 2, 2, 1),
    (12, 3, True, "kmeans", 4, 3, 1),
    (20, 2, True, "bfs", 3, 0, 1),
    (10, 6, True, "kmeans", 0, 0, 2),
]


@pytest.mark.parametrize("case", VOC_CASES, ids=lambda c: "k%d_L%d_%s_%s_s%d_w%d_m%d" % (c[0], c[1], "ragged" if c[2] else "full", c[3], c[4], c[5], c[6]))
def test_oracle_vocabulary_equals_reference_dbow2(case, tmp_path):
    k, L, ragged, order, scoring, weighting, mll = case
    voc = synth.vocabulary(k, L, seed=31 * k + L, ragged=ragged, order=order, min_leaf_level=mll)
    path = str(tmp_path / "voc.txt")
    synth.write_vocabulary_text(path, voc, scoring, weighting)
    ref = ol.RefVocabulary(path)
    orc_file = ol.OracleVocabulary(path=path)
    orc_tab = ol.OracleVocabulary(voc=voc, scoring=scoring, weighting=weighting)
    ri, oi = ref.info(), orc_file.info()
    assert all(ri[key] == oi[key] for key in ri)
    assert oi == orc_tab.info() and oi["n_nodes"] == len(voc["parent"])
    desc = synth.descriptors(1200, 5 + k)
    desc[400:800] = desc[:400]                      # repeated words: the `+=` chains of addWeight
    min_leaf = mll if ragged else L
    for levelsup in (4, 2, 0, L, L + 1):
        nid_level = L - levelsup
        rw, rwt, rn = ref.descend(desc, levelsup)
        for o in (orc_file, orc_tab):
            ow, owt, on = o.descend(desc, levelsup)
            assert np.array_equal(rw, ow) and rwt.tobytes() == owt.tobytes() and np.array_equal(rn, on)
        r = ref.transform(desc, levelsup)
        for o in (orc_file, orc_tab):
            t = o.transform(desc, levelsup)
            assert np.array_equal(r[0], t[0]) and r[1].tobytes() == t[1].tobytes()            # BowVector, bit-exact doubles
            if nid_level <= min_leaf:    # else the reference's `NodeId nid;` is indeterminate for leaves above that level
                assert all(np.array_equal(a, b) for a, b in zip(r[2:], t[2:]))
    a = ref.transform(desc[:700], 4)
    b = ref.transform(desc[300:1200], 4)
    for x, y in ((a, b), (b, a), (a, a)):
        want = ref.score(x[0], x[1], y[0], y[1])
        got = orc_file.score(x[0], x[1], y[0], y[1])
        assert np.float64(want).tobytes() == np.float64(got).tobytes()


def test_empty_inputs_match_reference(tmp_path):
    voc = synth.vocabulary(6, 2, seed=3)
    path = str(tmp_path / "voc.txt")
    synth.write_vocabulary_text(path, voc)
    ref, orc = ol.RefVocabulary(path), ol.OracleVocabulary(path=path)
    for v in (ref, orc):
        t = v.transform(np.zeros((0, 32), np.uint8), 4)
        assert len(t[0]) == 0 and len(t[2]) == 0 and list(t[3]) == [0]
