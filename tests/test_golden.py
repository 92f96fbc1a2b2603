"""Committed golden vectors (tests/golden/, made by tests/golden/make_golden.py from the reference's own src/ORBextractor.cc
compiled against the stand-in OpenCV headers): CPU: the oracle reproduces them; GPU: the HIP path reproduces them with no
oracle in the chain."""
import hashlib
import json
import os

import numpy as np
import pytest

from orb_slam_amd import capi, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = json.load(open(os.path.join(GOLD, "golden.json")))
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _kw_oracle(kw):
    return dict(kw)


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_reproduces_golden(name):
    import oracle_lib as orc
    c = CASES[name]
    if c["w"] > 1000:
        pytest.importorskip("numpy")
    img = synth.frame(c["w"], c["h"], c["family"], c["index"])
    assert sha(img) == c["frame_sha256"]                  # the synthetic generator itself is pinned
    o = orc.OracleExtractor(dumps=True, **c["kwargs"])
    k, d = o(img)
    assert len(k) == c["n"] and sha(k) == c["kps_sha256"] and sha(d) == c["desc_sha256"]
    assert [sha(o.level_plane(l, 0)) for l in range(len(c["pyramid_sha256"]))] == c["pyramid_sha256"]
    f = os.path.join(GOLD, name + ".npz")
    if os.path.exists(f):
        z = np.load(f)
        assert z["kps"].tobytes() == k.tobytes() and np.array_equal(z["desc"], d)


def test_oracle_reproduces_match_golden():
    import oracle_lib as orc
    Q, T = synth.descriptors(64, 1), synth.descriptors(500, 2)
    T[100:400:9] = T[7]
    z = np.load(os.path.join(GOLD, "match_64x500.npz"))
    i, b, s = orc.match_top2(Q, T)
    assert np.array_equal(i, z["idx"]) and np.array_equal(b, z["best"]) and np.array_equal(s, z["second"])


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_gpu_reproduces_golden(name, gpu_extractor_factory):
    c = CASES[name]
    img = synth.frame(c["w"], c["h"], c["family"], c["index"])
    k, d = gpu_extractor_factory(**c["kwargs"])(img)
    assert len(k) == c["n"]
    assert np.bincount(k["octave"], minlength=len(c["per_level"])).tolist() == c["per_level"] if len(k) else True
    assert sha(k) == c["kps_sha256"] and sha(d) == c["desc_sha256"]


@pytest.mark.gpu
def test_gpu_reproduces_match_golden():
    Q, T = synth.descriptors(64, 1), synth.descriptors(500, 2)
    T[100:400:9] = T[7]
    z = np.load(os.path.join(GOLD, "match_64x500.npz"))
    i, b, s = capi.match_top2(Q, T)
    assert np.array_equal(i, z["idx"]) and np.array_equal(b, z["best"]) and np.array_equal(s, z["second"])
