"""CPU side of the matcher drop-in (oracle/_ref/libprod_orbmatcher.so = orb_slam_amd/cpp/ORBmatcher.cc behind the reference harness): the
library loads without an oracle dependency, its host-only members equal the reference's, and a search WITHOUT a GPU fails loudly
(std::runtime_error out of the class, no CPU fallback).  The searches themselves: tests/test_gpu_orbmatcher_dropin.py."""
import os
import subprocess
import sys

import numpy as np
import pytest

import kf_pairs
import test_ref_pin_matcher as trm
from orb_slam_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = os.path.join(ROOT, "oracle", "_ref", "libref_orbmatcher.so")
PROD = os.path.join(ROOT, "oracle", "_ref", "libprod_orbmatcher.so")
pytestmark = pytest.mark.skipif(not (os.path.exists(REF) and os.path.exists(PROD)), reason="oracle/_ref/lib{ref,prod}_orbmatcher.so are built where /root/reference exists")


PROD_REAL_ACCESS = os.path.join(ROOT, "oracle", "_ref", "libprod_orbmatcher_realaccess.so")      # the same with orb_slam_amd/cpp/ORBmatcherAccess.h


@pytest.mark.parametrize("PROD", [PROD, PROD_REAL_ACCESS], ids=["stand-in access header", "ORBmatcherAccess.h"])
def test_product_library_links_the_c_abi_and_no_oracle(PROD):
    if not os.path.exists(PROD):
        pytest.skip(PROD + " is built where /root/reference exists")
    out = subprocess.run(["ldd", PROD], capture_output=True, text=True).stdout
    assert "liborbx.so" in out and "orb_oracle" not in out and "libref_" not in out
    syms = subprocess.run(["nm", "-D", "--undefined-only", PROD], capture_output=True, text=True).stdout
    assert "orbs_window_search_batch_device" in syms and "orbs_list_search_batch_device" in syms and "orbs_triangulation_search_batch_device" in syms
    assert "orbs_agreement_batch_device" in syms and "orc_" not in syms


def test_host_members_equal_the_reference():
    r, p = trm.load(REF), trm.load(PROD)
    rng = np.random.default_rng(5)
    for _ in range(200):
        h = rng.integers(0, int(rng.choice([2, 5, 40, 400])), 30).astype(np.int32)
        a, b = np.zeros(3, np.int32), np.zeros(3, np.int32)
        r.ref_matcher_three_maxima(h.ctypes.data, 30, a.ctypes.data)
        p.ref_matcher_three_maxima(h.ctypes.data, 30, b.ctypes.data)
        assert tuple(a) == tuple(b)
    d = synth.descriptors(300, 8)
    for i in range(0, 300, 2):
        assert r.ref_matcher_descriptor_distance(d[i].ctypes.data, d[i + 1].ctypes.data) == p.ref_matcher_descriptor_distance(d[i].ctypes.data, d[i + 1].ctypes.data)
    pr = kf_pairs.pair(3, 300, 300, line_noise=1.5)
    F = np.ascontiguousarray(pr["F"].reshape(9))
    s2 = kf_pairs.LEVEL_SIGMA2
    hits = 0
    for i in range(300):
        j = int(rng.integers(0, 300))
        args = (pr["k1"]["x"][j], pr["k1"]["y"][j], pr["k2"]["x"][i], pr["k2"]["y"][i], int(pr["k2"]["octave"][i]), F.ctypes.data, s2.ctypes.data, 8)
        a = r.ref_matcher_check_epipolar(*args)
        assert a == p.ref_matcher_check_epipolar(*args)
        hits += a
    assert 5 < hits < 295


def test_a_search_without_a_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present: the searches run (tests/test_gpu_orbmatcher_dropin.py)")
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
            "import test_ref_pin_matcher as trm\n"
            "L = trm.load(%r)\n"
            "trm.ref = lambda: L\n"
            "trm.test_window_search(14, 2, 1, 100, True, -1, 2**31 - 1, False)\n" % (ROOT, os.path.join(ROOT, "tests"), PROD))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "ORB_SLAM::ORBmatcher" in r.stderr and "no usable MI355X" in r.stderr, r.stderr[-2000:]
