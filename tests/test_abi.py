"""C-ABI checks that need no GPU: the library loads, exports exactly what include/orbx.h declares, the pure
host entry points work, and compute entry points FAIL LOUDLY without a device (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np
import pytest

from orb_slam_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_functions():
    src = open(os.path.join(ROOT, "include", "orbx.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(orb[xm]_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    L = capi.lib()
    declared = _header_functions()
    assert len(declared) >= 18
    missing = [f for f in declared if not hasattr(L, f)]
    assert not missing, missing
    assert sorted(capi.EXPORTS) == declared            # the ctypes binding covers the whole header


def test_keypoint_layout_is_opencv24():
    assert capi.KP_DTYPE.itemsize == 28
    assert [capi.KP_DTYPE.fields[n][1] for n in capi.KP_DTYPE.names] == [0, 4, 8, 12, 16, 20, 24]
    assert ctypes.sizeof(capi.Params) == 16 * 4


def test_default_params_are_the_reference_defaults():
    p = capi.Params()
    capi.lib().orbx_default_params(ctypes.byref(p))
    assert (p.nfeatures, p.nlevels, p.score_type, p.fast_th) == (1000, 8, capi.FAST_SCORE, 20)
    assert abs(p.scale_factor - 1.2) < 1e-6 and p.max_batch == 1


def test_host_hamming_and_accept_rule():
    rng = np.random.default_rng(0)
    a, b = rng.integers(0, 256, 32, dtype=np.uint8), rng.integers(0, 256, 32, dtype=np.uint8)
    assert capi.hamming256(a, b) == int(np.unpackbits(a ^ b).sum())
    best = np.array([10, 50, 51, 31, 0, 40], np.int32)
    sec = np.array([20, 90, 200, 50, 0, 2**31 - 1], np.int32)
    # best<=50 && best < 0.6*second :  10<12 yes; 50<54 yes; 51 no (>50); 31<30.000002 no (float math, as the reference); 0<0 no; 40 < huge yes
    assert capi.count_accepted(best, sec, 50, 0.6) == 3


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.mark.skipif(_have_gpu(), reason="checks the no-device behaviour")
def test_no_cpu_fallback_without_device():
    with pytest.raises(capi.OrbxError) as e:
        capi.ORBextractor()
    assert e.value.code == capi.ORBX_ERR_DEVICE
    with pytest.raises(capi.OrbxError) as e:
        capi.match_top2(np.zeros((4, 32), np.uint8), np.zeros((4, 32), np.uint8))
    assert e.value.code == capi.ORBX_ERR_DEVICE


def test_product_never_references_the_oracle():
    """nothing under orb_slam_amd/ or include/ may import, include or link anything under oracle/"""
    bad = []
    for base in ("orb_slam_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".so", ".pyc")):
                    continue
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"oracle_lib|liborb_oracle|#include\s*\"[^\"]*oracle|orc_[a-z]+\(", txt):
                    bad.append(os.path.join(dp, f))
    assert not bad, bad
