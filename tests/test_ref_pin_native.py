"""The FMA caveat, pinned.  The reference's own build (CMakeLists.txt:12-13: -O3 -march=native, GCC's default -ffp-contract=fast) fuses the
multiply-adds of two float expressions of src/ORBextractor.cc — `x*b + y*a` / `x*a - y*b` in computeOrbDescriptor (:165-166) and the Harris
response (:118-119); everything else in this repository (oracle, oracle/_ref, golden fixtures, the device's default mode) evaluates them
unfused.  oracle/_ref_native/libref_orbextractor.so is the same translation unit with the reference's flags.  Here:
  * how much the two builds of the REFERENCE differ (>= 500 frames): a handful of descriptor bits per 10^5 key points, a third of the Harris
    responses in their last bits;
  * the oracle's fp_contract mode reproduces the native build byte for byte (the 7 x 4 configurations of tests/test_ref_pin.py + strided +
    both blur roundings), as its default mode reproduces the contraction-off build.
The device's ORBX_FP_CONTRACT mode against the same native build: tests/test_gpu_parity.py::test_fp_contract_mode_equals_the_native_reference_build."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import synth
from test_ref_pin import EXTRACT_CASES

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if not ol.native_available() and os.path.exists("/root/reference/src/ORBextractor.cc"):
    subprocess.call(["make", "-C", os.path.join(ROOT, "oracle"), "ref"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
pytestmark = pytest.mark.skipif(not (ol.native_available() and ol.ref_available()), reason="oracle/_ref_native not built and /root/reference absent")


def test_the_two_builds_really_differ_in_their_object_code():
    def fused(path):
        out = subprocess.run(["objdump", "-d", path], capture_output=True, text=True).stdout
        return sum(1 for l in out.splitlines() if "\tvfm" in l or "\tvfnm" in l)
    assert fused(os.path.join(ol.NATIVE_DIR, "libref_orbextractor.so")) >= 30
    assert fused(os.path.join(ol.REF_DIR, "libref_orbextractor.so")) == 0


@pytest.mark.parametrize("case", EXTRACT_CASES, ids=lambda c: "%dx%d_n%d_s%s_l%d_t%d_th%d" % c)
def test_oracle_fp_contract_mode_equals_the_native_build(case):
    w, h, nf, sf, nl, st, th = case
    ref = ol.RefExtractor(nf, sf, nl, st, th, native=True)
    orc = ol.OracleExtractor(nf, sf, nl, st, th, fp_contract=True)
    for fam in (synth.NOISE, synth.BLOCKS, synth.FLAT, synth.LOWTEX):
        for idx in (0, 5):
            img = synth.frame(w, h, fam, idx)
            rk, rd = ref(img)
            ok, od = orc(img)
            assert len(rk) == len(ok), (fam, idx, len(rk), len(ok))
            assert rk.tobytes() == ok.tobytes(), (fam, idx)
            assert rd.tobytes() == od.tobytes(), (fam, idx)


def test_native_build_hd_strided_and_both_blur_roundings():
    img = synth.frame(1920, 1080, synth.BLOCKS, 3)
    rk, rd = ol.RefExtractor(2000, native=True)(img)
    ok, od = ol.OracleExtractor(2000, fp_contract=True)(img)
    assert len(rk) == 2000 and rk.tobytes() == ok.tobytes() and rd.tobytes() == od.tobytes()
    img = synth.frame(640, 480, synth.BLOCKS, 2)
    for mode in (0, 1):
        rk, rd = ol.RefExtractor(1000, blur_mode=mode, native=True)(img)
        ok, od = ol.OracleExtractor(1000, blur_mode=mode, fp_contract=True)(img)
        assert rk.tobytes() == ok.tobytes() and rd.tobytes() == od.tobytes()


def test_divergence_between_the_reference_builds_is_small_and_reported():
    """>= 500 VGA frames, FAST and Harris score: what `bit-exact against the reference` leaves open when the reference's flags are not stated"""
    report = {}
    for st, name, frames in ((ol.FAST_SCORE, "fast", 520), (ol.HARRIS_SCORE, "harris", 120)):
        iso, nat = ol.RefExtractor(1000, scoreType=st), ol.RefExtractor(1000, scoreType=st, native=True)
        kp = bits = frames_diff = resp = order = 0
        for i in range(frames):
            img = synth.frame(640, 480, [synth.BLOCKS, synth.NOISE, synth.LOWTEX, synth.MIDTEX][i % 4], i // 4)
            k0, d0 = iso(img)
            k1, d1 = nat(img)
            assert len(k0) == len(k1)
            kp += len(k0)
            same_sel = np.array_equal(k0["x"], k1["x"]) and np.array_equal(k0["y"], k1["y"]) and np.array_equal(k0["octave"], k1["octave"])
            order += not same_sel
            if same_sel:
                assert np.array_equal(k0["angle"].view(np.uint32), k1["angle"].view(np.uint32))         # IC_Angle is integer moments + OpenCV's fastAtan2
                b = int(np.unpackbits(d0 ^ d1).sum())
                bits += b
                frames_diff += b > 0
                resp += int((k0["response"].view(np.uint32) != k1["response"].view(np.uint32)).sum())
        report[name] = dict(frames=frames, keypoints=kp, descriptor_bits_differing=bits, frames_with_a_differing_bit=frames_diff,
                            responses_differing=resp, frames_with_another_selection=order)
    print("fma divergence between the two builds of the reference:", report)
    f, hr = report["fast"], report["harris"]
    assert f["keypoints"] > 400_000 and f["responses_differing"] == 0 and f["frames_with_another_selection"] == 0
    assert 0 < f["descriptor_bits_differing"] < f["keypoints"] * 1e-3            # a few bits per 10^5 key points: real, small
    assert hr["responses_differing"] > 0.05 * hr["keypoints"]                      # last-bit differences of the Harris response
