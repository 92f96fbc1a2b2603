#!/usr/bin/env python3
"""Create the committed golden fixtures under tests/golden/ from the REFERENCE'S OWN src/ORBextractor.cc.

The reference ships no tests / golden vectors, but its extractor translation unit compiles where it lies against
the stand-in OpenCV headers (oracle/Makefile -> oracle/_ref/libref_orbextractor.so).  The keypoints and
descriptors stored here are that library's outputs on exact-integer synthetic frames: tests/test_golden.py has the oracle
reproduce them on the CPU and the HIP path reproduce them on the GPU with no oracle in the chain — product against reference
output, like tests/golden/make_golden_matcher.py does for ORBmatcher.cc.  (The OpenCV pixel primitives behind the stand-in
headers are restatements; the per-level pyramid hashes are stage artefacts of the oracle, which this script first
checks against the reference's final outputs.)  Runs only where /root/reference exists; the outputs are committed."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as orc
from orb_slam_amd import synth

CASES = [  # name, w, h, family, index, extractor kwargs, store full outputs?
    ("vga_blocks_f0", 640, 480, synth.BLOCKS, 0, dict(nfeatures=1000), True),
    ("qvga_noise_f1", 320, 240, synth.NOISE, 1, dict(nfeatures=300, nlevels=5), True),
    ("vga_noise_f0", 640, 480, synth.NOISE, 0, dict(nfeatures=1000), False),
    ("vga_lowtex_f0", 640, 480, synth.LOWTEX, 0, dict(nfeatures=1000), False),
    ("vga_blocks_f7_nf2000", 640, 480, synth.BLOCKS, 7, dict(nfeatures=2000), False),
    ("vga_blocks_f2_harris", 640, 480, synth.BLOCKS, 2, dict(nfeatures=1000, scoreType=0), False),
    ("hd_blocks_f0", 1920, 1080, synth.BLOCKS, 0, dict(nfeatures=2000), False),
    # round 6: the correlated stream (S-warp): two consecutive frames of a sequence and one deep into the camera path (roll 5 deg, zoom 6 %)
    ("vga_warp_f130", 640, 480, synth.WARP, 130, dict(nfeatures=1000), False),
    ("vga_warp_f131", 640, 480, synth.WARP, 131, dict(nfeatures=1000), False),
    ("vga_warp_f191", 640, 480, synth.WARP, 191, dict(nfeatures=1000), False),
]
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
out = {}
gold = os.path.join(ROOT, "tests", "golden")
os.makedirs(gold, exist_ok=True)
assert orc.ref_available(), "oracle/_ref/libref_orbextractor.so missing: run `make -C oracle ref` where /root/reference exists"
for name, w, h, fam, idx, kw, full in CASES:
    img = synth.frame(w, h, fam, idx)
    k, d = orc.RefExtractor(**kw)(img)                    # /root/reference/src/ORBextractor.cc itself
    o = orc.OracleExtractor(dumps=True, **kw)
    ok, od = o(img)
    assert k.tobytes() == ok.tobytes() and d.tobytes() == od.tobytes(), "oracle and reference disagree on " + name
    nl = kw.get("nlevels", 8)
    rec = {"w": w, "h": h, "family": fam, "index": idx, "kwargs": kw, "n": int(len(k)),
           "made_by": "/root/reference/src/ORBextractor.cc (oracle/_ref/libref_orbextractor.so)",
           "frame_sha256": sha(img), "kps_sha256": sha(k), "desc_sha256": sha(d),
           "pyramid_sha256": [sha(o.level_plane(l, 0)) for l in range(nl)],
           "per_level": np.bincount(k["octave"], minlength=nl).tolist() if len(k) else [0] * nl}
    if full:
        np.savez_compressed(os.path.join(gold, name + ".npz"), kps=k, desc=d)
    out[name] = rec
    print(name, rec["n"], rec["kps_sha256"][:12])
# matcher golden: small problem with ties
Q, T = synth.descriptors(64, 1), synth.descriptors(500, 2)
T[100:400:9] = T[7]
i, b, s = orc.match_top2(Q, T)
np.savez_compressed(os.path.join(gold, "match_64x500.npz"), idx=i, best=b, second=s)
json.dump(out, open(os.path.join(gold, "golden.json"), "w"), indent=1, sort_keys=True)
