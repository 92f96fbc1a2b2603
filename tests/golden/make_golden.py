#!/usr/bin/env python3
"""Create the committed golden fixtures under tests/golden/ from the CPU oracle.

The reference has no tests / golden vectors of its own and cannot be built here, so these
vectors are OUR oracle's outputs on exact-integer synthetic frames: they pin the oracle against drift and
let the GPU tests run without the oracle.  Re-run only when the oracle definition changes on purpose."""
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
]
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
out = {}
gold = os.path.join(ROOT, "tests", "golden")
os.makedirs(gold, exist_ok=True)
for name, w, h, fam, idx, kw, full in CASES:
    img = synth.frame(w, h, fam, idx)
    o = orc.OracleExtractor(dumps=True, **kw)
    k, d = o(img)
    nl = kw.get("nlevels", 8)
    rec = {"w": w, "h": h, "family": fam, "index": idx, "kwargs": kw, "n": int(len(k)),
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
