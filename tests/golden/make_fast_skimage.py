#!/usr/bin/env python3
"""Third-party cross-check of the FAST-9 corner criterion, independent of this repository and of OpenCV:
scikit-image 0.18.3's `skimage.feature.corner_fast(image, n=9, threshold=t)` (the Rosten-Drummond segment test as
implemented in skimage/feature/corner_cy.pyx: >= n contiguous ring pixels all > p + t or all < p - t, strict).
That is the set of pixels cv::FAST(threshold=t) treats as corners BEFORE its non-maximum suppression.

scikit-image is not installed for the test interpreter; it happens to exist in this image's /opt/conda (python 3.9):
    /opt/conda/bin/python3.9 tests/golden/make_fast_skimage.py
writes tests/golden/fast_skimage.npz (bit-packed corner masks), which tests/test_oracle_kat.py compares with the oracle."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import skimage
from skimage.feature import corner_fast
from orb_slam_amd import synth

assert skimage.__version__.startswith("0.18"), skimage.__version__
out = {}
for name, w, h, fam, idx in (("blocks", 320, 240, synth.BLOCKS, 0), ("noise", 160, 120, synth.NOISE, 3), ("lowtex", 200, 150, synth.LOWTEX, 1)):
    img = synth.frame(w, h, fam, idx)
    for t in (20, 7):
        # float input keeps the integer scale (no /255), so `> p + t` is an exact integer comparison in float64
        resp = corner_fast(img.astype(np.float64), n=9, threshold=float(t))
        mask = resp > 0
        out["%s_t%d" % (name, t)] = np.packbits(mask)
        out["%s_shape" % name] = np.array([h, w, fam, idx])
        print(name, t, int(mask.sum()))
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "fast_skimage.npz"), **out)
