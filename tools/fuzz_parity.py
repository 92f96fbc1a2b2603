#!/usr/bin/env python3
"""Randomised differential test of the extractor: random image sizes, constructor arguments and image families through
orbx_extract (HIP) and the CPU oracle; keypoints and descriptors must agree byte for byte, geometry errors must agree too.
usage: fuzz_parity.py [cases] [seed]   — prints one JSON line."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import oracle_lib as ol
from orb_slam_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
ok = geo = lim = 0
bad = []
t0 = time.time()
for c in range(cases):
    w = int(rng.integers(120, 1400)); h = int(rng.integers(100, 1000))
    if rng.random() < 0.3:
        w, h = [(640, 480), (752, 480), (1280, 720), (320, 240), (1241, 376), (1920, 1080)][int(rng.integers(0, 6))]
    nf = int(rng.choice([50, 200, 500, 1000, 1500, 2000, 3000]))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.25, 1.3, 1.5, 2.0]))
    nl = int(rng.integers(1, 9))
    st = int(rng.random() < 0.25) ^ 1            # mostly FAST_SCORE (1), sometimes HARRIS_SCORE (0)
    th = int(rng.choice([5, 7, 10, 20, 20, 30, 50]))
    fam = int(rng.choice([0, 1, 1, 1, 3, 4, 5]))
    fpc = bool(rng.random() < 0.25)                  # orbx_params::fp_contract (round 6)
    blur = int(rng.random() < 0.2)
    img = synth.frame(w, h, fam, int(rng.integers(0, 1000)))
    try:
        ex = capi.ORBextractor(nfeatures=nf, scaleFactor=sf, nlevels=nl, scoreType=st, fastTh=th, blur_rounding=blur, fp_contract=fpc)
        got = ex(img)
        ex.close()
    except capi.OrbxError as e:
        if e.code == capi.ORBX_ERR_GEOMETRY:
            geo += 1            # geometry the reference itself cannot process (documented deviation): nothing to compare
            continue
        if e.code == capi.ORBX_ERR_CAPACITY:
            lim += 1            # implementation limit (e.g. one grid cell wider than 2000 px: a handful of features on a 2100-px-wide image)
            continue
        bad.append(dict(case=c, w=w, h=h, nf=nf, sf=sf, nl=nl, st=st, th=th, fam=fam, err=e.code))
        continue
    k, d = ol.OracleExtractor(nf, sf, nl, st, th, blur_mode=blur, fp_contract=fpc)(img)
    if len(k) == len(got[0]) and k.tobytes() == got[0].tobytes() and d.tobytes() == got[1].tobytes():
        ok += 1
    else:
        bad.append(dict(case=c, w=w, h=h, nf=nf, sf=sf, nl=nl, st=st, th=th, fam=fam, blur=blur, n_gpu=len(got[0]), n_oracle=len(k)))
print(json.dumps({"cases": cases, "seed": seed, "bit_exact": ok, "geometry_the_reference_cannot_process": geo, "implementation_limit": lim, "mismatches": bad, "seconds": round(time.time() - t0, 1), "build": capi.build_id()}))
