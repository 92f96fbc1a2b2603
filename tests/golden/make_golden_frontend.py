#!/usr/bin/env python3
"""Golden fixtures for the rows either side of the path (bag of words, Frame-side steps, greedy window search,
distinctive descriptor), made from the CPU oracle on the committed golden keypoints / descriptors of vga_blocks_f0.
The small vocabulary is committed as the reference's text format (voc_k6_L3.txt).  Re-run only on purpose."""
import hashlib, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as ol
from orb_slam_amd import capi, synth
from golden_inputs import frontend_inputs

gold = os.path.join(ROOT, "tests", "golden")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()
vpath = os.path.join(gold, "voc_k6_L3.txt")
if not os.path.exists(vpath):
    synth.write_vocabulary_text(vpath, synth.vocabulary(6, 3, seed=11, order="kmeans"))
I = frontend_inputs()
out = {}
voc = ol.OracleVocabulary(path=vpath)
t = voc.transform(I["desc"], 2)
out["bow"] = {"n_bow": int(len(t[0])), "n_fv": int(len(t[2])), "ids": sha(t[0]), "vals": sha(t[1]), "fv_node": sha(t[2]), "fv_off": sha(t[3]), "fv_feat": sha(t[4])}
b = ol.frame_bounds(I["cam"], capi.Bounds)
un = ol.frame_undistort(I["cam"], I["kps"])
off, feat = ol.frame_grid(b, un)
out["frame"] = {"bounds": list(b.astuple()[:4]), "un": sha(un), "off": sha(off), "feat": sha(feat), "in_grid": int(off[-1])}
out["search"] = {}
for rule, th, ratio, check in I["rules"]:
    r = ol.window_search(b, rule, th, ratio, check, un, I["desc"], off, feat, I["claimed"] if rule == 0 else None, I["qxyr"], I["qlev"], I["qdesc"], I["qangle"], I["qvalid"])
    out["search"]["rule%d" % rule] = {"nmatches": int(r[0]), "q2t": sha(r[1]), "t2q": sha(r[2]), "best": sha(r[3]), "second": sha(r[4])}
segs = I["seg_off"]
out["distinctive"] = [list(map(int, ol.distinctive(I["desc"][segs[p]:segs[p + 1]]))) for p in range(len(segs) - 1)]
json.dump(out, open(os.path.join(gold, "golden_frontend.json"), "w"), indent=1, sort_keys=True)
print(json.dumps(out)[:400])
