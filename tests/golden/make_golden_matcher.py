#!/usr/bin/env python3
"""Golden outputs of the REFERENCE's own src/ORBmatcher.cc (oracle/_ref/libref_orbmatcher.so: the reference translation unit compiled
where it lies against plain-data stand-ins, oracle/Makefile) on the seeded problems of tests/golden_matcher_inputs.py
-> tests/golden/golden_matcher.npz.  Needs /root/reference (build container only); re-run only on purpose.
The FeatureVectors the vocabulary-node searches walk come from the oracle's DBoW2 restatement (itself pinned to the reference's DBoW2)."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import kf_pairs
import oracle_lib as ol
import golden_matcher_inputs as gi
from orb_slam_amd import capi, synth
import test_ref_pin_matcher as rp            # its ctypes prototypes of the reference wrappers

P = lambda a: a.ctypes.data
b = capi.image_bounds(gi.CAM)
out = {}
# WindowSearch and SearchForInitialization
for name, seed, check in (("window_a", 501, True), ("window_b", 502, False)):
    w = gi.window_problem(seed)
    off, feat = ol.frame_grid(b, w["k2"]); featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
    n1, n2 = len(w["k1"]), len(w["k2"])
    t2q = np.zeros(n2, np.int32)
    n = rp.ref().ref_window_search(ctypes.addressof(b), 0.8, int(check), P(w["k1"]), P(w["d1"]), P(w["state1"]), n1, P(w["k2"]), P(w["d2"]), P(off), P(featp), n2,
                                   w["win"], -1, 2**31 - 1, P(t2q))
    out[name + "_t2q"], out[name + "_n"] = t2q, np.int32(n)
    prev = w["prev"].copy(); q2t = np.zeros(n1, np.int32)
    n = rp.ref().ref_search_for_initialization(ctypes.addressof(b), 0.9, int(check), P(w["k1"]), P(w["d1"]), n1, P(w["k2"]), P(w["d2"]), P(off), P(featp), n2, P(prev),
                                               w["win"], P(q2t))
    out[name + "_init_q2t"], out[name + "_init_n"] = q2t, np.int32(n)
# SearchByProjection(F, vpMapPoints, th)
m = gi.mappoint_problem(511)
off, feat = ol.frame_grid(b, m["k"]); featp = np.ascontiguousarray(np.append(feat, 0).astype(np.int32))
t2q = np.zeros(len(m["k"]), np.int32)
n = rp.ref().ref_search_by_projection_mappoints(ctypes.addressof(b), 0.8, m["th"], P(m["k"]), P(m["desc"]), P(off), P(featp), len(m["k"]), P(m["claimed"]), P(gi.SCALE), 8,
                                                P(m["qxy"]), P(m["qlevel"]), P(m["qcos"]), P(m["qdesc"]), P(m["qstate"]), len(m["qxy"]), P(t2q))
t2q[t2q == -2] = -1
out["mappoints_t2q"], out["mappoints_n"] = t2q, np.int32(n)
# the vocabulary-node searches
OV = ol.OracleVocabulary(voc=synth.vocabulary(**gi.VOC_ARGS))
for name, seed in (("kf_a", 521), ("kf_b", 522)):
    pr = gi.keyframe_pair(seed)
    c = lambda t: (np.ascontiguousarray(t[2], np.uint32), np.ascontiguousarray(t[3], np.int32), np.ascontiguousarray(t[4], np.uint32))
    fv1, fv2 = c(OV.transform(pr["d1"], gi.LEVELSUP)), c(OV.transform(pr["d2"], gi.LEVELSUP))
    n1, n2 = len(pr["d1"]), len(pr["d2"])
    a1, a2 = np.ascontiguousarray(pr["k1"]["angle"]), np.ascontiguousarray(pr["k2"]["angle"])
    F = np.ascontiguousarray(pr["F"].reshape(9))
    for check in (0, 1):
        t2q = np.zeros(n2, np.int32)
        n = rp.ref().ref_search_by_bow(0.75, check, P(fv1[0]), P(fv1[1]), P(fv1[2]), len(fv1[0]), P(pr["d1"]), P(a1), P(pr["s1"]), n1, P(fv2[0]), P(fv2[1]), P(fv2[2]),
                                       len(fv2[0]), P(pr["d2"]), P(a2), n2, P(t2q))
        out["%s_bow_t2q_%d" % (name, check)], out["%s_bow_n_%d" % (name, check)] = t2q, np.int32(n)
        q2t = np.zeros(n1, np.int32)
        n = rp.ref().ref_search_by_bow_kf(0.6, check, P(fv1[0]), P(fv1[1]), P(fv1[2]), len(fv1[0]), P(pr["d1"]), P(a1), P(pr["s1"]), n1, P(fv2[0]), P(fv2[1]), P(fv2[2]),
                                          len(fv2[0]), P(pr["d2"]), P(a2), P(pr["s2"]), n2, P(q2t))
        out["%s_bowkf_q2t_%d" % (name, check)], out["%s_bowkf_n_%d" % (name, check)] = q2t, np.int32(n)
        q2t = np.zeros(n1, np.int32)
        n = rp.ref().ref_search_for_triangulation(0.6, check, P(F), P(kf_pairs.LEVEL_SIGMA2), 8, P(fv1[0]), P(fv1[1]), P(fv1[2]), len(fv1[0]), P(pr["k1"]), P(pr["d1"]),
                                                  P(pr["mp1"]), n1, P(fv2[0]), P(fv2[1]), P(fv2[2]), len(fv2[0]), P(pr["k2"]), P(pr["d2"]), P(pr["mp2"]), n2, P(q2t))
        out["%s_tri_q2t_%d" % (name, check)], out["%s_tri_n_%d" % (name, check)] = q2t, np.int32(n)
np.savez_compressed(os.path.join(ROOT, "tests", "golden", "golden_matcher.npz"), **out)
print({k: (int(v) if v.ndim == 0 else int((v >= 0).sum())) for k, v in out.items()})
