"""Golden fixtures made by the REFERENCE's own src/ORBmatcher.cc (tests/golden/golden_matcher.npz, tests/golden/make_golden_matcher.py:
the reference translation unit compiled where it lies, run in the build container).  CPU: the oracle reproduces them.  GPU: the HIP
searches reproduce them with no oracle in the loop — WindowSearch, SearchForInitialization, SearchByProjection(F, vpMapPoints, th),
SearchByBoW (KeyFrame-Frame, KeyFrame-KeyFrame) and SearchForTriangulation, with and without the rotation check."""
import os

import numpy as np
import pytest

import golden_matcher_inputs as gi
import kf_pairs
from orb_slam_amd import capi, synth

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_matcher.npz"))
WINDOWS = (("window_a", 501, True), ("window_b", 502, False))
KFS = (("kf_a", 521), ("kf_b", 522))


def test_oracle_reproduces_reference_golden():
    import oracle_lib as ol
    b = capi.image_bounds(gi.CAM)
    for name, seed, check in WINDOWS:
        w = gi.window_problem(seed)
        off, feat = ol.frame_grid(b, w["k2"])
        n1 = len(w["k1"]); lvl = w["k1"]["octave"]
        r = ol.window_search(b, capi.RULE_WINDOW, capi.TH_HIGH, 0.8, check, w["k2"], w["d2"], off, feat, None,
                             np.stack([w["k1"]["x"], w["k1"]["y"], np.full(n1, w["win"], np.float32)], -1), np.stack([lvl, lvl], -1), w["d1"], w["k1"]["angle"],
                             (w["state1"] == 1).astype(np.uint8))
        assert r[0] == G[name + "_n"] and np.array_equal(r[2], G[name + "_t2q"])
        r = ol.window_search(b, capi.RULE_INIT, capi.TH_LOW, 0.9, check, w["k2"], w["d2"], off, feat, None,
                             np.concatenate([w["prev"], np.full((n1, 1), w["win"], np.float32)], 1), np.zeros((n1, 2), np.int32), w["d1"], w["k1"]["angle"],
                             (lvl == 0).astype(np.uint8))
        assert r[0] == G[name + "_init_n"] and np.array_equal(r[1], G[name + "_init_q2t"])
    m = gi.mappoint_problem(511)
    off, feat = ol.frame_grid(b, m["k"])
    r = ol.window_search(b, capi.RULE_MAPPOINTS, capi.TH_HIGH, 0.8, False, m["k"], m["desc"], off, feat, m["claimed"], np.concatenate([m["qxy"], m["R"][:, None]], 1),
                         np.stack([m["qlevel"] - 1, m["qlevel"]], -1), m["qdesc"], None, (m["qstate"] == 1).astype(np.uint8))
    assert r[0] == G["mappoints_n"] and np.array_equal(r[2], G["mappoints_t2q"])
    OV = ol.OracleVocabulary(voc=synth.vocabulary(**gi.VOC_ARGS))
    for name, seed in KFS:
        pr = gi.keyframe_pair(seed)
        t1, t2 = OV.transform(pr["d1"], gi.LEVELSUP), OV.transform(pr["d2"], gi.LEVELSUP)
        fv1, fv2 = (t1[2], t1[3], t1[4]), (t2[2], t2[3], t2[4])
        for check in (0, 1):
            r = ol.search_by_bow(capi.TH_LOW, 0.75, check, fv1, pr["d1"], pr["k1"]["angle"], (pr["s1"] == 1).astype(np.uint8), fv2, pr["d2"], pr["k2"]["angle"])
            assert r[0] == G["%s_bow_n_%d" % (name, check)] and np.array_equal(r[2], G["%s_bow_t2q_%d" % (name, check)])
            r = ol.search_by_bow_kf(capi.TH_LOW, 0.6, check, fv1, pr["d1"], pr["k1"]["angle"], (pr["s1"] == 1).astype(np.uint8), fv2, pr["d2"], pr["k2"]["angle"],
                                    (pr["s2"] == 1).astype(np.uint8))
            assert r[0] == G["%s_bowkf_n_%d" % (name, check)] and np.array_equal(r[1], G["%s_bowkf_q2t_%d" % (name, check)])
            r = ol.search_for_triangulation(capi.TH_LOW, check, pr["F"], kf_pairs.LEVEL_SIGMA2, fv1, pr["k1"], pr["d1"], pr["mp1"], fv2, pr["k2"], pr["d2"], pr["mp2"])
            assert r[0] == G["%s_tri_n_%d" % (name, check)] and np.array_equal(r[1], G["%s_tri_q2t_%d" % (name, check)])


def _device_window_search(rule, th, ratio, check, kt, dt, claimed, qxyr, qlev, qdesc, qangle, qvalid):
    """one problem through orbf_undistort_grid_batch_device + orbs_window_search_batch_device (no oracle involved)"""
    import torch
    b = capi.image_bounds(gi.CAM)
    nt, nq = len(kt), len(qxyr)
    cap = max(nt, nq)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()
    pad = lambda a: np.concatenate([a, np.zeros((cap - len(a),) + a.shape[1:], a.dtype)])
    st = torch.cuda.current_stream().cuda_stream
    dK = T(pad(kt).view(np.uint8).reshape(cap, 28)); dnt, dnq = T(np.array([nt], np.int32)), T(np.array([nq], np.int32))
    dUn = torch.zeros((cap, 28), dtype=torch.uint8, device="cuda"); dOff = torch.zeros(capi.GRID_CELLS + 1, dtype=torch.int32, device="cuda")
    dFeat = torch.zeros(cap, dtype=torch.int32, device="cuda")
    capi.undistort_grid_batch_device(gi.CAM, b, dK.data_ptr(), dnt.data_ptr(), 1, cap, dUn.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), st)
    dD, dC = T(pad(dt)), T(pad(claimed if claimed is not None else np.zeros(nt, np.uint8)))
    dQX, dQL, dQD = T(pad(qxyr.astype(np.float32))), T(pad(qlev.astype(np.int32))), T(pad(qdesc))
    dQA = T(pad(qangle.astype(np.float32) if qangle is not None else np.zeros(nq, np.float32))); dQV = T(pad(qvalid.astype(np.uint8)))
    o = [torch.full((cap,), -9, dtype=torch.int32, device="cuda") for _ in range(4)]; nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    capi.window_search_batch_device(b, rule, th, ratio, check, dUn.data_ptr(), dD.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), dnt.data_ptr(), cap,
                                    dC.data_ptr() if claimed is not None else 0, dQX.data_ptr(), dQL.data_ptr(), dQD.data_ptr(), dQA.data_ptr(), dQV.data_ptr(),
                                    dnq.data_ptr(), cap, 1, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(), nm.data_ptr(), st)
    torch.cuda.synchronize()
    return int(nm.item()), o[0].cpu().numpy()[:nq], o[1].cpu().numpy()[:nt]


@pytest.mark.gpu
def test_gpu_window_searches_reproduce_reference_golden():
    pytest.importorskip("torch")
    for name, seed, check in WINDOWS:
        w = gi.window_problem(seed)
        n1 = len(w["k1"]); lvl = w["k1"]["octave"]
        n, q2t, t2q = _device_window_search(capi.RULE_WINDOW, capi.TH_HIGH, 0.8, check, w["k2"], w["d2"], None,
                                            np.stack([w["k1"]["x"], w["k1"]["y"], np.full(n1, w["win"], np.float32)], -1), np.stack([lvl, lvl], -1), w["d1"],
                                            w["k1"]["angle"], w["state1"] == 1)
        assert n == G[name + "_n"] and np.array_equal(t2q, G[name + "_t2q"])
        n, q2t, t2q = _device_window_search(capi.RULE_INIT, capi.TH_LOW, 0.9, check, w["k2"], w["d2"], None,
                                            np.concatenate([w["prev"], np.full((n1, 1), w["win"], np.float32)], 1), np.zeros((n1, 2), np.int32), w["d1"],
                                            w["k1"]["angle"], lvl == 0)
        assert n == G[name + "_init_n"] and np.array_equal(q2t, G[name + "_init_q2t"])
    m = gi.mappoint_problem(511)
    n, q2t, t2q = _device_window_search(capi.RULE_MAPPOINTS, capi.TH_HIGH, 0.8, False, m["k"], m["desc"], m["claimed"], np.concatenate([m["qxy"], m["R"][:, None]], 1),
                                        np.stack([m["qlevel"] - 1, m["qlevel"]], -1), m["qdesc"], None, m["qstate"] == 1)
    assert n == G["mappoints_n"] and np.array_equal(t2q, G["mappoints_t2q"])


@pytest.mark.gpu
@pytest.mark.parametrize("check", [0, 1])
def test_gpu_vocabulary_node_searches_reproduce_reference_golden(check):
    torch = pytest.importorskip("torch")
    import kf_device as kd
    cap = 1000
    raw = [gi.keyframe_pair(seed) for _, seed in KFS]
    voc = synth.vocabulary(**gi.VOC_ARGS)
    # SearchForTriangulation: "has a map point" flags as they are
    S = kd.setup(raw, cap, voc=voc, k=gi.VOC_ARGS["k"], L=gi.VOC_ARGS["L"], levelsup=gi.LEVELSUP)
    q2t, t2q, best, second, nm, nqh = kd.run_triangulation(S, cap, capi.TH_LOW, bool(check), kf_pairs.LEVEL_SIGMA2)
    for i, (name, _) in enumerate(KFS):
        pos = kd.host_fv(S["A"], i)[2].astype(np.int64)
        assert nm[i] == G["%s_tri_n_%d" % (name, check)]
        np.testing.assert_array_equal(kd.by_feature(pos, S["n1"][i], q2t[i, :nqh[i]]), G["%s_tri_q2t_%d" % (name, check)])
    S["V"].close()
    # the two SearchByBoW: validity = "holds a good map point" (kd derives it as 1 - mp flag)
    bow = [dict(p, mp1=(p["s1"] != 1).astype(np.uint8), mp2=(p["s2"] != 1).astype(np.uint8)) for p in raw]
    S = kd.setup(bow, cap, voc=voc, k=gi.VOC_ARGS["k"], L=gi.VOC_ARGS["L"], levelsup=gi.LEVELSUP)
    q2t, t2q, nm, nqh, V1, V2 = kd.run_bow_kf(S, cap, capi.TH_LOW, 0.6, bool(check))
    for i, (name, _) in enumerate(KFS):
        pos = kd.host_fv(S["A"], i)[2].astype(np.int64)
        assert nm[i] == G["%s_bowkf_n_%d" % (name, check)]
        np.testing.assert_array_equal(kd.by_feature(pos, S["n1"][i], q2t[i, :nqh[i]]), G["%s_bowkf_q2t_%d" % (name, check)])
    # KeyFrame-Frame: nothing claimed on the frame side
    P, st = S["P"], S["st"]
    o = kd.outputs(torch, P, cap)
    dV1 = torch.from_numpy(V1).cuda(); dA1 = torch.from_numpy(np.ascontiguousarray(S["K1"]["angle"])).cuda()
    capi.list_search_batch_device(capi.RULE_BOW, capi.TH_LOW, 0.75, bool(check), S["dK2"].data_ptr(), S["B"]["D"].data_ptr(), S["B"]["feat"].data_ptr(),
                                  S["nlist"].data_ptr(), S["B"]["n"].data_ptr(), cap, 0, S["qrange"].data_ptr(), S["A"]["feat"].data_ptr(), S["A"]["D"].data_ptr(),
                                  dA1.data_ptr(), dV1.data_ptr(), S["nq"].data_ptr(), cap, P, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), o[3].data_ptr(),
                                  o[4].data_ptr(), st)
    torch.cuda.synchronize()
    t2q, nm = o[1].cpu().numpy(), o[4].cpu().numpy()
    for i, (name, _) in enumerate(KFS):
        pos = kd.host_fv(S["A"], i)[2].astype(np.int64)
        assert nm[i] == G["%s_bow_n_%d" % (name, check)]
        np.testing.assert_array_equal(kd.inverse_by_feature(pos, t2q[i, :S["n2"][i]]), G["%s_bow_t2q_%d" % (name, check)])
    S["V"].close()
