"""Committed golden vectors of the rows either side of the path (tests/golden/golden_frontend.json, made by
tests/golden/make_golden_frontend.py from the oracle).  CPU: the oracle still reproduces them, and — where oracle/_ref is
available — so does the reference's own DBoW2.  GPU: the HIP path reproduces them without the oracle."""
import hashlib
import json
import os

import numpy as np
import pytest

from golden_inputs import GOLD, frontend_inputs
from orb_slam_amd import capi

G = json.load(open(os.path.join(GOLD, "golden_frontend.json")))
VOC = os.path.join(GOLD, "voc_k6_L3.txt")
sha = lambda a: hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def _bow_matches(t):
    g = G["bow"]
    return (len(t[0]), len(t[2]), sha(t[0]), sha(t[1]), sha(t[2]), sha(t[3]), sha(t[4])) == (
        g["n_bow"], g["n_fv"], g["ids"], g["vals"], g["fv_node"], g["fv_off"], g["fv_feat"])


def test_oracle_reproduces_frontend_golden():
    import oracle_lib as ol
    I = frontend_inputs()
    assert _bow_matches(ol.OracleVocabulary(path=VOC).transform(I["desc"], 2))
    if ol.ref_available():
        assert _bow_matches(ol.RefVocabulary(VOC).transform(I["desc"], 2))            # the reference's own DBoW2
    b = ol.frame_bounds(I["cam"], capi.Bounds)
    un = ol.frame_undistort(I["cam"], I["kps"])
    off, feat = ol.frame_grid(b, un)
    g = G["frame"]
    assert list(b.astuple()[:4]) == g["bounds"] and (sha(un), sha(off), sha(feat), int(off[-1])) == (g["un"], g["off"], g["feat"], g["in_grid"])
    for rule, th, ratio, check in I["rules"]:
        r = ol.window_search(b, rule, th, ratio, check, un, I["desc"], off, feat, I["claimed"] if rule == 0 else None, I["qxyr"], I["qlev"],
                             I["qdesc"], I["qangle"], I["qvalid"])
        gs = G["search"]["rule%d" % rule]
        assert (int(r[0]), sha(r[1]), sha(r[2]), sha(r[3]), sha(r[4])) == (gs["nmatches"], gs["q2t"], gs["t2q"], gs["best"], gs["second"]), rule
    segs = I["seg_off"]
    assert [list(map(int, ol.distinctive(I["desc"][segs[p]:segs[p + 1]]))) for p in range(len(segs) - 1)] == G["distinctive"]


@pytest.mark.gpu
def test_gpu_reproduces_frontend_golden():
    torch = pytest.importorskip("torch")
    I = frontend_inputs()
    V = capi.ORBVocabulary.loadFromTextFile(VOC)
    assert _bow_matches(V.transform(I["desc"], 2))
    b = capi.image_bounds(I["cam"])
    un, off, feat = capi.undistort_grid(I["cam"], b, I["kps"])
    g = G["frame"]
    assert list(b.astuple()[:4]) == g["bounds"] and (sha(un), sha(off), sha(feat), int(off[-1])) == (g["un"], g["off"], g["feat"], g["in_grid"])
    n = len(un)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
    dUn, dD = t(un.view(np.uint8).reshape(n, 28)), t(I["desc"])
    dOff, dFeat = t(off), t(np.concatenate([feat, np.zeros(n - len(feat), np.int32)]))
    dn = t(np.array([n], np.int32))
    dQX, dQL, dQD, dQA, dQV, dC = t(I["qxyr"]), t(I["qlev"]), t(I["qdesc"]), t(I["qangle"]), t(I["qvalid"]), t(I["claimed"])
    for rule, th, ratio, check in I["rules"]:
        q2t = torch.zeros(n, dtype=torch.int32, device="cuda"); t2q = torch.zeros(n, dtype=torch.int32, device="cuda")
        best = torch.zeros(n, dtype=torch.int32, device="cuda"); second = torch.zeros(n, dtype=torch.int32, device="cuda")
        nm = torch.zeros(1, dtype=torch.int32, device="cuda")
        capi.window_search_batch_device(b, rule, th, ratio, check, dUn.data_ptr(), dD.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), dn.data_ptr(), n,
                                        dC.data_ptr() if rule == 0 else 0, dQX.data_ptr(), dQL.data_ptr(), dQD.data_ptr(), dQA.data_ptr(), dQV.data_ptr(),
                                        dn.data_ptr(), n, 1, q2t.data_ptr(), t2q.data_ptr(), best.data_ptr(), second.data_ptr(), nm.data_ptr(),
                                        torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        gs = G["search"]["rule%d" % rule]
        got = (int(nm.item()), sha(q2t.cpu().numpy()), sha(t2q.cpu().numpy()), sha(best.cpu().numpy()), sha(second.cpu().numpy()))
        assert got == (gs["nmatches"], gs["q2t"], gs["t2q"], gs["best"], gs["second"]), rule
    gi, gm = capi.distinctive(I["desc"], I["seg_off"])
    assert [[int(a), int(c)] for a, c in zip(gi, gm)] == G["distinctive"]
