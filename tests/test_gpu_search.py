"""GPU parity of the greedy grid-window searches (include/orbs.h) against oracle/search_oracle.cpp: every query's final
match, both inverse maps, the two scan distances and the return value, for all four accept rules."""
import numpy as np
import pytest

import oracle_lib as ol
from orb_slam_amd import capi, synth

pytestmark = pytest.mark.gpu

CAM = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), 640, 480)


@pytest.fixture(params=["bucketed", "plain", "bucketed-wide", "plain-wide"])
def index_form(request):
    """The grid searches come in two index forms with the same results (include/orbs.h): the level-bucketed index (frames up to ~2400
    features) and the plain 64 x 48 CSR scan (larger frames, or ORBS_BUCKETS=0) — and, since round 6, in two launch shapes: 256 threads per
    problem (launches with many problems) and 1024 (few problems: the one-problem calls of ORB_SLAM::ORBmatcher).  The same problems go through all."""
    capi.set_search_buckets(1 if request.param.startswith("bucketed") else 0)
    capi.set_search_wide_max(1 << 30 if request.param.endswith("wide") else 0)
    yield request.param
    capi.set_search_buckets(-1)
    capi.set_search_wide_max(-2)


def _problem(seed, nt, nq, radius, crowd=False, level_mode="pm1"):
    """a train frame + ordered queries that compete for its features: queries are noisy copies of train features (several
    per feature when crowd), so later queries meet claimed candidates"""
    rng = np.random.default_rng(seed)
    k = np.zeros(nt, dtype=capi.KP_DTYPE)
    if crowd and nt:
        cx, cy = rng.random(40) * 600 + 20, rng.random(40) * 440 + 20
        c = rng.integers(0, 40, nt)
        k["x"] = (cx[c] + rng.normal(0, 6, nt)).astype(np.float32)
        k["y"] = (cy[c] + rng.normal(0, 6, nt)).astype(np.float32)
    else:
        k["x"] = (rng.random(nt) * 640).astype(np.float32)
        k["y"] = (rng.random(nt) * 480).astype(np.float32)
    k["angle"] = (rng.random(nt) * 360).astype(np.float32)
    k["octave"] = rng.integers(0, 8, nt)
    k["size"], k["class_id"] = 31, -1
    desc = synth.descriptors(nt, seed + 1000)
    if crowd and nt:
        desc[rng.integers(0, nt, nt // 3)] = desc[0]                     # duplicate descriptors: distance ties inside windows
    src = rng.integers(0, max(nt, 1), nq) if nt else np.zeros(nq, np.int64)
    qxyr = np.zeros((nq, 3), np.float32)
    qlev = np.zeros((nq, 2), np.int32)
    qdesc = synth.descriptors(nq, seed + 2000)
    qangle = (rng.random(nq) * 360).astype(np.float32)
    lv = k["octave"][src] if nt else rng.integers(0, 8, nq)
    rad = np.full(nq, radius, np.float32) if np.isscalar(radius) else np.asarray(radius, np.float32)[lv]
    if nt:
        qxyr[:, 0] = k["x"][src] + rng.normal(0, 1, nq) * rad / 3
        qxyr[:, 1] = k["y"][src] + rng.normal(0, 1, nq) * rad / 3
        flips = rng.integers(0, 256, (nq, 6))
        qd = desc[src].copy()
        for j in range(6):
            qd[np.arange(nq), flips[:, j] // 8] ^= (1 << (flips[:, j] % 8)).astype(np.uint8)
        keep = rng.random(nq) < 0.85
        qdesc[keep] = qd[keep]
        near = rng.random(nq) < 0.7
        qangle[near] = (k["angle"][src][near] + rng.normal(12, 8, int(near.sum()))).astype(np.float32) % np.float32(360)
    qxyr[:, 2] = rad
    if level_mode == "pm1":
        qlev[:, 0], qlev[:, 1] = lv - 1, lv + 1
    elif level_mode == "same":
        qlev[:, 0], qlev[:, 1] = lv, lv
    elif level_mode == "below":
        qlev[:, 0], qlev[:, 1] = lv - 1, lv
    else:
        qlev[:] = -1
    qvalid = (rng.random(nq) < 0.9).astype(np.uint8)
    claimed = (rng.random(nt) < 0.2).astype(np.uint8)
    return dict(kps=k, desc=desc, qxyr=qxyr, qlev=qlev, qdesc=qdesc, qangle=qangle, qvalid=qvalid, claimed=claimed)


def _run_batch(problems, rule, th, ratio, check, use_claimed, use_valid, cap, qcap):
    torch = pytest.importorskip("torch")
    P = len(problems)
    b = capi.image_bounds(CAM)
    K = np.zeros((P, cap), dtype=capi.KP_DTYPE)
    D = np.zeros((P, cap, 32), np.uint8)
    C = np.zeros((P, cap), np.uint8)
    QX = np.zeros((P, qcap, 3), np.float32); QL = np.zeros((P, qcap, 2), np.int32); QD = np.zeros((P, qcap, 32), np.uint8)
    QA = np.zeros((P, qcap), np.float32); QV = np.zeros((P, qcap), np.uint8)
    nt = np.array([len(p["kps"]) for p in problems], np.int32)
    nq = np.array([len(p["qxyr"]) for p in problems], np.int32)
    for i, p in enumerate(problems):
        K[i, :nt[i]] = p["kps"]; D[i, :nt[i]] = p["desc"]; C[i, :nt[i]] = p["claimed"]
        QX[i, :nq[i]] = p["qxyr"]; QL[i, :nq[i]] = p["qlev"]; QD[i, :nq[i]] = p["qdesc"]; QA[i, :nq[i]] = p["qangle"]; QV[i, :nq[i]] = p["qvalid"]
    st = torch.cuda.current_stream().cuda_stream
    dK = torch.from_numpy(K.view(np.uint8).reshape(P, cap, 28)).cuda()
    dnt, dnq = torch.from_numpy(nt).cuda(), torch.from_numpy(nq).cuda()
    dUn = torch.zeros((P, cap, 28), dtype=torch.uint8, device="cuda")
    dOff = torch.zeros((P, capi.GRID_CELLS + 1), dtype=torch.int32, device="cuda")
    dFeat = torch.zeros((P, cap), dtype=torch.int32, device="cuda")
    capi.undistort_grid_batch_device(CAM, b, dK.data_ptr(), dnt.data_ptr(), P, cap, dUn.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), st)
    dD, dC = torch.from_numpy(D).cuda(), torch.from_numpy(C).cuda()
    dQX, dQL, dQD = torch.from_numpy(QX).cuda(), torch.from_numpy(QL).cuda(), torch.from_numpy(QD).cuda()
    dQA, dQV = torch.from_numpy(QA).cuda(), torch.from_numpy(QV).cuda()
    q2t = torch.full((P, qcap), -9, dtype=torch.int32, device="cuda"); t2q = torch.full((P, cap), -9, dtype=torch.int32, device="cuda")
    best = torch.full((P, qcap), -9, dtype=torch.int32, device="cuda"); second = torch.full((P, qcap), -9, dtype=torch.int32, device="cuda")
    nm = torch.full((P,), -9, dtype=torch.int32, device="cuda")
    capi.window_search_batch_device(b, rule, th, ratio, check, dUn.data_ptr(), dD.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), dnt.data_ptr(), cap,
                                    dC.data_ptr() if use_claimed else 0, dQX.data_ptr(), dQL.data_ptr(), dQD.data_ptr(), dQA.data_ptr(),
                                    dQV.data_ptr() if use_valid else 0, dnq.data_ptr(), qcap, P, q2t.data_ptr(), t2q.data_ptr(), best.data_ptr(),
                                    second.data_ptr(), nm.data_ptr(), st)
    # the same launch WITHOUT the two distance outputs (what ORB_SLAM::ORBmatcher asks for): matches and counts must not depend on
    # which outputs are asked for
    q2t_n = torch.full((P, qcap), -9, dtype=torch.int32, device="cuda"); t2q_n = torch.full((P, cap), -9, dtype=torch.int32, device="cuda")
    nm_n = torch.full((P,), -9, dtype=torch.int32, device="cuda")
    capi.window_search_batch_device(b, rule, th, ratio, check, dUn.data_ptr(), dD.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), dnt.data_ptr(), cap,
                                    dC.data_ptr() if use_claimed else 0, dQX.data_ptr(), dQL.data_ptr(), dQD.data_ptr(), dQA.data_ptr(),
                                    dQV.data_ptr() if use_valid else 0, dnq.data_ptr(), qcap, P, q2t_n.data_ptr(), t2q_n.data_ptr(), 0, 0, nm_n.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(nm, nm_n) and torch.equal(q2t, q2t_n) and torch.equal(t2q, t2q_n), "results depend on whether best / second are asked for"
    un = dUn.cpu().numpy().reshape(P, cap * 28).view(capi.KP_DTYPE).reshape(P, cap)
    off, feat = dOff.cpu().numpy(), dFeat.cpu().numpy()
    q2t, t2q, best, second, nm = (t.cpu().numpy() for t in (q2t, t2q, best, second, nm))
    total = 0
    for i, p in enumerate(problems):
        want = ol.window_search(b, rule, th, ratio, check, un[i, :nt[i]], p["desc"], off[i], feat[i, :max(off[i][-1], 1)],
                                p["claimed"] if use_claimed else None, p["qxyr"], p["qlev"], p["qdesc"], p["qangle"],
                                p["qvalid"] if use_valid else None)
        assert nm[i] == want[0], (i, nm[i], want[0])
        np.testing.assert_array_equal(q2t[i, :nq[i]], want[1], err_msg="q2t problem %d" % i)
        np.testing.assert_array_equal(t2q[i, :nt[i]], want[2], err_msg="t2q problem %d" % i)
        np.testing.assert_array_equal(best[i, :nq[i]], want[3], err_msg="best problem %d" % i)
        np.testing.assert_array_equal(second[i, :nq[i]], want[4], err_msg="second problem %d" % i)
        total += want[0]
    return total


SCALE = np.float32(1.2) ** np.arange(8, dtype=np.float32)


@pytest.mark.parametrize("rule,th,ratio,check,level_mode,radius", [
    (capi.RULE_MAPPOINTS, capi.TH_HIGH, 0.8, False, "below", 4.0 * SCALE),      # Tracking.cc:724 SearchByProjection(F, localMapPoints, th)
    (capi.RULE_MAPPOINTS, capi.TH_HIGH, 0.8, False, "below", 20.0 * SCALE),
    (capi.RULE_WINDOW, capi.TH_HIGH, 0.8, True, "same", 100.0),                 # Tracking.cc:502 WindowSearch(last, current, 100, ...)
    (capi.RULE_WINDOW, capi.TH_HIGH, 0.8, True, "same", 200.0),                 # Tracking.cc:497
    (capi.RULE_WINDOW, capi.TH_HIGH, 0.9, False, "same", 15.0),                 # Tracking.cc:528 SearchByProjection(last, current, 15, ...)
    (capi.RULE_BEST, capi.TH_HIGH, 0.9, True, "pm1", 15.0 * SCALE),             # Tracking.cc:565 SearchByProjection(current, last, 15)
    (capi.RULE_BEST, capi.TH_HIGH, 0.9, False, "all", 30.0),
    (capi.RULE_INIT, capi.TH_LOW, 0.9, True, "same", 100.0),                    # Tracking.cc:353 SearchForInitialization(..., 100)
    (capi.RULE_INIT, capi.TH_HIGH, 0.9, False, "all", 40.0),
    (capi.RULE_FREE, capi.TH_LOW, 0.9, False, "below", 3.0 * SCALE),            # LocalMapping.cc Fuse(pKF, mapPoints, th = 3)
    (capi.RULE_FREE, capi.TH_HIGH, 0.9, True, "below", 10.0 * SCALE),           # LoopClosing.cc:370 SearchByProjection(pKF, Scw, points, matched, 10)
], ids=["mappoints_r4", "mappoints_r20", "window100_rot", "window200_rot", "window15", "best15_rot", "best30_all", "init100_rot", "init40_all",
        "free_fuse3", "free_sim10"])
def test_rules_against_oracle(rule, th, ratio, check, level_mode, radius, index_form):
    cap = qcap = 1000
    problems = [_problem(10 + i, nt, nq, radius, crowd=(i % 2 == 1), level_mode=level_mode)
                for i, (nt, nq) in enumerate([(1000, 1000), (1000, 1000), (700, 1000), (1000, 300), (1, 5), (0, 7), (5, 0), (64, 65), (999, 513)])]
    if rule == capi.RULE_INIT:
        for p in problems:                       # SearchForInitialization only takes level-0 queries (src/ORBmatcher.cc:613-614)
            p["qlev"][:] = 0 if level_mode == "same" else -1
            p["kps"]["octave"][: len(p["kps"]) * 3 // 4] = 0
    total = _run_batch(problems, rule, th, ratio, check, use_claimed=(rule == capi.RULE_MAPPOINTS), use_valid=True, cap=cap, qcap=qcap)
    assert total > 200                           # the scenario really produces matches (and therefore contention)


def test_contention_changes_results():
    """sanity of the scenario itself: with claims honoured the outcome differs from an order-free top-2 (otherwise the
    sequential masking would be untested)"""
    p = _problem(77, 1000, 1000, 30.0, crowd=True, level_mode="all")
    b = capi.image_bounds(CAM)
    un = ol.frame_undistort(CAM, p["kps"])
    off, feat = ol.frame_grid(b, un)
    n1, q2t, _, best, _ = ol.window_search(b, capi.RULE_BEST, 100, 0.9, False, un, p["desc"], off, feat, None, p["qxyr"], p["qlev"], p["qdesc"], None, None)
    free = 0
    for q in range(1000):
        c = ol.frame_features_in_area(b, un, off, feat, *[float(v) for v in p["qxyr"][q]], -1, -1)
        if len(c):
            d = [int(np.unpackbits(p["qdesc"][q] ^ p["desc"][i]).sum()) for i in c]
            free += int(best[q] != min(d))
    assert free > 20


def test_large_frames_and_capacity(index_form):
    radius = 25.0
    problems = [_problem(200 + i, 2000, 2000, radius, crowd=(i == 1), level_mode="pm1") for i in range(2)]
    _run_batch(problems, capi.RULE_WINDOW, 100, 0.8, True, False, False, cap=2000, qcap=2000)
    # beyond ~2850 train features the descriptors stay in global memory: a 4000-feature frame (the initialisation extractor of an
    # nFeatures = 2000 setup) is searched exactly as well, under every kind of rule
    big = [_problem(300 + i, 4000, 3000, 20.0, crowd=(i == 1), level_mode="pm1") for i in range(2)]
    _run_batch(big, capi.RULE_WINDOW, 100, 0.8, True, False, False, cap=4000, qcap=3000)
    _run_batch(big, capi.RULE_INIT, 50, 0.9, True, False, False, cap=4000, qcap=3000)
    assert max(capi.lib().orbs_lds_bytes(2000, 2000), capi.lib().orbs_lds_bytes(4000, 3000)) < 160 * 1024 < capi.lib().orbs_lds_bytes(8192, 8192)
    torch = pytest.importorskip("torch")
    z = torch.zeros(64, dtype=torch.int32, device="cuda")
    with pytest.raises(capi.OrbxError) as e:
        capi.window_search_batch_device(capi.image_bounds(CAM), capi.RULE_BEST, 100, 0.9, False, z.data_ptr(), z.data_ptr(), z.data_ptr(), z.data_ptr(),
                                        z.data_ptr(), 8192, 0, z.data_ptr(), z.data_ptr(), z.data_ptr(), 0, 0, z.data_ptr(), 8192, 1, z.data_ptr(),
                                        z.data_ptr(), 0, 0, z.data_ptr())
    assert e.value.code == capi.ORBX_ERR_CAPACITY


@pytest.mark.parametrize("check", [True, False], ids=["rot", "norot"])
def test_search_by_bow_pipeline(check):
    """ORBmatcher::SearchByBoW(KeyFrame*, Frame&, ...) end to end on the device: both FeatureVectors from the BoW transform,
    the merge walk as per-query list ranges, then the in-order list search with the TH_LOW / strict-ratio rule."""
    torch = pytest.importorskip("torch")
    P, cap = 5, 1000
    voc = synth.vocabulary(10, 4, seed=6)
    V = capi.ORBVocabulary.from_nodes(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    OV = ol.OracleVocabulary(voc=voc)
    rng = np.random.default_rng(9)
    nK = np.array([1000, 1000, 600, 1, 0], np.int32)
    nF = np.array([1000, 800, 1000, 1000, 10], np.int32)
    KD = np.stack([synth.descriptors(cap, 700 + i) for i in range(P)])
    FD = np.stack([synth.descriptors(cap, 800 + i) for i in range(P)])
    KA = (rng.random((P, cap)) * 360).astype(np.float32)
    FA = (rng.random((P, cap)) * 360).astype(np.float32)
    for i in range(P):                           # frame features = noisy copies of key-frame features (several per feature → contention)
        src = rng.integers(0, max(nK[i], 1), cap)
        keep = rng.random(cap) < 0.8
        noisy = KD[i, src].copy()
        bits = rng.integers(0, 256, (cap, 3))
        for j in range(3):
            noisy[np.arange(cap), bits[:, j] // 8] ^= (1 << (bits[:, j] % 8)).astype(np.uint8)
        if nK[i]:
            FD[i, keep] = noisy[keep]
            FA[i, keep] = ((KA[i, src] + rng.normal(10, 6, cap).astype(np.float32)) % np.float32(360))[keep]
    KV = (rng.random((P, cap)) < 0.85).astype(np.uint8)
    st = torch.cuda.current_stream().cuda_stream
    i32, u8, f64 = torch.int32, torch.uint8, torch.float64

    def transform(D, n):
        dD, dn = torch.from_numpy(D).cuda(), torch.from_numpy(n).cuda()
        o = dict(bid=torch.zeros((P, cap), dtype=i32, device="cuda"), bval=torch.zeros((P, cap), dtype=f64, device="cuda"),
                 node=torch.zeros((P, cap), dtype=i32, device="cuda"), off=torch.zeros((P, cap + 1), dtype=i32, device="cuda"),
                 feat=torch.zeros((P, cap), dtype=i32, device="cuda"), cnt=torch.zeros((2, P), dtype=i32, device="cuda"), D=dD, n=dn)
        V.transform_batch_device(dD.data_ptr(), dn.data_ptr(), P, cap, 2, o["bid"].data_ptr(), o["bval"].data_ptr(), o["cnt"][0].data_ptr(),
                                 o["node"].data_ptr(), o["off"].data_ptr(), o["feat"].data_ptr(), o["cnt"][1].data_ptr(), st)
        return o

    K, F = transform(KD, nK), transform(FD, nF)
    qrange = torch.zeros((P, cap, 2), dtype=i32, device="cuda")
    nq = torch.zeros(P, dtype=i32, device="cuda")
    capi.bow_ranges_batch_device(K["node"].data_ptr(), K["off"].data_ptr(), K["cnt"][1].data_ptr(), F["node"].data_ptr(), F["off"].data_ptr(),
                                 F["cnt"][1].data_ptr(), cap, P, qrange.data_ptr(), nq.data_ptr(), st)
    # the frame's list length = number of its features with a (non-stopped) word
    torch.cuda.synchronize()
    f_off = F["off"].cpu().numpy(); f_cnt = F["cnt"].cpu().numpy()
    nlist = torch.from_numpy(np.array([f_off[i, f_cnt[1, i]] for i in range(P)], np.int32)).cuda()
    fkps = np.zeros((P, cap), dtype=capi.KP_DTYPE)
    fkps["angle"] = FA
    dFK = torch.from_numpy(fkps.view(np.uint8).reshape(P, cap, 28)).cuda()
    dKA, dKV = torch.from_numpy(KA).cuda(), torch.from_numpy(KV).cuda()
    q2t = torch.full((P, cap), -9, dtype=i32, device="cuda"); t2q = torch.full((P, cap), -9, dtype=i32, device="cuda")
    best = torch.full((P, cap), -9, dtype=i32, device="cuda"); second = torch.full((P, cap), -9, dtype=i32, device="cuda")
    nm = torch.zeros(P, dtype=i32, device="cuda")
    capi.list_search_batch_device(capi.RULE_BOW, capi.TH_LOW, 0.75, check, dFK.data_ptr(), F["D"].data_ptr(), F["feat"].data_ptr(), nlist.data_ptr(),
                                  F["n"].data_ptr(), cap, 0, qrange.data_ptr(), K["feat"].data_ptr(), K["D"].data_ptr(), dKA.data_ptr(), dKV.data_ptr(),
                                  nq.data_ptr(), cap, P, q2t.data_ptr(), t2q.data_ptr(), best.data_ptr(), second.data_ptr(), nm.data_ptr(), st)
    q2t_n = torch.full((P, cap), -9, dtype=i32, device="cuda"); t2q_n = torch.full((P, cap), -9, dtype=i32, device="cuda"); nm_n = torch.zeros(P, dtype=i32, device="cuda")
    capi.list_search_batch_device(capi.RULE_BOW, capi.TH_LOW, 0.75, check, dFK.data_ptr(), F["D"].data_ptr(), F["feat"].data_ptr(), nlist.data_ptr(),
                                  F["n"].data_ptr(), cap, 0, qrange.data_ptr(), K["feat"].data_ptr(), K["D"].data_ptr(), dKA.data_ptr(), dKV.data_ptr(),
                                  nq.data_ptr(), cap, P, q2t_n.data_ptr(), t2q_n.data_ptr(), 0, 0, nm_n.data_ptr(), st)      # without the distance outputs
    torch.cuda.synchronize()
    assert torch.equal(nm, nm_n) and torch.equal(q2t, q2t_n) and torch.equal(t2q, t2q_n), "results depend on whether best / second are asked for"
    k_node, k_off, k_feat, k_cnt = (K[x].cpu().numpy() for x in ("node", "off", "feat", "cnt"))
    f_node, f_feat = F["node"].cpu().numpy(), F["feat"].cpu().numpy()
    q2t, t2q, best, second, nm, nqh = (x.cpu().numpy() for x in (q2t, t2q, best, second, nm, nq))
    total = 0
    for i in range(P):
        kfv = (k_node[i, :k_cnt[1, i]].view(np.uint32), k_off[i, :k_cnt[1, i] + 1], k_feat[i, :k_off[i, k_cnt[1, i]]].view(np.uint32))
        ffv = (f_node[i, :f_cnt[1, i]].view(np.uint32), f_off[i, :f_cnt[1, i] + 1], f_feat[i, :f_off[i, f_cnt[1, i]]].view(np.uint32))
        # the device FeatureVectors are the oracle's (already covered by test_gpu_bow; re-checked here on the way)
        want_fv = OV.transform(KD[i, :nK[i]], 2)
        assert np.array_equal(kfv[0], want_fv[2]) and np.array_equal(kfv[1], want_fv[3]) and np.array_equal(kfv[2], want_fv[4])
        w = ol.search_by_bow(capi.TH_LOW, 0.75, check, kfv, KD[i, :nK[i]], KA[i, :nK[i]], KV[i, :nK[i]], ffv, FD[i, :nF[i]], FA[i, :nF[i]])
        assert nm[i] == w[0], (i, nm[i], w[0])
        nqi = nqh[i]
        assert nqi == len(kfv[2])
        # device results are per query POSITION (key-frame FeatureVector order); map to key-frame feature indices
        pos_feat = kfv[2].astype(np.int64)
        got_q2t = np.full(nK[i], -1, np.int32); got_best = np.full(nK[i], -1, np.int32); got_second = np.full(nK[i], -1, np.int32)
        got_q2t[pos_feat] = q2t[i, :nqi]; got_best[pos_feat] = best[i, :nqi]; got_second[pos_feat] = second[i, :nqi]
        np.testing.assert_array_equal(got_q2t, w[1])
        got_t2q = t2q[i, :nF[i]].copy()
        m = got_t2q >= 0
        got_t2q[m] = pos_feat[got_t2q[m]]
        np.testing.assert_array_equal(got_t2q, w[2])
        np.testing.assert_array_equal(got_best, w[3])
        np.testing.assert_array_equal(got_second, w[4])
        total += w[0]
    assert total > 250


# ---- round 6 (VERDICT r05 missing #4): a stream with real inter-frame correspondence.  S-warp frames (orb_slam_amd/csrc/synth_frames.c: one base
# texture per 64-frame sequence seen through a slowly panning / rolling / zooming camera + noise) give the searches what a camera gives them: most
# queries HAVE a true partner a pixel or two away, distances are low and tie often, the rotation histogram has one dominant bin and a tail.
def _warp_pairs(first, n, nfeat=1000):
    o = ol.OracleExtractor(nfeatures=nfeat)
    frames = synth.frames(640, 480, synth.WARP, first, n + 1)
    ext = [o(f) for f in frames]
    return [(ext[i], ext[i + 1]) for i in range(n)]


@pytest.mark.parametrize("rule,th,ratio,check,level_mode,radius,floor", [
    (capi.RULE_WINDOW, capi.TH_HIGH, 0.8, True, "same", 15.0, 0.60),           # WindowSearch between consecutive frames, rotation check
    (capi.RULE_WINDOW, capi.TH_HIGH, 0.6, True, "same", 100.0, 0.55),          # Tracking.cc:502's window with the strict ratio
    (capi.RULE_BEST, capi.TH_HIGH, 0.9, True, "pm1", 15.0 * SCALE, 0.65),      # SearchByProjection(current, last, 15): best only, rotation check
    (capi.RULE_INIT, capi.TH_LOW, 0.9, True, "same", 100.0, 0.40),             # SearchForInitialization: level-0 queries only
], ids=["window15_rot", "window100_ratio06_rot", "best15_rot", "init100_rot"])
def test_searches_on_a_correlated_stream(rule, th, ratio, check, level_mode, radius, floor, index_form):
    problems = []
    for (k1, d1), (k2, d2) in _warp_pairs(64 * 5 + 3, 6) + _warp_pairs(64 * 9 + 50, 3):       # early in a sequence and deep into its camera path
        nq = len(k1)
        lv = k1["octave"]
        rad = np.full(nq, radius, np.float32) if np.isscalar(radius) else np.asarray(radius, np.float32)[lv]
        qlev = np.stack([lv - 1, lv + 1], -1) if level_mode == "pm1" else np.stack([lv, lv], -1)
        qvalid = np.ones(nq, np.uint8) if rule != capi.RULE_INIT else (lv == 0).astype(np.uint8)
        problems.append(dict(kps=k2, desc=d2, claimed=np.zeros(len(k2), np.uint8), qxyr=np.stack([k1["x"], k1["y"], rad], -1).astype(np.float32),
                             qlev=qlev.astype(np.int32), qdesc=d1, qangle=np.ascontiguousarray(k1["angle"]), qvalid=qvalid))
    total = _run_batch(problems, rule, th, ratio, check, use_claimed=False, use_valid=True, cap=1000, qcap=1000)
    nq_total = sum(int(p["qvalid"].sum()) for p in problems)
    assert total > floor * nq_total, (total, nq_total)     # camera-like match densities (S-blocks pairs: a few per thousand)
