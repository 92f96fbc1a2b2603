#!/usr/bin/env python3
"""Randomised differential test of the rows either side of the path: bag-of-words transform, undistortion + grid, window queries,
greedy searches (every rule, incl. the key-frame pair searches over vocabulary nodes), distinctive descriptor — HIP vs the CPU oracle on random shapes / parameters.
usage: fuzz_frontend.py [cases] [seed]   — prints one JSON line."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import kf_device as kd
import kf_pairs
import oracle_lib as ol
from orb_slam_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
bad, done = [], {"bow": 0, "kf_search": 0, "frame": 0, "area": 0, "search": 0, "distinctive": 0}
t0 = time.time()
st = torch.cuda.current_stream().cuda_stream
T = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()


def same(a, b):
    return all(np.array_equal(x, y) for x, y in zip(a, b)) and a[1].tobytes() == b[1].tobytes()


for c in range(cases):
    # ---- bag of words
    k, L = int(rng.integers(2, 21)), int(rng.integers(1, 5))
    if k ** L > 20000:
        L = max(1, L - 1)
    scoring, weighting = int(rng.integers(0, 6)), int(rng.integers(0, 4))
    voc = synth.vocabulary(k, L, seed=int(rng.integers(1, 10**6)), ragged=bool(rng.random() < 0.5), order=str(rng.choice(["bfs", "kmeans"])),
                           stop_frac=float(rng.choice([0.0, 0.02, 0.3])))
    V = capi.ORBVocabulary.from_nodes(k, L, scoring, weighting, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    O = ol.OracleVocabulary(voc=voc, scoring=scoring, weighting=weighting)
    n = int(rng.choice([0, 1, 17, 300, 1000, 2047, 4096]))
    d = synth.descriptors(max(n, 1), int(rng.integers(1, 10**6)))[:n]
    if n > 10:
        d[n // 2:] = d[: n - n // 2]
    lu = int(rng.integers(0, L + 2))
    if not same(V.transform(d, lu), O.transform(d, lu)):
        bad.append(("bow", c, k, L, scoring, weighting, n, lu))
    done["bow"] += 1
    V.close()
    # ---- key-frame pair searches over the vocabulary nodes: SearchForTriangulation and SearchByBoW(KF, KF), two pairs per case
    if c % 2 == 0:
        capk = int(rng.choice([64, 500, 1000, 2000]))
        prs = [kf_pairs.pair(int(rng.integers(1, 10**6)), int(rng.integers(0, capk + 1)), int(rng.integers(0, capk + 1)), max_flips=int(rng.choice([4, 24, 60])),
                             line_noise=float(rng.choice([0.5, 2.0, 4.0])), p_mp1=float(rng.choice([0.0, 0.3, 0.9])), p_mp2=float(rng.choice([0.0, 0.3])))
               for _ in range(2)]
        lsup = int(rng.integers(0, L + 1))
        S = kd.setup(prs, capk, voc=voc, k=k, L=L, levelsup=lsup)
        chk = bool(rng.random() < 0.5); thl = int(rng.choice([50, 30, 80])); ratio = float(rng.choice([0.6, 0.75, 0.9]))
        sig2 = (kf_pairs.LEVEL_SIGMA2 * np.float32(rng.choice([1.0, 0.3, 4.0]))).astype(np.float32)
        g = kd.run_triangulation(S, capk, thl, chk, sig2)
        h = kd.run_bow_kf(S, capk, thl, ratio, chk)
        for i, p in enumerate(prs):
            fv1, fv2 = kd.host_fv(S["A"], i), kd.host_fv(S["B"], i)
            n1, n2 = S["n1"][i], S["n2"][i]
            pos = fv1[2].astype(np.int64)
            wt = ol.search_for_triangulation(thl, chk, p["F"], sig2, fv1, p["k1"], p["d1"], p["mp1"], fv2, p["k2"], p["d2"], p["mp2"])
            gt = (int(g[4][i]), kd.by_feature(pos, n1, g[0][i, :g[5][i]]), kd.inverse_by_feature(pos, g[1][i, :n2]), kd.by_feature(pos, n1, g[2][i, :g[5][i]]),
                  kd.by_feature(pos, n1, g[3][i, :g[5][i]]))
            if gt[0] != wt[0] or any(not np.array_equal(x, y) for x, y in zip(gt[1:], wt[1:])):
                bad.append(("triangulation", c, i, k, L, lsup, n1, n2, thl, chk))
            wb = ol.search_by_bow_kf(thl, ratio, chk, fv1, p["d1"], p["k1"]["angle"], h[4][i, :n1], fv2, p["d2"], p["k2"]["angle"], h[5][i, :n2])
            gb = (int(h[2][i]), kd.by_feature(pos, n1, h[0][i, :h[3][i]]), kd.inverse_by_feature(pos, h[1][i, :n2]))
            if gb[0] != wb[0] or any(not np.array_equal(x, y) for x, y in zip(gb[1:], wb[1:])):
                bad.append(("bow_kf", c, i, k, L, lsup, n1, n2, thl, ratio, chk))
            done["kf_matches"] = done.get("kf_matches", 0) + int(wt[0]) + int(wb[0])
        S["V"].close()
        done["kf_search"] += 2
    # ---- frame steps + searches
    w, h = int(rng.integers(200, 1300)), int(rng.integers(150, 1000))
    dist = [(0.2624, -0.9531, -0.0054, 0.0026), (-0.2834, 0.0739, 0.0002, 0.00002), (0.0, 0.0, 0.0, 0.0), (-0.1, 0.02, 0.001, -0.002, 0.003)][int(rng.integers(0, 4))]
    cam = capi.Camera.make(float(w * (0.7 + rng.random() * 0.3)), float(w * (0.7 + rng.random() * 0.3)), w / 2 + rng.normal(0, 5), h / 2 + rng.normal(0, 5), dist, w, h)
    try:
        b = capi.image_bounds(cam)
    except capi.OrbxError:
        done["degenerate_camera"] = done.get("degenerate_camera", 0) + 1      # the undistorted corners do not span a box (the reference divides by zero)
        continue
    if b.astuple()[:4] != ol.frame_bounds(cam, capi.Bounds).astuple()[:4]:
        bad.append(("bounds", c))
    nt = int(rng.choice([0, 1, 50, 500, 1000, 2000]))
    kp = np.zeros(nt, dtype=capi.KP_DTYPE)
    kp["x"] = (rng.random(nt) * (w + 30) - 15).astype(np.float32); kp["y"] = (rng.random(nt) * (h + 30) - 15).astype(np.float32)
    if nt > 100 and rng.random() < 0.5:
        kp["x"][: nt // 2] = w / 2 + rng.normal(0, 10, nt // 2); kp["y"][: nt // 2] = h / 2 + rng.normal(0, 10, nt // 2)
    kp["angle"] = (rng.random(nt) * 360).astype(np.float32); kp["octave"] = rng.integers(0, 8, nt)
    un, off, feat = capi.undistort_grid(cam, b, kp)
    wun = ol.frame_undistort(cam, kp); woff, wfeat = ol.frame_grid(b, wun)
    if un.tobytes() != wun.tobytes() or not np.array_equal(off, woff) or not np.array_equal(feat, wfeat):
        bad.append(("frame", c, w, h, nt))
    done["frame"] += 1
    nq = int(rng.choice([0, 1, 64, 300, 1000]))
    if nt and nq:
        src = rng.integers(0, nt, nq)
        rad = rng.choice([3.0, 10.0, 40.0, 150.0])
        qxyr = np.stack([wun["x"][src] + rng.normal(0, rad / 3, nq), wun["y"][src] + rng.normal(0, rad / 3, nq), np.full(nq, rad)], -1).astype(np.float32)
        lv = wun["octave"][src]
        lo, hi = int(rng.integers(-1, 2)), int(rng.integers(0, 2))
        qlev = np.stack([lv - lo, lv + hi], -1).astype(np.int32)
        if rng.random() < 0.2:
            qlev[:] = -1
        desc = synth.descriptors(nt, int(rng.integers(1, 10**6)))
        qdesc = desc[src].copy(); qdesc[np.arange(nq), rng.integers(0, 32, nq)] ^= np.uint8(1 << int(rng.integers(0, 8)))
        qangle = ((wun["angle"][src] + rng.normal(10, 8, nq)) % 360).astype(np.float32)
        qvalid = (rng.random(nq) < 0.9).astype(np.uint8); claimed = (rng.random(nt) < 0.2).astype(np.uint8)
        seg, cand = capi.features_in_area(b, wun, woff, wfeat, qxyr[:50], qlev[:50])
        for q in range(min(50, nq)):
            if not np.array_equal(cand[seg[q]:seg[q + 1]], ol.frame_features_in_area(b, wun, woff, wfeat, float(qxyr[q, 0]), float(qxyr[q, 1]), float(qxyr[q, 2]), int(qlev[q, 0]), int(qlev[q, 1]))):
                bad.append(("area", c, q)); break
        done["area"] += 1
        rule = int(rng.choice([0, 1, 2, 3, 5])); th = int(rng.choice([50, 100, 64])); ratio = float(rng.choice([0.6, 0.75, 0.9])); check = bool(rng.random() < 0.5)
        cap = max(nt, nq)
        dUn = T(np.concatenate([wun, np.zeros(cap - nt, capi.KP_DTYPE)]).view(np.uint8)); dD = T(np.concatenate([desc, np.zeros((cap - nt, 32), np.uint8)]))
        dOff, dFeat = T(woff), T(np.concatenate([wfeat, np.zeros(cap - len(wfeat), np.int32)]))
        pad = lambda a, m: np.concatenate([a, np.zeros((m - len(a),) + a.shape[1:], a.dtype)])
        dQX, dQL, dQD, dQA, dQV, dC = T(pad(qxyr, cap)), T(pad(qlev, cap)), T(pad(qdesc, cap)), T(pad(qangle, cap)), T(pad(qvalid, cap)), T(pad(claimed, cap))
        dnt, dnq = T(np.array([nt], np.int32)), T(np.array([nq], np.int32))
        o_q2t = torch.zeros(cap, dtype=torch.int32, device="cuda"); o_t2q = torch.zeros(cap, dtype=torch.int32, device="cuda")
        o_b = torch.zeros(cap, dtype=torch.int32, device="cuda"); o_s = torch.zeros(cap, dtype=torch.int32, device="cuda"); o_n = torch.zeros(1, dtype=torch.int32, device="cuda")
        capi.window_search_batch_device(b, rule, th, ratio, check, dUn.data_ptr(), dD.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), dnt.data_ptr(), cap,
                                        dC.data_ptr() if rule == 0 else 0, dQX.data_ptr(), dQL.data_ptr(), dQD.data_ptr(), dQA.data_ptr(), dQV.data_ptr(), dnq.data_ptr(), cap, 1,
                                        o_q2t.data_ptr(), o_t2q.data_ptr(), o_b.data_ptr(), o_s.data_ptr(), o_n.data_ptr(), st)
        # the same without the distance outputs: matches and count must not differ
        n_q2t = torch.zeros(cap, dtype=torch.int32, device="cuda"); n_t2q = torch.zeros(cap, dtype=torch.int32, device="cuda"); n_n = torch.zeros(1, dtype=torch.int32, device="cuda")
        capi.window_search_batch_device(b, rule, th, ratio, check, dUn.data_ptr(), dD.data_ptr(), dOff.data_ptr(), dFeat.data_ptr(), dnt.data_ptr(), cap,
                                        dC.data_ptr() if rule == 0 else 0, dQX.data_ptr(), dQL.data_ptr(), dQD.data_ptr(), dQA.data_ptr(), dQV.data_ptr(), dnq.data_ptr(), cap, 1,
                                        n_q2t.data_ptr(), n_t2q.data_ptr(), 0, 0, n_n.data_ptr(), st)
        torch.cuda.synchronize()
        if not (torch.equal(o_n, n_n) and torch.equal(o_q2t[:nq], n_q2t[:nq]) and torch.equal(o_t2q[:nt], n_t2q[:nt])):
            bad.append(("search without distance outputs", c, rule, th, ratio, check, nt, nq, float(rad)))
        wnt = ol.window_search(b, rule, th, ratio, check, wun, desc, woff, wfeat, claimed if rule == 0 else None, qxyr, qlev, qdesc, qangle, qvalid)
        got = (int(o_n.item()), o_q2t.cpu().numpy()[:nq], o_t2q.cpu().numpy()[:nt], o_b.cpu().numpy()[:nq], o_s.cpu().numpy()[:nq])
        if got[0] != wnt[0] or any(not np.array_equal(x, y) for x, y in zip(got[1:], wnt[1:])):
            bad.append(("search", c, rule, th, ratio, check, nt, nq, float(rad)))
        done["search"] += 1
    # ---- distinctive descriptor
    sizes = rng.integers(0, 70, int(rng.integers(1, 30)))
    segs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    dd = synth.descriptors(max(int(segs[-1]), 1), int(rng.integers(1, 10**6)))[: segs[-1]]
    if len(dd) > 8:
        dd[::3] = dd[0]
    gi, gm = capi.distinctive(dd, segs)
    for p in range(len(sizes)):
        if (int(gi[p]), int(gm[p])) != ol.distinctive(dd[segs[p]:segs[p + 1]]):
            bad.append(("distinctive", c, p)); break
    done["distinctive"] += 1
print(json.dumps({"cases": cases, "seed": seed, "checked": done, "mismatches": bad, "seconds": round(time.time() - t0, 1), "build": capi.build_id()}))
