#!/usr/bin/env python3
"""Timing of the KeyFrame-to-KeyFrame searches on the device, stage by stage: BoW transform of both key frames, the merge walk
(orbs_bow_ranges), then SearchForTriangulation (epipolar test in the scan) and SearchByBoW(KF, KF) over the FeatureVector
lists.  Key-frame pairs from tests/kf_pairs.py (1000 features each, vocabulary 10^4 with levelsup 2 = 100 nodes, the node
count of the reference's 10^6 vocabulary at levelsup 4)."""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import kf_pairs
from orb_slam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--problems", type=int, default=512); ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--distinct", type=int, default=32); ap.add_argument("--iters", type=int, default=10)
a = ap.parse_args()
P, n = a.problems, a.n
voc = synth.vocabulary(10, 4, seed=6)
V = capi.ORBVocabulary.from_nodes(10, 4, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
base = [kf_pairs.pair(300 + i, n, n) for i in range(a.distinct)]
pick = [base[i % a.distinct] for i in range(P)]
D1 = np.stack([p["d1"] for p in pick]); D2 = np.stack([p["d2"] for p in pick])
K1 = np.stack([p["k1"] for p in pick]); K2 = np.stack([p["k2"] for p in pick])
M1 = np.stack([p["mp1"] for p in pick]); M2 = np.stack([p["mp2"] for p in pick])
Fm = np.stack([p["F"].reshape(9) for p in pick])
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).cuda()
i32, f64 = torch.int32, torch.float64
st = torch.cuda.current_stream().cuda_stream
dn = torch.full((P,), n, dtype=i32, device="cuda")

def fv_buffers(D):
    return dict(D=t(D), bid=torch.zeros((P, n), dtype=i32, device="cuda"), bval=torch.zeros((P, n), dtype=f64, device="cuda"),
                node=torch.zeros((P, n), dtype=i32, device="cuda"), off=torch.zeros((P, n + 1), dtype=i32, device="cuda"),
                feat=torch.zeros((P, n), dtype=i32, device="cuda"), cnt=torch.zeros((2, P), dtype=i32, device="cuda"))
A, B = fv_buffers(D1), fv_buffers(D2)
def transform(o):
    V.transform_batch_device(o["D"].data_ptr(), dn.data_ptr(), P, n, 2, o["bid"].data_ptr(), o["bval"].data_ptr(), o["cnt"][0].data_ptr(),
                             o["node"].data_ptr(), o["off"].data_ptr(), o["feat"].data_ptr(), o["cnt"][1].data_ptr(), st)
qrange = torch.zeros((P, n, 2), dtype=i32, device="cuda"); nq = torch.zeros(P, dtype=i32, device="cuda")
def ranges():
    capi.bow_ranges_batch_device(A["node"].data_ptr(), A["off"].data_ptr(), A["cnt"][1].data_ptr(), B["node"].data_ptr(), B["off"].data_ptr(),
                                 B["cnt"][1].data_ptr(), n, P, qrange.data_ptr(), nq.data_ptr(), st)
transform(A); transform(B); ranges(); torch.cuda.synchronize()
b_off, b_cnt = B["off"].cpu().numpy(), B["cnt"].cpu().numpy()
nlist = t(np.array([b_off[i, b_cnt[1, i]] for i in range(P)], np.int32))
dK1, dK2 = t(K1.view(np.uint8).reshape(P, n, 28)), t(K2.view(np.uint8).reshape(P, n, 28))
dMP2, dQV, dF = t(M2), t((1 - M1).astype(np.uint8)), t(Fm)
dV1, dC2, dA1 = t((1 - M1).astype(np.uint8)), t(M2), t(K1["angle"])
q2t, t2q, best, sec = (torch.zeros((P, n), dtype=i32, device="cuda") for _ in range(4))
nm = torch.zeros(P, dtype=i32, device="cuda")
def tri():
    capi.triangulation_search_batch_device(capi.TH_LOW, True, dF.data_ptr(), kf_pairs.LEVEL_SIGMA2, dK2.data_ptr(), B["D"].data_ptr(), B["feat"].data_ptr(),
                                           nlist.data_ptr(), dn.data_ptr(), n, dMP2.data_ptr(), qrange.data_ptr(), A["feat"].data_ptr(), dK1.data_ptr(),
                                           A["D"].data_ptr(), dQV.data_ptr(), nq.data_ptr(), n, P, q2t.data_ptr(), t2q.data_ptr(), best.data_ptr(),
                                           sec.data_ptr(), nm.data_ptr(), st)
def bow_kf():
    capi.list_search_batch_device(capi.RULE_BOW, capi.TH_LOW - 1, 0.6, True, dK2.data_ptr(), B["D"].data_ptr(), B["feat"].data_ptr(), nlist.data_ptr(),
                                  dn.data_ptr(), n, dC2.data_ptr(), qrange.data_ptr(), A["feat"].data_ptr(), A["D"].data_ptr(), dA1.data_ptr(), dV1.data_ptr(),
                                  nq.data_ptr(), n, P, q2t.data_ptr(), t2q.data_ptr(), best.data_ptr(), sec.data_ptr(), nm.data_ptr(), st)
def timed(f):
    for _ in range(2): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.iters): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.iters
out = {"problems": P, "n": n}
out["transform_ms_per_frame_batch"] = round(timed(lambda: transform(A)), 4)
out["ranges_ms"] = round(timed(ranges), 4)
out["triangulation_ms"] = round(timed(tri), 4); out["triangulation_mean_matches"] = round(float(nm.float().mean()), 1)
out["bow_kf_ms"] = round(timed(bow_kf), 4); out["bow_kf_mean_matches"] = round(float(nm.float().mean()), 1)
out["triangulation_pairs_per_s"] = round(P / out["triangulation_ms"] * 1e3)
out["triangulation_pipeline_pairs_per_s"] = round(P / (2 * out["transform_ms_per_frame_batch"] + out["ranges_ms"] + out["triangulation_ms"]) * 1e3)
# the same search on one host core through the CPU oracle (test infrastructure), a bounded sample of pairs, results compared
import time
import oracle_lib as ol
tri(); torch.cuda.synchronize()
g_q2t, g_nm, g_nq = q2t.cpu().numpy(), nm.cpu().numpy(), nq.cpu().numpy()
def host_fv(o, i):
    node, off, feat, cnt = (o[x].cpu().numpy() for x in ("node", "off", "feat", "cnt"))
    nn = cnt[1, i]
    return node[i, :nn].view(np.uint32), off[i, :nn + 1], feat[i, :off[i, nn]].view(np.uint32)
ns, ok, t_cpu = min(a.distinct, 16), True, 0.0
for i in range(ns):
    p = pick[i]; f1, f2 = host_fv(A, i), host_fv(B, i)
    t0 = time.time()
    w = ol.search_for_triangulation(capi.TH_LOW, True, p["F"], kf_pairs.LEVEL_SIGMA2, f1, p["k1"], p["d1"], p["mp1"], f2, p["k2"], p["d2"], p["mp2"])
    t_cpu += time.time() - t0
    got = np.full(n, -1, np.int32); got[f1[2].astype(np.int64)] = g_q2t[i, :g_nq[i]]
    ok = ok and g_nm[i] == w[0] and np.array_equal(got, w[1])
out["cpu_oracle"] = {"pairs_per_s": round(ns / t_cpu, 1), "cores": 1, "sample": "%d pairs, search only (FeatureVectors given)" % ns, "matches_device": bool(ok)}
print(json.dumps(out))
