"""Device side of the KeyFrame-pair search tests (shared by tests/test_gpu_search_kf.py and tools/fuzz_frontend.py): batch the
pairs, run the BoW transform of both key frames and the merge-walk ranges on the GPU.  Test infrastructure."""
import numpy as np

from orb_slam_amd import capi, synth


def device_feature_vectors(torch, V, D, n, P, cap, levelsup, st):
    i32, f64 = torch.int32, torch.float64
    dD, dn = torch.from_numpy(D).cuda(), torch.from_numpy(n).cuda()
    o = dict(bid=torch.zeros((P, cap), dtype=i32, device="cuda"), bval=torch.zeros((P, cap), dtype=f64, device="cuda"),
             node=torch.zeros((P, cap), dtype=i32, device="cuda"), off=torch.zeros((P, cap + 1), dtype=i32, device="cuda"),
             feat=torch.zeros((P, cap), dtype=i32, device="cuda"), cnt=torch.zeros((2, P), dtype=i32, device="cuda"), D=dD, n=dn)
    V.transform_batch_device(dD.data_ptr(), dn.data_ptr(), P, cap, levelsup, o["bid"].data_ptr(), o["bval"].data_ptr(), o["cnt"][0].data_ptr(),
                             o["node"].data_ptr(), o["off"].data_ptr(), o["feat"].data_ptr(), o["cnt"][1].data_ptr(), st)
    return o


def host_fv(o, i):
    node, off, feat, cnt = (o[x].cpu().numpy() for x in ("node", "off", "feat", "cnt"))
    nn = cnt[1, i]
    return node[i, :nn].view(np.uint32), off[i, :nn + 1], feat[i, :off[i, nn]].view(np.uint32)


def setup(pairs, cap, seed_voc=6, k=10, L=4, levelsup=2, voc=None):
    import torch
    P = len(pairs)
    voc = voc if voc is not None else synth.vocabulary(k, L, seed=seed_voc)
    V = capi.ORBVocabulary.from_nodes(k, L, 0, 0, voc["parent"], voc["is_leaf"], voc["desc"], voc["weight"])
    n1 = np.array([len(p["d1"]) for p in pairs], np.int32); n2 = np.array([len(p["d2"]) for p in pairs], np.int32)
    D1 = np.zeros((P, cap, 32), np.uint8); D2 = np.zeros((P, cap, 32), np.uint8)
    K1 = np.zeros((P, cap), dtype=capi.KP_DTYPE); K2 = np.zeros((P, cap), dtype=capi.KP_DTYPE)
    M1 = np.zeros((P, cap), np.uint8); M2 = np.zeros((P, cap), np.uint8)
    Fm = np.zeros((P, 9), np.float32)
    for i, p in enumerate(pairs):
        D1[i, :n1[i]] = p["d1"]; D2[i, :n2[i]] = p["d2"]; K1[i, :n1[i]] = p["k1"]; K2[i, :n2[i]] = p["k2"]
        M1[i, :n1[i]] = p["mp1"]; M2[i, :n2[i]] = p["mp2"]; Fm[i] = p["F"].reshape(9)
    st = torch.cuda.current_stream().cuda_stream
    A = device_feature_vectors(torch, V, D1, n1, P, cap, levelsup, st)
    B = device_feature_vectors(torch, V, D2, n2, P, cap, levelsup, st)
    qrange = torch.zeros((P, cap, 2), dtype=torch.int32, device="cuda")
    nq = torch.zeros(P, dtype=torch.int32, device="cuda")
    capi.bow_ranges_batch_device(A["node"].data_ptr(), A["off"].data_ptr(), A["cnt"][1].data_ptr(), B["node"].data_ptr(), B["off"].data_ptr(),
                                 B["cnt"][1].data_ptr(), cap, P, qrange.data_ptr(), nq.data_ptr(), st)
    torch.cuda.synchronize()
    b_off, b_cnt = B["off"].cpu().numpy(), B["cnt"].cpu().numpy()
    nlist = torch.from_numpy(np.array([b_off[i, b_cnt[1, i]] for i in range(P)], np.int32)).cuda()
    dK1 = torch.from_numpy(K1.view(np.uint8).reshape(P, cap, 28)).cuda(); dK2 = torch.from_numpy(K2.view(np.uint8).reshape(P, cap, 28)).cuda()
    return dict(torch=torch, P=P, st=st, V=V, A=A, B=B, qrange=qrange, nq=nq, nlist=nlist, dK1=dK1, dK2=dK2, n1=n1, n2=n2, K1=K1, K2=K2,
                M1=M1, M2=M2, dF=torch.from_numpy(Fm).cuda())


def outputs(torch, P, cap):
    i32 = torch.int32
    return [torch.full((P, cap), -9, dtype=i32, device="cuda") for _ in range(4)] + [torch.zeros(P, dtype=i32, device="cuda")]


def by_feature(pos_feat, n, arr, fill=-1):
    out = np.full(n, fill, np.int32)
    out[pos_feat] = arr[:len(pos_feat)]
    return out


def run_triangulation(S, cap, th, check, sigma2):
    torch, P, st = S["torch"], S["P"], S["st"]
    q2t, t2q, best, second, nm = outputs(torch, P, cap)
    dMP2 = torch.from_numpy(S["M2"]).cuda()
    dQV = torch.from_numpy((1 - S["M1"]).astype(np.uint8)).cuda()                   # a query is valid when it has NO map point yet
    capi.triangulation_search_batch_device(th, check, S["dF"].data_ptr(), sigma2, S["dK2"].data_ptr(), S["B"]["D"].data_ptr(),
                                           S["B"]["feat"].data_ptr(), S["nlist"].data_ptr(), S["B"]["n"].data_ptr(), cap, dMP2.data_ptr(),
                                           S["qrange"].data_ptr(), S["A"]["feat"].data_ptr(), S["dK1"].data_ptr(), S["A"]["D"].data_ptr(), dQV.data_ptr(),
                                           S["nq"].data_ptr(), cap, P, q2t.data_ptr(), t2q.data_ptr(), best.data_ptr(), second.data_ptr(), nm.data_ptr(), st)
    torch.cuda.synchronize()
    return tuple(x.cpu().numpy() for x in (q2t, t2q, best, second, nm, S["nq"]))


def run_bow_kf(S, cap, th_low, ratio, check):
    """SearchByBoW(KeyFrame*, KeyFrame*): rule BOW, th = TH_LOW - 1, pKF2 features WITHOUT a good map point marked as claimed"""
    torch, P, st = S["torch"], S["P"], S["st"]
    q2t, t2q, best, second, nm = outputs(torch, P, cap)
    V1, V2 = (1 - S["M1"]).astype(np.uint8), (1 - S["M2"]).astype(np.uint8)      # "holds a good map point"
    dV1, dC2 = torch.from_numpy(V1).cuda(), torch.from_numpy((1 - V2).astype(np.uint8)).cuda()
    dA1 = torch.from_numpy(np.ascontiguousarray(S["K1"]["angle"])).cuda()
    capi.list_search_batch_device(capi.RULE_BOW, th_low - 1, ratio, check, S["dK2"].data_ptr(), S["B"]["D"].data_ptr(), S["B"]["feat"].data_ptr(),
                                  S["nlist"].data_ptr(), S["B"]["n"].data_ptr(), cap, dC2.data_ptr(), S["qrange"].data_ptr(), S["A"]["feat"].data_ptr(),
                                  S["A"]["D"].data_ptr(), dA1.data_ptr(), dV1.data_ptr(), S["nq"].data_ptr(), cap, P, q2t.data_ptr(), t2q.data_ptr(),
                                  best.data_ptr(), second.data_ptr(), nm.data_ptr(), st)
    q2t_n, t2q_n, _, _, nm_n = outputs(torch, P, cap)                                # ... and without the distance outputs: the same matches
    capi.list_search_batch_device(capi.RULE_BOW, th_low - 1, ratio, check, S["dK2"].data_ptr(), S["B"]["D"].data_ptr(), S["B"]["feat"].data_ptr(),
                                  S["nlist"].data_ptr(), S["B"]["n"].data_ptr(), cap, dC2.data_ptr(), S["qrange"].data_ptr(), S["A"]["feat"].data_ptr(),
                                  S["A"]["D"].data_ptr(), dA1.data_ptr(), dV1.data_ptr(), S["nq"].data_ptr(), cap, P, q2t_n.data_ptr(), t2q_n.data_ptr(),
                                  0, 0, nm_n.data_ptr(), st)
    torch.cuda.synchronize()
    assert torch.equal(nm, nm_n) and torch.equal(q2t, q2t_n) and torch.equal(t2q, t2q_n), "results depend on whether best / second are asked for"
    return tuple(x.cpu().numpy() for x in (q2t, t2q, nm, S["nq"])) + (V1, V2)


def inverse_by_feature(pos_feat, t2q_row):
    out = t2q_row.copy()
    m = out >= 0
    out[m] = pos_feat[out[m]]
    return out
