#!/usr/bin/env python3
"""Randomised differential test of the descriptor matcher (orbm_match_top2_device, orbm_match_top2_batch_device,
orbm_match_top2_segments_device) against the sequential-scan oracle: random query / train sizes around every tile boundary of the
kernels (32-row MFMA tiles, 128 / 256-query blocks, the train splits of large scans), empty sets, planted duplicates, exact matches
and low-entropy descriptors (dense distance ties: first index and the multiplicity of the second-best must survive), arrays that
start 4 bytes into their allocation, and the three kernel families (FP4 MFMA, int8 MFMA, xor + popcount).  Integer exact or it counts as bad.
usage: fuzz_match.py [cases] [seed]   — prints one JSON line."""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle_lib as orc
from orb_slam_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    workers = int(float(q) / float(p)) if q != "max" else len(os.sched_getaffinity(0))
except Exception:
    workers = len(os.sched_getaffinity(0))
workers = max(1, min(workers, 32))
pool = ThreadPoolExecutor(workers)
EDGES = [1, 2, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 511, 512, 513, 1000, 1023, 1024, 1025, 2000, 2048, 4095, 4096, 4097]


def size(big):
    r = rng.random()
    if r < 0.05:
        return 0
    if r < 0.45:
        return int(rng.choice([e for e in EDGES if e < big]))
    if r < 0.9 or big <= 3000:
        return int(rng.integers(1, min(3000, big)))
    return int(rng.integers(3000, big))


def descriptors(n, kind):
    if n == 0:
        return np.zeros((0, 32), np.uint8)
    if kind == 0:
        return synth.descriptors(n, int(rng.integers(1, 1 << 30)))
    if kind == 1:                                                          # 2 bits of entropy per byte: dense ties
        return rng.integers(0, 4, size=(n, 32)).astype(np.uint8)
    d = synth.descriptors(n, int(rng.integers(1, 1 << 30)))                 # clusters: a few centres, a few flipped bits each
    centres = d[rng.integers(0, max(n // 50, 1), n)]
    flips = rng.integers(0, 256, size=(n, 3))
    out = centres.copy()
    for k in range(3):
        out[np.arange(n), flips[:, k] // 8] ^= (1 << (flips[:, k] % 8)).astype(np.uint8)
    return out


def oracle(Q, T):
    blocks = [(a, min(a + 256, len(Q))) for a in range(0, len(Q), 256)] or [(0, 0)]
    res = list(pool.map(lambda ab: orc.match_top2(Q[ab[0]:ab[1]], T), blocks))
    return [np.concatenate([r[k] for r in res]) for k in range(3)]


def dev(a, off):
    """device copy of a byte array starting `off` bytes into its allocation"""
    buf = torch.zeros(a.size + off + 4, dtype=torch.uint8, device="cuda")
    if a.size:
        buf[off:off + a.size] = torch.from_numpy(a.reshape(-1)).cuda()
    return buf, buf.data_ptr() + off


ok = pairs = 0
bad = []
kinds = {"dense": 0, "batch": 0, "segments": 0}
t0 = time.time()
st = torch.cuda.current_stream().cuda_stream
for c in range(cases):
    path = int(rng.choice([0, 1, 2, 2]))                                   # 2 = FP4 MFMA, 1 = int8 MFMA, 0 = xor + popcount
    capi.set_match_path(path)
    mode = rng.choice(["dense", "dense", "batch", "segments"])
    kinds[str(mode)] += 1
    kq, kt = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    off = int(rng.choice([0, 0, 4, 8, 12]))
    try:
        if mode == "dense":
            nq, nt = size(6000), size(60000)
            Q, T = descriptors(nq, kq), descriptors(nt, kt)
            if nq and nt and rng.random() < 0.5:
                k = min(nq, nt, 64)
                Q[rng.integers(0, nq, k)] = T[rng.integers(0, nt, k)]     # exact matches
                T[rng.integers(0, nt, max(nt // 20, 1))] = T[int(rng.integers(0, nt))]   # repeated train rows
            bq, pq = dev(Q, off); bt, pt = dev(T, off)
            out = torch.full((3, max(nq, 1)), -7, dtype=torch.int32, device="cuda")
            capi.match_top2_device(pq, nq, pt, nt, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st)
            torch.cuda.synchronize()
            o = out.cpu().numpy()[:, :nq]
            r = oracle(Q, T)
            good = all(np.array_equal(o[k], r[k]) for k in range(3))
            pairs += nq * nt
        elif mode == "batch":
            B, cap = int(rng.integers(1, 24)), int(rng.choice([100, 500, 1000, 1000, 2000, 2500]))
            nq = np.minimum(np.array([size(cap + 1) for _ in range(B)]), cap).astype(np.int32)
            nt = np.minimum(np.array([size(cap + 1) for _ in range(B)]), cap).astype(np.int32)
            Q = np.stack([descriptors(cap, kq) for _ in range(B)]); T = np.stack([descriptors(cap, kt) for _ in range(B)])
            for i in range(B):
                if nt[i] > 3 and rng.random() < 0.5:
                    T[i, rng.integers(0, nt[i], max(int(nt[i]) // 10, 1))] = T[i, 0]
            bq, pq = dev(Q, off); bt, pt = dev(T, off)
            dnq, dnt = torch.from_numpy(nq).cuda(), torch.from_numpy(nt).cuda()
            out = torch.full((3, B, cap), -7, dtype=torch.int32, device="cuda")
            capi.match_top2_batch_device(pq, dnq.data_ptr(), pt, dnt.data_ptr(), B, cap, out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr(), st)
            torch.cuda.synchronize()
            o = out.cpu().numpy()
            good = True
            for i in range(B):
                r = orc.match_top2(Q[i, :nq[i]], T[i, :nt[i]])
                good &= all(np.array_equal(o[k, i, :nq[i]], r[k]) for k in range(3)) and bool((o[:, i, nq[i]:] == -7).all())
                pairs += int(nq[i]) * int(nt[i])
        else:
            nq, nt = max(size(3000), 1), max(size(8000), 1)
            Q, T = descriptors(nq, kq), descriptors(nt, kt)
            T[: nt // 3] = T[0]
            lens = rng.integers(0, int(rng.choice([4, 40, 300])) + 1, nq)
            lens[rng.integers(0, nq, max(nq // 10, 1))] = 0
            seg = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
            cand = rng.integers(0, nt, int(seg[-1])).astype(np.int32)
            bq, pq = dev(Q, off); bt, pt = dev(T, off)
            dseg, dcand = torch.from_numpy(seg).cuda(), torch.from_numpy(np.concatenate([cand, [0]]).astype(np.int32)).cuda()
            out = torch.full((3, nq), -7, dtype=torch.int32, device="cuda")
            rc = capi.lib().orbm_match_top2_segments_device(pq, nq, pt, nt, dseg.data_ptr(), dcand.data_ptr(), out[0].data_ptr(), out[1].data_ptr(),
                                                            out[2].data_ptr(), st)
            torch.cuda.synchronize()
            o = out.cpu().numpy()
            r = orc.match_top2_segments(Q, T, seg, cand)
            good = rc == 0 and all(np.array_equal(o[k], r[k]) for k in range(3))
            pairs += int(seg[-1])
    except Exception as e:                                                  # a refused call is a failure here: every case is a legal one
        good = False
        bad.append({"case": c, "mode": str(mode), "error": repr(e)[:200]})
    if good:
        ok += 1
    elif not bad or bad[-1].get("case") != c:
        bad.append({"case": c, "mode": str(mode), "path": path, "kinds": [kq, kt], "offset": off})
capi.set_match_path(-1)
print(json.dumps({"cases": cases, "seed": seed, "ok": ok, "mismatches": len(bad), "by_entry_point": kinds, "pairs_checked": pairs,
                  "seconds": round(time.time() - t0, 1), "oracle_threads": workers, "build": capi.build_id(), "bad": bad[:10]}))
sys.exit(1 if bad else 0)
