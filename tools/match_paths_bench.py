#!/usr/bin/env python3
"""Timing of the dense top-2 kernel families side by side (development aid; bench.py is the contract): the per-frame batch form
(B x (n x n)) and one large problem (N x N), each through path 0 (xor + popcount), 1 (int8 MFMA), 2 (FP4 MFMA); results compared."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from orb_slam_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1024); ap.add_argument("--n", type=int, default=1000)
ap.add_argument("--big", type=int, default=100000); ap.add_argument("--reps", type=int, default=20)
ap.add_argument("--paths", default="1,2,0"); ap.add_argument("--sustain", type=float, default=1.0)
a = ap.parse_args()
st = torch.cuda.current_stream().cuda_stream
out = {}

def timed(fn, reps):
    """per-call events (min / median / max) and the SUSTAINED average: back-to-back calls for >= --sustain seconds between one event pair
    (the matrix-core kernels clock down under load: the sustained figure is the one bench.py's timed region sees)"""
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    n = max(10, int(a.sustain * 1e3 / ts[len(ts) // 2]))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); e1.synchronize()
    return {"min_ms": round(ts[0], 4), "median_ms": round(ts[len(ts) // 2], 4), "max_ms": round(ts[-1], 4), "sustained_ms": round(e0.elapsed_time(e1) / n, 4), "sustained_calls": n}

B, n = a.batch, a.n
D = torch.from_numpy(synth.descriptors((B + 1) * n, 5).reshape(B + 1, n, 32)).cuda()
nq = torch.full((B,), n, dtype=torch.int32, device="cuda")
ref = None
for p in [int(x) for x in a.paths.split(",")]:
    capi.set_match_path(p)
    o = torch.full((3, B, n), -7, dtype=torch.int32, device="cuda")
    fn = lambda: capi.match_top2_batch_device(D[1:].data_ptr(), nq.data_ptr(), D[:-1].data_ptr(), nq.data_ptr(), B, n, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), st)
    r = timed(fn, a.reps)
    r["pairs_per_s"] = B * n * n / (r["median_ms"] * 1e-3)
    if ref is None: ref = o.clone(); r["equal_to_first"] = True
    else: r["equal_to_first"] = bool((o == ref).all().item())
    out["batch_path%d" % p] = r
if a.big:
    N = a.big
    Q = torch.from_numpy(synth.descriptors(N, 78)).cuda(); T = torch.from_numpy(synth.descriptors(N, 79)).cuda()
    ref = None
    for p in [int(x) for x in a.paths.split(",")]:
        capi.set_match_path(p)
        o = torch.full((3, N), -7, dtype=torch.int32, device="cuda")
        fn = lambda: capi.match_top2_device(Q.data_ptr(), N, T.data_ptr(), N, o[0].data_ptr(), o[1].data_ptr(), o[2].data_ptr(), st)
        r = timed(fn, max(5, a.reps // 2 if p else 5))
        r["pairs_per_s"] = float(N) * N / (r["median_ms"] * 1e-3)
        if ref is None: ref = o.clone(); r["equal_to_first"] = True
        else: r["equal_to_first"] = bool((o == ref).all().item())
        out["big_path%d" % p] = r
capi.set_match_path(-1)
print(json.dumps(out, indent=1))
