#!/usr/bin/env python3
"""Randomised differential test of the THROUGHPUT path (orbx_extract_batch_device: per-level k_resize, full-batch k_fast_cells, k_describe_od with the blur
per keypoint window — or, in one case of seven, k_blur / k_blur_mfma + k_describe —, frame -> XCD block renumbering from 64 frames): random image sizes,
constructor arguments (both float modes, both blur roundings), families incl. the correlated S-warp stream, row pitches incl. byte-unaligned ones, and launch
group sizes (32 .. 96 frames, max_batch sometimes smaller than the batch so that a call spans several launch groups); every frame
of every case against the CPU oracle, byte for byte.  tools/fuzz_parity.py does the same for the one-frame call (orbx_extract).
usage: fuzz_batch.py [cases] [seed]   — prints one JSON line."""
import json, os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import torch
import oracle_lib as ol
from orb_slam_amd import capi, synth

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
try:
    q, p = open("/sys/fs/cgroup/cpu.max").read().split()
    workers = int(float(q) / float(p)) if q != "max" else len(os.sched_getaffinity(0))
except Exception:
    workers = len(os.sched_getaffinity(0))
workers = max(1, min(workers, 32))
ok = geo = lim = frames_checked = n_od = n_fpc = 0
bad = []
t0 = time.time()
for c in range(cases):
    w = int(rng.integers(160, 900)); h = int(rng.integers(120, 700))
    if rng.random() < 0.35:
        w, h = [(640, 480), (752, 480), (1280, 720), (320, 240), (1241, 376), (1920, 1080)][int(rng.integers(0, 6))]
    nf = int(rng.choice([200, 500, 1000, 1500, 2000]))
    sf = float(rng.choice([1.1, 1.2, 1.2, 1.25, 1.3, 1.5]))
    nl = int(rng.integers(2, 9))
    st = int(rng.random() < 0.2) ^ 1
    th = int(rng.choice([5, 7, 10, 20, 20, 30]))
    blur = int(rng.random() < 0.2)
    B = int(rng.integers(32, 97)) if w * h <= 1280 * 720 else int(rng.integers(32, 41))
    max_batch = B if rng.random() < 0.7 else int(rng.integers(32, B + 1))
    pad = int(rng.choice([0, 0, 4, 8]))                      # row stride beyond the width (multiple of 4: the aligned kernels)
    if rng.random() < 0.5:
        pad = (-w) % 16 + 16 * int(rng.integers(0, 3))       # rows of whole 16-byte chunks: what k_blur_mfma (levels up to 1024 px wide) takes
    fams = rng.choice([0, 1, 1, 1, 3, 4, 5, 5], size=B)
    r = rng.random()
    if r < 0.15:
        pad = int(rng.choice([1, 2, 3, 5, 7, 13]))           # byte-unaligned pitches (round 6: the 16-byte LDS-DMA takes any source alignment)
    fpc = bool(rng.random() < 0.25)                          # orbx_params::fp_contract: the reference as its own build flags contract it
    od = 0 if rng.random() < 0.15 else 1                     # 0: blur kernels + k_describe; 1 (default): k_describe_od
    try:
        capi.geometry(w, h, nfeatures=nf, scaleFactor=sf, nlevels=nl, scoreType=st, fastTh=th)
    except capi.OrbxError as e:
        if e.code == capi.ORBX_ERR_GEOMETRY:
            geo += 1
        else:
            lim += 1
        continue
    first = int(rng.integers(0, 100000))
    frames = np.stack([synth.frame(w, h, int(fams[i]), first + i) for i in range(B)])
    rs = w + pad
    buf = np.zeros((B, h, rs), np.uint8)
    buf[:, :, :w] = frames
    d_img = torch.from_numpy(buf).cuda()
    ex = capi.ORBextractor(nfeatures=nf, scaleFactor=sf, nlevels=nl, scoreType=st, fastTh=th, blur_rounding=blur, max_batch=max_batch, fp_contract=fpc)
    ex.set_blur_on_demand(od)
    n_od += od; n_fpc += int(fpc)
    cap = ex.max_keypoints
    d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda")
    d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
    d_st = torch.full((B,), -99, dtype=torch.int32, device="cuda")
    ex.extract_batch_device(d_img.data_ptr(), B, w, h, rs, rs * h, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, d_st.data_ptr(),
                            torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    n = d_n.cpu().numpy(); stt = d_st.cpu().numpy()
    kps = d_kps.cpu().numpy().view(np.uint8).reshape(B, cap, 28); desc = d_desc.cpu().numpy()
    ex.close()

    def oracle(idx):
        o = ol.OracleExtractor(nf, sf, nl, st, th, blur_mode=blur, fp_contract=fpc)
        return [(j, o(frames[j])) for j in idx]
    want = [None] * B
    with ThreadPoolExecutor(workers) as pool:
        for part in pool.map(oracle, [list(range(i, B, workers)) for i in range(workers)]):
            for j, r in part:
                want[j] = r
    wrong = [j for j in range(B) if stt[j] != 0 or n[j] != len(want[j][0]) or kps[j, :n[j]].tobytes() != want[j][0].tobytes()
             or desc[j, :n[j]].tobytes() != want[j][1].tobytes()]
    frames_checked += B
    if wrong:
        bad.append(dict(case=c, w=w, h=h, nf=nf, sf=sf, nl=nl, st=st, th=th, blur=blur, B=B, max_batch=max_batch, pad=pad, fp_contract=fpc, on_demand=od, frames=wrong[:8]))
    else:
        ok += 1
print(json.dumps({"cases": cases, "seed": seed, "bit_exact_cases": ok, "frames_checked": frames_checked, "geometry_the_reference_cannot_process": geo,
                  "implementation_limit": lim, "cases_on_demand": n_od, "cases_fp_contract": n_fpc, "mismatches": bad, "seconds": round(time.time() - t0, 1), "build": capi.build_id()}))
