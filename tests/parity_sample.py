"""Checker for the throughput configuration (orb_slam_amd/pipeline.py): the frames of a LanePipeline step — all of them (bench.py's
default since round 4, on a thread pool) or a sample (every lane border from both sides, the step border, a few interior frames) —
compared with the CPU oracle: keypoints and descriptors
byte for byte, the top-2 match against the previous frame integer for integer.

Test infrastructure (it imports the oracle): used by tests/test_gpu_bench_shapes.py and by bench.py's parity leg, which runs
AFTER the timed region on the outputs the last timed step left on the device — it checks the measured path, it is never the
thing measured."""
import numpy as np

import oracle_lib as orc


def sample_indices(B, G, extra=4):
    """frames of a B-frame step run as G lanes of b = B / G: first and last frame of every lane (the first one's predecessor
    arrives through the hand-off buffer, the last one feeds it; frame 0's predecessor is the previous step's last frame) plus
    `extra` frames from lane interiors"""
    b = B // G
    s = set()
    for g in range(G):
        s.add(g * b)
        s.add(g * b + b - 1)
    for k in range(extra):
        s.add(((2 * k + 1) * B) // (2 * extra))
    return sorted(j for j in s if 0 <= j < B)


def check_step(pipe, host_frame, sample, nfeatures, has_previous_step=True, oracle_kw=None, threads=1):
    """pipe: a LanePipeline whose last step has completed (device synchronised).  host_frame(j) -> the uint8 image of frame j of
    that step (j = -1: the last frame of the step before).  `sample`: frame indices to check (range(B) = every frame).  With
    threads > 1 the oracle runs on a pool (one extractor instance per thread: instances are not re-entrant; the C calls release
    the GIL).  Returns {"frames": checked, "mismatches": count, "detail": [...]}"""
    import threading
    from concurrent.futures import ThreadPoolExecutor
    sample = list(sample)
    need = sorted(set(sample) | ({j - 1 for j in sample if j > 0 or has_previous_step} if pipe.do_match else set()))
    local = threading.local()

    def want(j):
        if not hasattr(local, "o"):
            local.o = orc.OracleExtractor(nfeatures=nfeatures, **(oracle_kw or {}))
        return j, local.o(host_frame(j))

    n_all = pipe.counts().cpu().numpy()
    st_all = pipe.status().cpu().numpy()
    # the step's outputs, one bulk copy per lane
    host = [(ln.kps.cpu().numpy(), ln.desc.cpu().numpy(), ln.match.cpu().numpy() if pipe.do_match else None) for ln in pipe.lanes]

    def check(j, cache):
        ok, od = cache[j]
        n = int(n_all[j])
        g, r = divmod(j, pipe.b)
        hk, hd, hm = host[g]
        bad = []
        if st_all[j] != 0:
            bad.append("status %d" % st_all[j])
        if n != len(ok):
            bad.append("n %d != %d" % (n, len(ok)))
        else:
            if hk[r, :n].tobytes() != ok.tobytes():
                bad.append("keypoints differ")
            if not np.array_equal(hd[1 + r, :n], od):
                bad.append("descriptors differ")
        if pipe.do_match and not bad:
            m = hm[:, r, :n]
            if j == 0 and not has_previous_step:
                ri = np.full(n, -1, np.int32)
                rb = rs = np.full(n, 2**31 - 1, np.int32)
            else:
                ri, rb, rs = orc.match_top2(od, cache[j - 1][1])
            if not (np.array_equal(m[0], ri) and np.array_equal(m[1], rb) and np.array_equal(m[2], rs)):
                bad.append("top-2 match vs previous frame differs")
        return {"frame": int(j), "what": bad} if bad else None

    if threads > 1:
        with ThreadPoolExecutor(threads) as pool:
            cache = dict(pool.map(want, need))
            res = list(pool.map(lambda j: check(j, cache), sample))
    else:
        cache = dict(want(j) for j in need)
        res = [check(j, cache) for j in sample]
    detail = [r for r in res if r]
    return {"frames": len(sample), "mismatches": len(detail), "detail": detail[:8]}
