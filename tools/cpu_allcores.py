#!/usr/bin/env python3
"""the CPU oracle on ALL host cores (one extractor instance per process — instances are not
re-entrant, like the reference's), frames sharded, extract + top-2 match vs the previous frame of the same worker."""
import multiprocessing as mp, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))

def work(args):
    idx, w, h, nf, seconds = args
    import oracle_lib as orc
    from orb_slam_amd import synth
    o = orc.OracleExtractor(nfeatures=nf)
    imgs = synth.frames(w, h, synth.BLOCKS, 7000 + 64 * idx, 16)
    prev = o(imgs[0])[1]
    n, t = 0, time.perf_counter()
    while time.perf_counter() - t < seconds:
        _, d = o(imgs[(n + 1) % 16])
        if len(d) and len(prev): orc.match_top2(d, prev)
        prev = d; n += 1
    return n, time.perf_counter() - t

if __name__ == "__main__":
    w, h, nf = (int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480, 1000)
    cores = len(os.sched_getaffinity(0))
    with mp.get_context("spawn").Pool(cores) as pool:
        res = pool.map(work, [(i, w, h, nf, 10.0) for i in range(cores)])
    fps = sum(n / t for n, t in res)
    print("oracle on %d cores: %.0f frames/s (%dx%d, nFeatures %d, extract + match) = %.1f per core" % (cores, fps, w, h, nf, fps / cores))
