#!/usr/bin/env python3
"""What does running the VALU-bound part of one lane NEXT TO the memory-bound parts of another buy?  (37 % of the
VALU issue slots of a step are idle, the step is the plain sum of its kernels.)  Two extractor handles on two streams; each part of
the extraction (orbx_extract_batch_device_phases) timed alone and in pairs, wall clock around `reps` back-to-back repetitions."""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=512)
    ap.add_argument("--width", type=int, default=640)
    ap.add_argument("--height", type=int, default=480)
    ap.add_argument("--nfeatures", type=int, default=1000)
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    import torch
    from orb_slam_amd import capi, synth
    dev = torch.device("cuda", 0)
    b, w, h = a.frames, a.width, a.height
    P, D, S = capi.PHASE_PYRAMID, capi.PHASE_DETECT, capi.PHASE_DESCRIBE

    class Lane:
        def __init__(self, first):
            self.ex = capi.ORBextractor(nfeatures=a.nfeatures, device=0, max_batch=b)
            self.ex_next = capi.ORBextractor(nfeatures=a.nfeatures, device=0, max_batch=b)      # the pyramid of step i + 1 (the library refuses parts queued out of order on one handle)
            self.stream = torch.cuda.ExternalStream(capi.stream_create(0), device=dev)
            self.img = torch.from_numpy(synth.frames(w, h, synth.BLOCKS, first, b)).to(dev)
            cap = self.cap = self.ex.max_keypoints
            self.kps = torch.zeros((b, cap, 7), dtype=torch.float32, device=dev)
            self.desc = torch.zeros((b + 1, cap, 32), dtype=torch.uint8, device=dev)
            self.n = torch.zeros(b + 1, dtype=torch.int32, device=dev)
            self.match = torch.zeros((3, b, cap), dtype=torch.int32, device=dev)

        def run(self, phases, ex=None):
            (ex or self.ex).extract_batch_device(self.img.data_ptr(), b, w, h, w, w * h, self.kps.data_ptr(), self.desc[1].data_ptr(), self.n[1:].data_ptr(),
                                         self.cap, 0, self.stream.cuda_stream, phases=phases)

        def do_match(self):
            capi.match_top2_batch_device(self.desc[1].data_ptr(), self.n[1:].data_ptr(), self.desc[0].data_ptr(), self.n.data_ptr(), b, self.cap,
                                         self.match[0].data_ptr(), self.match[1].data_ptr(), self.match[2].data_ptr(), self.stream.cuda_stream)

    A, B = Lane(0), Lane(5000)
    torch.cuda.synchronize()
    for ln in (A, B):
        ln.run(capi.PHASE_ALL)
        ln.desc[0].copy_(ln.desc[1])
        ln.n[0:1].copy_(ln.n[1:2])
    torch.cuda.synchronize()

    def f_detect(ln): ln.run(D)
    def f_pyr(ln): ln.run(P)
    def f_desc(ln): ln.run(S)
    def f_match(ln): ln.do_match()
    def f_mem(ln): ln.run(S); ln.do_match(); ln.run(P, ln.ex_next)          # describe + match of step i, pyramid of step i + 1
    def f_all(ln): ln.run(capi.PHASE_ALL); ln.do_match()

    def timeit(jobs):
        """jobs: list of (lane, fn); every job queued `reps` times on its lane's stream, all lanes together"""
        for ln, fn in jobs:
            fn(ln)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(a.reps):
            for ln, fn in jobs:
                fn(ln)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) * 1e3 / a.reps

    out = {"frames_per_lane": b, "size": [w, h], "nfeatures": a.nfeatures, "reps": a.reps, "ms": {}}
    r = out["ms"]
    r["pyramid"] = timeit([(A, f_pyr)])
    r["detect (FAST + blur || selection)"] = timeit([(A, f_detect)])
    r["describe"] = timeit([(A, f_desc)])
    r["match"] = timeit([(A, f_match)])
    r["describe + match + pyramid"] = timeit([(B, f_mem)])
    r["all, one lane"] = timeit([(A, f_all)])
    r["all, two lanes free-running"] = timeit([(A, f_all), (B, f_all)])
    r["detect(A) || describe + match + pyramid(B)"] = timeit([(A, f_detect), (B, f_mem)])
    r["detect(A) || describe(B)"] = timeit([(A, f_detect), (B, f_desc)])
    r["detect(A) || pyramid(B)"] = timeit([(A, f_detect), (B, f_pyr)])
    r["detect(A) || match(B)"] = timeit([(A, f_detect), (B, f_match)])
    r["detect(A) || detect(B)"] = timeit([(A, f_detect), (B, f_detect)])
    r["describe(A) || describe(B)"] = timeit([(A, f_desc), (B, f_desc)])
    for k in r:
        r[k] = round(r[k], 4)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
