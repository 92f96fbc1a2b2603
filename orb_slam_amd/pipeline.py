"""Host side of the throughput configuration (NOTES.md §4.5): a step's B consecutive frames go through G lanes of B/G
consecutive frames, each lane with its own extractor handle and HIP stream, every frame matched against its predecessor.

Lanes never join.  The only cross-lane dependency is the one frame per lane whose predecessor lies in the lane to its left
(lane 0: in the last lane's slice of the previous step): that frame's descriptors travel through a two-slot hand-off buffer
ordered by HIP events.  torch is used for device memory, streams and events only."""
import ctypes
import os
import time

import torch

from . import capi


class _Lane:
    """One contiguous b-frame slice of every step.  Slot 0 of desc / n holds the frame before the slice."""

    def __init__(self, ex, device, b, stream):
        dev = torch.device("cuda", device)
        self.ex = ex
        self.stream = stream
        cap = self.ex.max_keypoints
        self.kps = torch.zeros((b, cap, 7), dtype=torch.float32, device=dev)
        self.desc = torch.zeros((b + 1, cap, 32), dtype=torch.uint8, device=dev)
        self.n = torch.zeros(b + 1, dtype=torch.int32, device=dev)
        self.status = torch.zeros(b, dtype=torch.int32, device=dev)
        self.match = torch.zeros((3, b, cap), dtype=torch.int32, device=dev)       # train index, best, second per query slot
        # hand-off of the slice's last frame to the lane on the right: two slots (step parity)
        self.h_desc = torch.zeros((2, cap, 32), dtype=torch.uint8, device=dev)
        self.h_n = torch.zeros((2, 1), dtype=torch.int32, device=dev)
        self.h_written = [None, None]
        self.h_consumed = [None, None]


class LanePipeline:
    def __init__(self, width, height, batch, lanes=4, nfeatures=1000, device=0, do_match=True, autotune=True, placement=None,
                 **extractor_kw):
        """autotune: create the candidate stream sets tune() chooses from (nothing is probed until tune() is called).
        placement (or ORBX_LANE_PLACEMENT=k in the environment): use candidate k and never probe — what a process that was tuned once,
        or one of 8 ranks that should not each spend probe steps, passes."""
        G = max(1, min(lanes, batch))
        while batch % G:
            G -= 1
        self.w, self.h, self.B, self.G, self.b = width, height, batch, G, batch // G
        self.do_match = do_match
        # Stream placement.  The HIP runtime binds a stream to one of its hardware queues (GPU_MAX_HW_QUEUES, 4 by default) when the
        # stream is created: new queues until 4 exist, then the least-loaded one.  Streams on one hardware queue are launched in
        # order.  Measured best (NOTES.md §4.5, rocprofv3 Queue_Id column): every lane stream on a hardware queue of its own, the
        # blur side streams (created inside the extractor handles) sharing those queues.  Creating the G handles first and the G
        # lane streams after them, back to back, gives that placement in a fresh process — but any library that created streams
        # earlier (torch's pool, an RCCL communicator) shifts it.  So the pipeline does not trust the creation order: it creates
        # three candidate sets of lane streams (with 0, 1 and 2 spacer streams in front, i.e. rotated against the side streams'
        # queues); tune() — an explicit, blocking call — times them and keeps the fastest (`self.placement` reports the timings
        # and the choice).  Raw HIP streams from the C ABI (torch creates its pool streams lazily, in an order of its own).
        if placement is None and os.environ.get("ORBX_LANE_PLACEMENT", "") != "":
            try:
                placement = int(os.environ["ORBX_LANE_PLACEMENT"])
            except ValueError:
                raise ValueError("ORBX_LANE_PLACEMENT=%r: expected a candidate index 0..2" % os.environ["ORBX_LANE_PLACEMENT"]) from None
        if placement is not None and not 0 <= placement <= 2:       # (LanePipeline.h refuses the same values)
            raise ValueError("lane placement %d: expected a candidate index 0..2" % placement)
        dev = torch.device("cuda", device)
        handles = [capi.ORBextractor(nfeatures=nfeatures, device=device, max_batch=self.b, **extractor_kw) for _ in range(G)]
        self._raw_sets, self._spacers = [], []
        ncand = 3 if (G > 1 and (autotune or placement is not None)) else 1
        if ncand == 3 and os.environ.get("ORBX_LANE_CANDIDATES", "").isdigit():      # experiment switch (NOTES.md 10.9): more rotations against the side streams' queues
            ncand = max(3, min(8, int(os.environ["ORBX_LANE_CANDIDATES"])))
        # ORBX_LANE_PRIORITIES="p0,p1,..." (an experiment switch, NOTES.md 10.9): HIP stream priorities of the lanes, cyclically (-1 high, 0 normal, 1 low)
        prios = [int(x) for x in os.environ.get("ORBX_LANE_PRIORITIES", "").split(",") if x.strip() != ""]

        def lane_stream(g):
            if not prios:
                return capi.stream_create(device)
            p = ctypes.c_void_p()
            rc = capi.lib().orbx_stream_create_priority(device, prios[g % len(prios)], ctypes.byref(p))
            if rc != capi.ORBX_OK:
                raise capi.OrbxError(rc, "orbx_stream_create_priority")
            return p.value

        for spacer in range(ncand):
            self._spacers += [capi.stream_create(device) for _ in range(spacer)]
            self._raw_sets.append([lane_stream(g) for g in range(G)])
        self._sets = [[torch.cuda.ExternalStream(p, device=dev) for p in raw] for raw in self._raw_sets]
        chosen = min(placement, ncand - 1) if placement is not None else 0          # (one lane: a single candidate)
        self.lanes = [_Lane(handles[g], device, self.b, self._sets[chosen][g]) for g in range(G)]
        self.device = device
        self.cap = self.lanes[0].ex.max_keypoints
        self.steps_done = 0
        self.match_events = []
        self.placement = {"candidates": ncand, "chosen": chosen, "probe_ms_per_step": None, "fixed": placement is not None, "untuned_steps": 0,
                          "note": "candidate k = lane streams created behind k spacer streams (and behind the handles' side streams)"}
        self._fixed = placement is not None or ncand == 1
        self._tuning = False
        torch.cuda.synchronize(dev)      # the zero-fills above ran on the default stream; the lane streams do not wait for it

    def _use_set(self, k):
        for g, ln in enumerate(self.lanes):
            ln.stream = self._sets[k][g]

    def _sync_lanes(self):
        """waits for this pipeline's own streams only (every side stream joins its lane stream before the lane's last kernel):
        other streams and handles of the process keep running"""
        for ln in self.lanes:
            ln.stream.synchronize()

    def _reset_handoff(self):
        self.steps_done = 0
        self.match_events = []
        for ln in self.lanes:
            ln.h_written, ln.h_consumed = [None, None], [None, None]
            with torch.cuda.stream(ln.stream):
                ln.n.zero_()
                ln.h_n.zero_()

    def tune(self, d_frames_ptr, frame_stride=None, row_stride=None, timed_steps=3):
        """Explicit, BLOCKING placement probe: one untimed + `timed_steps` timed steps over the B frames at d_frames_ptr on every
        candidate stream set; the fastest stays.  Call it once after construction, before the real stream starts (the bench does, in
        front of its warm-up).  State the probes touch (step counter, hand-off slots, counts) is reset, so the first real step starts
        exactly as without them.  Only this pipeline's streams are synchronised.  No-op with a fixed placement."""
        if self._fixed:
            return self.placement
        row_stride = row_stride or self.w
        frame_stride = frame_stride or row_stride * self.h
        ms = []
        self._tuning = True
        for k in range(len(self._sets)):
            self._use_set(k)
            self._reset_handoff()
            self._sync_lanes()
            self.step(d_frames_ptr, frame_stride, row_stride)
            self._sync_lanes()
            t = time.perf_counter()
            for _ in range(timed_steps):
                self.step(d_frames_ptr, frame_stride, row_stride)
            self._sync_lanes()
            ms.append((time.perf_counter() - t) * 1e3 / timed_steps)
        best = min(range(len(ms)), key=lambda k: ms[k])
        self._use_set(best)
        self._reset_handoff()
        self._sync_lanes()
        self._fixed, self._tuning = True, False
        self.placement.update({"chosen": best, "probe_ms_per_step": [round(v, 4) for v in ms]})
        return self.placement

    def step(self, d_frames_ptr, frame_stride=None, row_stride=None, timed=False):
        """d_frames_ptr: device address of the step's first frame (B frames, frame_stride bytes apart).  Asynchronous: lane g
        extracts frames [g*b, (g+1)*b) on its own stream, publishes its last frame, takes the frame before its slice from the
        lane on its left and matches every frame against its predecessor.  Never probes or blocks (tune() is the explicit probe)."""
        w, h, b, G, cap = self.w, self.h, self.b, self.G, self.cap
        row_stride = row_stride or w
        frame_stride = frame_stride or row_stride * h
        if not self._fixed and not self._tuning:
            # autotune was asked for but tune() has not run: the step goes through candidate 0 (round 3 made the probe an explicit
            # call; step() never blocks).  Said once, and counted in self.placement.
            if self.placement["untuned_steps"] == 0:
                import warnings
                warnings.warn("LanePipeline.step() before tune(): running on stream candidate 0 (call tune() once, or pass placement=k)", stacklevel=2)
            self.placement["untuned_steps"] += 1
        i = self.steps_done
        par = i & 1
        for g, ln in enumerate(self.lanes):
            s = ln.stream
            with torch.cuda.stream(s):
                ln.ex.extract_batch_device(d_frames_ptr + g * b * frame_stride, b, w, h, row_stride, frame_stride, ln.kps.data_ptr(),
                                           ln.desc[1].data_ptr(), ln.n[1:].data_ptr(), cap, ln.status.data_ptr(), s.cuda_stream)
                if not self.do_match:
                    continue
                if ln.h_consumed[par] is not None:
                    s.wait_event(ln.h_consumed[par])              # the slot's reader of step i-2 is done
                ln.h_desc[par].copy_(ln.desc[b], non_blocking=True)
                ln.h_n[par].copy_(ln.n[b:b + 1], non_blocking=True)
                ln.h_written[par] = torch.cuda.Event()
                ln.h_written[par].record(s)
                src, sp = (self.lanes[g - 1], par) if g > 0 else (self.lanes[G - 1], par ^ 1)
                if g > 0 or i > 0:
                    s.wait_event(src.h_written[sp])
                    ln.desc[0].copy_(src.h_desc[sp], non_blocking=True)
                    ln.n[0:1].copy_(src.h_n[sp], non_blocking=True)
                    src.h_consumed[sp] = torch.cuda.Event()
                    src.h_consumed[sp].record(s)
                if timed:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(s)
                capi.match_top2_batch_device(ln.desc[1].data_ptr(), ln.n[1:].data_ptr(), ln.desc[0].data_ptr(), ln.n.data_ptr(),
                                             b, cap, ln.match[0].data_ptr(), ln.match[1].data_ptr(), ln.match[2].data_ptr(), s.cuda_stream)
                if timed:
                    e1.record(s)
                    self.match_events.append((e0, e1))
        self.steps_done += 1

    # ---- results of the last step (call after torch.cuda.synchronize())
    def counts(self):
        return torch.cat([ln.n[1:] for ln in self.lanes])

    def keypoints(self):
        return torch.cat([ln.kps for ln in self.lanes])

    def descriptors(self):
        return torch.cat([ln.desc[1:] for ln in self.lanes])

    def matches(self):
        """(3, B, cap): train index in the previous frame, best and second-best distance per query slot"""
        return torch.cat([ln.match for ln in self.lanes], dim=1)

    def status(self):
        return torch.cat([ln.status for ln in self.lanes])

    # ---- per-kernel timing (HIP events between the kernels of every lane; costs a few percent of throughput)
    def stage_timing(self, mode):
        for ln in self.lanes:
            ln.ex.stage_timing(mode)
        if mode == 2:
            self.match_events = []

    def stage_times(self):
        """{stage: (total ms, launches)} summed over the lanes; one launch = one lane's slice of b frames"""
        stage = {}
        for ln in self.lanes:
            for k, (ms, n) in ln.ex.stage_times().items():
                t = stage.get(k, (0.0, 0))
                stage[k] = (t[0] + ms, t[1] + n)
        if self.do_match and self.match_events:
            stage["match"] = (sum(e0.elapsed_time(e1) for e0, e1 in self.match_events), len(self.match_events))
        return stage

    def close(self):
        torch.cuda.synchronize(self.device)
        for ln in self.lanes:
            ln.ex.close()
        for p in [q for raw in self._raw_sets for q in raw] + self._spacers:
            capi.stream_destroy(self.device, p)
        self._raw_sets, self._spacers = [], []
