"""Deterministic inputs of the front-end golden fixtures: everything derives from the committed golden keypoints /
descriptors of vga_blocks_f0 by exact integer / float32 arithmetic (no random generator)."""
import os

import numpy as np

from orb_slam_amd import capi

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def frontend_inputs():
    z = np.load(os.path.join(GOLD, "vga_blocks_f0.npz"))
    kps = z["kps"].copy()
    desc = z["desc"].copy()
    n = len(kps)
    i = np.arange(n)
    cam = capi.Camera.make(517.3, 516.5, 318.6, 255.3, (0.2624, -0.9531, -0.0054, 0.0026), 640, 480)
    # queries: keypoint (i*7+3) mod n, displaced by a few pixels, a few descriptor bits flipped, angle shifted
    src = (i * 7 + 3) % n
    qxyr = np.zeros((n, 3), np.float32)
    qxyr[:, 0] = kps["x"][src] + ((i % 5) - 2).astype(np.float32)
    qxyr[:, 1] = kps["y"][src] + ((i % 3) - 1).astype(np.float32)
    qxyr[:, 2] = np.float32(12.0) * (np.float32(1.2) ** kps["octave"][src].astype(np.float32))
    lv = kps["octave"][src]
    qlev = np.stack([lv - 1, lv + 1], -1).astype(np.int32)
    qdesc = desc[src].copy()
    qdesc[i, i % 32] ^= (1 << (i % 8)).astype(np.uint8)
    qdesc[i, (i * 3) % 32] ^= (1 << ((i // 3) % 8)).astype(np.uint8)
    qdesc[::6] = desc[(i[::6] * 11 + 1) % n]                       # some queries carry a foreign descriptor
    qangle = ((kps["angle"][src] + (i % 40).astype(np.float32)) % np.float32(360)).astype(np.float32)
    qvalid = (i % 10 != 9).astype(np.uint8)
    claimed = (i % 6 == 0).astype(np.uint8)
    rules = [(0, 100, 0.8, False), (1, 100, 0.8, True), (2, 100, 0.9, True), (3, 50, 0.9, True)]
    seg_off = np.array([0, 1, 3, 6, 10, 17, 30, 64, 130, 131, 131, 260], np.int32)
    return dict(kps=kps, desc=desc, cam=cam, qxyr=qxyr, qlev=qlev, qdesc=qdesc, qangle=qangle, qvalid=qvalid, claimed=claimed, rules=rules, seg_off=seg_off)
