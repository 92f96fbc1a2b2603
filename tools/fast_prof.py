#!/usr/bin/env python3
"""Where a k_fast_cells wave spends its life: per-phase s_memtime deltas summed over all waves, read from an INSTRUMENTED build of the
library (tools/build_prof_variant.py -> build_variants/prof | profd; NOTES.md 9.6; not part of the product).
usage: ORBX_LIB=build_variants/prof/liborbx.so python tools/fast_prof.py [family] [w h] [nframes]"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from orb_slam_amd import capi, synth
fam = int(sys.argv[1]) if len(sys.argv) > 1 else 1
w, h = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
B = int(sys.argv[4]) if len(sys.argv) > 4 else 256
frames = synth.frames(w, h, fam, 0, B, threads=8)
ex = capi.ORBextractor(nfeatures=1000 if w <= 1024 else 2000, max_batch=B)
cap = ex.max_keypoints
d_img = torch.from_numpy(frames).cuda()
d_kps = torch.zeros((B, cap, 7), dtype=torch.float32, device="cuda"); d_desc = torch.zeros((B, cap, 32), dtype=torch.uint8, device="cuda")
d_n = torch.zeros(B, dtype=torch.int32, device="cuda")
L = capi.lib()
L.orbx_debug_fast_prof.argtypes = [ctypes.c_void_p, ctypes.c_int]
def run():
    ex.extract_batch_device(d_img.data_ptr(), B, w, h, w, w * h, d_kps.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, 0, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
run(); run()
SLOTS = 1 << 19
L.orbx_debug_fast_prof(None, 1)
run()
out = np.zeros(SLOTS * 12, dtype=np.uint32)
L.orbx_debug_fast_prof(out.ctypes.data, 0)
v = out.reshape(SLOTS, 12)
v = v[v[:, 10] == 1].astype(np.float64)
names = ["entry->DMA issued", "DMA wait (vmcnt 0)", "barrier after staging", "dense rounds", "drain (expand+score)", "barrier before NMS", "NMS", "barrier after NMS",
         "list output"]
if os.environ.get("PROF_KERNEL") == "describe":
    names = ["entry->tables built", "scalar loads landed + window DMA issue", "tables' barrier (behind the DMA issue since round 5)", "IC_Angle (patch loads + sums)", "atan2 + sincos", "window DMA wait", "16 BRIEF tests",
             "bit transpose", "-"]
def show(tag, r):
    m = r[:, :9].mean(0); tot = m.sum()
    print("%s: %d waves, mean wave life %.0f ticks" % (tag, len(r), tot))
    for i, nme in enumerate(names): print("  %-24s %8.0f ticks/wave  %5.1f %%" % (nme, m[i], 100 * m[i] / tot))
show("family %d %dx%d, %d frames, all levels" % (fam, w, h, B), v)
for lv in (0, 3, 7): show("  level %d" % lv, v[v[:, 11] == lv])
