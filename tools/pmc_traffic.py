#!/usr/bin/env python3
"""profiles/traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate passes as the TCC
block cannot hold both).  Per-launch HBM-side bytes per kernel = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (the counters are
in KiB).  usage: pmc_traffic.py fetch.csv write.csv out.json batch [sq_dir [sq_batch [workload]]]  gfx950 correction (MI355X_MICROARCH.md §HBM): FETCH_SIZE reports half the bytes of a streaming read.  Calibrated
here for the access widths of these kernels (tools/microbench/fetch_calib, profiles/r02_fetch_calibration.txt): coalesced
1, 4 and 16 B-per-lane reads of 1 GiB all report exactly 0.500; WRITE_SIZE is exact (1.000); 32-byte runs 640 B apart (a
keypoint patch row) report the 64-byte lines they touch (2.0 x the unique bytes).  So reads are doubled for every kernel;
for the gather-like loads of k_describe that is an upper bound.  Raw counters are kept in `detail`."""
import csv, json, os, re, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam_amd import capi
fetch_csv, write_csv, out, batch = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
STAGE = {"k_resize": "pyramid", "k_fast_cells": "fast_cells", "k_quota": "quota", "k_cell_select": "cell_select",
         "k_level_select": "level_select", "k_blur": "blur", "k_blur_mfma": "blur", "k_describe": "describe", "k_describe_od": "describe", "k_match_batch": "match", "k_match_batch_mfma": "match"}
# (a workload runs ONE of k_blur / k_blur_mfma: VGA-class levels take the matrix-core form, wider ones the VALU form)
def load(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter: continue
        m = re.search(r"orbx::(k_[a-z_]+)", r["Kernel_Name"])
        if m: acc[m.group(1)].append(float(r["Counter_Value"]))
    return acc
f, w = load(fetch_csv, "FETCH_SIZE"), load(write_csv, "WRITE_SIZE")
per_launch, detail = {}, {}
for k, stage in STAGE.items():
    if k not in f: continue
    n_per_step = 7 if k == "k_resize" else 1          # the pyramid stage is 7 launches per step
    fb = sum(f[k]) / len(f[k]) * 1024 * n_per_step
    wb = sum(w[k]) / len(w[k]) * 1024 * n_per_step if k in w else 0.0
    per_launch[stage] = int(2 * fb + wb)
    detail[stage] = {"fetch_bytes_raw": int(fb), "fetch_bytes_corrected": int(2 * fb), "write_bytes": int(wb), "dispatches_sampled": len(f[k])}
# optional: VALU / LDS activity from the SQ counter passes of tools/run_pmc.sh (argv[5] = directory with a_*/b_* CSVs)
valu = {}
if len(sys.argv) > 5:
    d = sys.argv[5]
    busy = load(d + "/b_counter_collection.csv", "SQ_ACTIVE_INST_VALU")
    lds = load(d + "/b_counter_collection.csv", "SQ_LDS_IDX_ACTIVE")
    insts = load(d + "/a_counter_collection.csv", "SQ_INSTS_VALU")
    dur = collections.defaultdict(list)
    for r in csv.DictReader(open(d + "/b_kernel_trace.csv")):
        m = re.search(r"orbx::(k_[a-z_]+)", r["Kernel_Name"])
        if m: dur[m.group(1)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, stage in STAGE.items():
        if k in busy and k in dur:
            cyc = sum(dur[k]) / len(dur[k]) * 1e-9 * 2.4e9
            valu[stage] = {"valu_busy": round(sum(busy[k]) / len(busy[k]) * 4 / (cyc * 1024), 3),       # quad-cycles -> SIMD cycles, 1024 SIMDs
                           "lds_busy": round(sum(lds[k]) / len(lds[k]) / (cyc * 256), 3) if k in lds else None,
                           "valu_wave_insts_per_launch": int(sum(insts[k]) / len(insts[k])) if k in insts else None}
# wave-level VALU instructions per FRAME (all dispatches of a kernel in the SQ pass / the frames that pass processed): the
# instruction-issue roofline of bench.py (`roofline_valu`): 1024 SIMDs x 2.4 GHz / 4 cycles per wave64 op = 6.144e11 wave-insts/s
per_frame = {}
if len(sys.argv) > 5:
    sq_batch = int(sys.argv[6]) if len(sys.argv) > 6 else 256
    frames = len(insts.get("k_fast_cells", [])) * sq_batch
    if frames:
        per_frame = {STAGE[k]: round(sum(v) / frames, 1) for k, v in insts.items() if k in STAGE}
json.dump({"workload": sys.argv[7] if len(sys.argv) > 7 else "vga_640x480_nf1000", "src_hash": capi.build_id(), "batch": batch, "sq_activity": valu, "valu_wave_insts_per_frame": per_frame,
           "per_launch_bytes": per_launch, "detail": detail,
           "read_correction": "FETCH_SIZE x 2 (gfx950: streaming reads of 1, 4 and 16 B per lane all report exactly half, tools/microbench/fetch_calib); WRITE_SIZE as is",
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) on bench.py"}, open(out, "w"), indent=1)
print(json.dumps(per_launch))
