#!/usr/bin/env python3
"""Per-kernel table of the matrix-pipe counters tools/run_pmc_match.sh collects (mean per dispatch) with what follows from them:
  MfmaUtil    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 1024 SIMDs)   — rocprofiler-sdk's derived metric (GRBM_GUI_ACTIVE arrives summed
                over the 8 XCDs of the MI355X; its definition takes the maximum over them)
  clock       = GRBM_GUI_ACTIVE per XCD / the dispatch's duration in the kernel trace of the same run
  frac of the 2.4 GHz peak = MfmaUtil x clock / 2.4 GHz                             — what bench.py's time-based `roofline.frac` must agree with
usage: pmc_mfma_table.py <dir with m100k_* and batch_* csv files>"""
import collections, csv, glob, os, re, sys
XCDS, SIMDS = 8, 1024
d = sys.argv[1]
for tag in ("m100k", "batch"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "%s_counter_collection.csv" % tag), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"])
            if "k_match" in n:
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "%s_kernel_trace.csv" % tag), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"\(.*", "", r["Kernel_Name"])
            if "k_match" in n:
                dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    for n, c in sorted(acc.items()):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        fp4 = m.get("SQ_INSTS_VALU_MFMA_F6F4", 0.0) > 0          # the FP4 kernels (v_mfma_scale_f32_32x32x64_f8f6f4: twice the K, twice the peak)
        mfma, busy = m.get("SQ_INSTS_VALU_MFMA_F6F4" if fp4 else "SQ_INSTS_VALU_MFMA_I8", 0.0), m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        if not mfma:
            continue
        active = m.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        util = busy / max(active * SIMDS, 1.0)
        t = sum(dur[n]) / max(len(dur[n]), 1)
        clk = active / t if t else 0.0
        print("%s (%s): %d dispatches, %.4f ms each under the counters" % (n.strip(), tag, max(len(v) for v in c.values()), t * 1e3))
        print("    " + "  ".join("%s %.4g" % (k, v) for k, v in sorted(m.items())))
        frac = busy / (SIMDS * t * 2.4e9) if t else 0.0           # needs no GRBM counter: busy cycles against the dispatch's own duration
        if clk > 2.5e9:       # a short dispatch: GRBM_GUI_ACTIVE also covers the counter start / stop around it, so the split is not meaningful
            split = "GRBM_GUI_ACTIVE spans more than this short dispatch (no MfmaUtil / clock split)"
        else:
            split = "MfmaUtil %.3f at a shader clock of %.2f GHz" % (util, clk * 1e-9)
        print("    MFMA busy cycles per %s: %.1f;  %s;  busy cycles / (1024 SIMDs x duration x 2.4 GHz) = %.3f of the %s peak;"
              "  VALU instructions per MFMA (all classes, the MFMAs included) %.2f;  VALU busy (SQ_ACTIVE_INST_VALU x 4 / SIMD cycles at 2.4 GHz) %.3f" %
              ("v_mfma_scale_f32_32x32x64_f8f6f4 (FP4)" if fp4 else "v_mfma_i32_32x32x32_i8", busy / mfma, split, frac, "FP4" if fp4 else "int8",
               m.get("SQ_INSTS_VALU", 0.0) / mfma, m.get("SQ_ACTIVE_INST_VALU", 0.0) * 4 / (SIMDS * t * 2.4e9) if t else 0.0))
