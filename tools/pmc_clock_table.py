#!/usr/bin/env python3
"""Shader clock per kernel from tools/run_pmc_clock.sh: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the dispatch's duration in the kernel
trace of the same run, and the VALU issue rate at THAT clock: SQ_INSTS_VALU x (cycles per instruction of the kernel's opcode mix,
profiles/valu_mix.json) / (1024 SIMDs x GRBM cycles per XCD).  bench.py's `roofline.valu_issue` prices the same instructions at the
nominal 2.4 GHz.  usage: pmc_clock_table.py <dir with vga_* / hd_* csv files>"""
import collections, csv, glob, json, os, re, sys
XCDS, SIMDS = 8, 1024
d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
try:
    mix = json.load(open(os.path.join(root, "profiles", "valu_mix.json")))["kernels"]
except Exception:
    mix = {}
KEY = {"k_fast_cells": "fast_cells", "k_blur": "blur", "k_describe": "describe", "k_describe_od": "describe", "k_resize": "pyramid", "k_match_batch": "match", "k_cell_select": "cell_select",
       "k_level_select": "level_select", "k_quota": "quota"}
for tag in ("vga", "hd"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "**", "%s_counter_collection.csv" % tag), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "").strip()
            if n.startswith("orbx::"):
                acc[n][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(os.path.join(d, "**", "%s_kernel_trace.csv" % tag), recursive=True):
        for r in csv.DictReader(open(f)):
            n = re.sub(r"[<(].*", "", r["Kernel_Name"]).replace("void ", "").strip()
            if n.startswith("orbx::"):
                dur[n].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9)
    print("%s stream, one launch = one step of the serial command" % tag)
    print("  %-26s %5s %10s %14s %10s %14s %22s" % ("kernel", "disp", "ms", "GRBM/XCD cyc", "clock GHz", "VALU insts", "VALU issue at that clock"))
    for n, c in sorted(acc.items(), key=lambda kv: -sum(dur[kv[0]])):
        m = {k: sum(v) / len(v) for k, v in c.items()}
        t = sum(dur[n]) / max(len(dur[n]), 1)
        act = m.get("GRBM_GUI_ACTIVE", 0.0) / XCDS
        key = next((v for k, v in KEY.items() if k in n), None)
        cpi = None
        if key in mix:
            cpi = 0.5 * (mix[key]["cycles_per_inst_lo"] + mix[key]["cycles_per_inst_hi"])
        issue = "%.3f (%.2f cyc/inst)" % (m.get("SQ_INSTS_VALU", 0.0) * cpi / (SIMDS * act), cpi) if cpi and act else "-"
        print("  %-26s %5d %10.4f %14.4g %10.2f %14.4g %22s" % (n[-26:], len(dur[n]), t * 1e3, act, act / t * 1e-9 if t else 0.0, m.get("SQ_INSTS_VALU", 0.0), issue))
