#!/usr/bin/env python3
"""profiles/clock.json from the table tools/pmc_clock_table.py prints (tools/run_pmc_clock.sh): the shader clock under every kernel of the
serial command, keyed by the library's source hash — bench.py prices its VALU-issue rooflines at that clock beside the nominal 2.4 GHz.
usage: pmc_clock_json.py pmc_clock.txt out.json   (on the GPU box: the hash comes from the loaded library)"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from orb_slam_amd import capi
out = {"src_hash": capi.build_id(), "source": "tools/run_pmc_clock.sh: GRBM_GUI_ACTIVE per XCD / dispatch duration, serial command", "vga": {}, "hd": {}}
sec = None
for ln in open(sys.argv[1]):
    if ln.startswith("vga stream"):
        sec = "vga"
    elif ln.startswith("hd stream"):
        sec = "hd"
    m = re.match(r"\s+orbx::(k_\w+)\s+\d+\s+[\d.]+\s+[\d.e+]+\s+([\d.]+)", ln)
    if m and sec:
        out[sec][m.group(1)] = float(m.group(2))
json.dump(out, open(sys.argv[2], "w"), indent=1)
