#!/usr/bin/env python3
"""Ratios counter / known bytes from the two rocprofv3 --pmc passes over tools/microbench/fetch_calib (FETCH_SIZE, WRITE_SIZE in KiB)."""
import csv, sys, re
GiB = 1 << 30
runs = GiB // 640
lines = sum(1 if ((640 * r + (r * 7 + 3) % 29) % 64) + 32 <= 64 else 2 for r in range(runs))
known = {"k_read4": GiB, "k_read16": GiB, "k_read1": GiB // 4, "k_patch": runs * 32, "k_write4": GiB}
for path, ctr in ((sys.argv[1], "FETCH_SIZE"), (sys.argv[2], "WRITE_SIZE")):
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != ctr:
            continue
        m = re.search(r"(k_[a-z0-9]+)", r["Kernel_Name"])
        if not m or m.group(1) not in known:
            continue
        k = m.group(1)
        v = float(r["Counter_Value"]) * 1024
        extra = "   (64-B lines touched: %d B -> ratio %.3f)" % (lines * 64, v / (lines * 64)) if k == "k_patch" and ctr == "FETCH_SIZE" else ""
        if (ctr == "WRITE_SIZE") == (k == "k_write4"):
            print("%-10s %-9s counter %13.0f B   known %13d B   ratio %.3f%s" % (ctr, k, v, known[k], v / known[k], extra))
