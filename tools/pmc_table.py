#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs into a per-kernel table (mean per dispatch)."""
import csv, sys, collections, re
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sys.argv[1:]:
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"])
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted(acc)
ctrs = sorted({c for n in names for c in acc[n]})
print("%-28s %6s " % ("kernel", "disp") + " ".join("%14s" % c[-14:] for c in ctrs))
for n in names:
    if not any(ns in n for ns in ("orbx::", "orbv::", "orbf::", "orbs::")):
        continue
    nd = max(len(v) for v in acc[n].values())
    print("%-28s %6d " % (n[-28:], nd) + " ".join("%14.4g" % (sum(acc[n][c]) / len(acc[n][c])) if acc[n][c] else "%14s" % "-" for c in ctrs))
