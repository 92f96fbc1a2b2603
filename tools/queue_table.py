#!/usr/bin/env python3
"""(Queue_Id, Stream_Id) -> kernels table from a rocprofv3 kernel trace CSV: how the HIP runtime placed the streams on its hardware queues."""
import csv, collections, re, sys
by = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(sys.argv[1])):
    m = re.search(r"(k_[a-z_]+)", r["Kernel_Name"])
    if m:
        by[(int(r["Queue_Id"]), int(r["Stream_Id"]))][m.group(1)] += 1
for k, v in sorted(by.items()):
    print("queue %d stream %d: %s" % (k[0], k[1], "blur only" if set(v) == {"k_blur"} else ("lane" + (" + blur" if "k_blur" in v else "")) + " (%d launches)" % sum(v.values())))
