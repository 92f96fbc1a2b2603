#!/usr/bin/env python3
"""Turn a rocprofv3 results .db (rocpd sqlite, --kernel-trace --stats) into the text summary we commit under profiles/."""
import sqlite3, sys
db, out = sys.argv[1], sys.argv[2]
c = sqlite3.connect(db)
rows = list(c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"))
with open(out, "w") as f:
    f.write("# rocprofv3 --kernel-trace --stats summary (durations in us)\n")
    if len(sys.argv) > 3:
        f.write("# command: %s\n" % sys.argv[3])
    f.write("%-70s %8s %14s %12s %8s\n" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in rows:
        if pct < 0.005:
            continue
        f.write("%-70s %8d %14.1f %12.2f %8.2f\n" % (name[:70], calls, tot, avg, pct))
print(open(out).read())
