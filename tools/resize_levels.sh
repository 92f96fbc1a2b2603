#!/bin/bash
# per-level durations of k_resize (and every other kernel) from a kernel trace of the serial command
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$1; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
ORBX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --output-format csv -d $D -o kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --lanes 1 --region-timing --min-seconds 0 --no-also --no-parity > $D/kt.log 2>&1
python - <<PY
import csv, collections, glob
f = glob.glob("$D/**/kt_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n = r["Kernel_Name"].split("(")[0]
    d[(n, r["Grid_Size"] if "Grid_Size" in r else r.get("Grid_Size_X"), r.get("Workgroup_Size", r.get("Workgroup_Size_X")))].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    v = sorted(v)
    print("%-60s grid %-10s wg %-5s n %3d  median %8.1f us  min %8.1f" % (k[0][:60], k[1], k[2], len(v), v[len(v) // 2], v[0]))
PY
