#!/usr/bin/env python3
"""Device timeline of ONE orbx_extract call from a rocprofv3 trace of orb_slam_amd/cpp/bench_single_frame:
  rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d <dir> -- orb_slam_amd/cpp/bench_single_frame 640 480 1000 60
  python tools/single_frame_timeline.py <dir>/<host>        (the directory holding *_kernel_trace.csv)
Prints start / end / duration / gap to the previous event (us) of every kernel and copy of the last-but-one call."""
import csv, glob, sys
d = sys.argv[1]
k = list(csv.DictReader(open(glob.glob(d + '/*kernel_trace.csv')[0])))
mf = glob.glob(d + '/*memory_copy_trace.csv')
m = list(csv.DictReader(open(mf[0]))) if mf else []
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:48] + '  grid ' + r['Grid_Size_X']) for r in k]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r['Direction']) for r in m]
ev.sort()
# a call ends with k_describe (and the D2H copy right behind it on the copy-engine path); it starts with the first event after that
calls, cur = [], []
for i, e in enumerate(ev):
    if cur and (('k_describe' in cur[-1][2] and 'DEVICE_TO_HOST' not in e[2]) or ('DEVICE_TO_HOST' in cur[-1][2] and len(cur) > 1 and 'k_describe' in cur[-2][2])):
        calls.append(cur); cur = []
    cur.append(e)
call = calls[-2]
t0, prev = call[0][0], None
print('%9s %9s %8s %8s  %s' % ('start', 'end', 'dur', 'gap', 'event (us)'))
for e in call:
    print('%9.1f %9.1f %8.1f %8.1f  %s' % ((e[0] - t0) / 1e3, (e[1] - t0) / 1e3, (e[1] - e[0]) / 1e3, (e[0] - prev) / 1e3 if prev else 0.0, e[2]))
    prev = e[1]
print('device span %.1f us; call period under the profiler %.1f us' % ((prev - t0) / 1e3, (calls[-1][0][0] - t0) / 1e3))
