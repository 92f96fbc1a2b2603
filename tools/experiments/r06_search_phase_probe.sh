#!/bin/bash
# Round 6: where a ONE-problem search's kernel time goes — the kernel trace of tools/bench_orbmatcher_dropin.py with the product library and with two
# timing-probe builds of orbs_search.hip (build_variants/noscan: -DORBS_PROBE_NO_SCAN, build_variants/nocommit: -DORBS_PROBE_NO_COMMIT; results of those are wrong by design).
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in product noscan nocommit; do
  D=gpurun_out/search_probe/$v; rm -rf $D; mkdir -p $D
  if [ $v = product ]; then LP=; else LP=$PWD/build_variants/$v; fi
  LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH rocprofv3 --kernel-trace --output-format csv -d $D -- python tools/bench_orbmatcher_dropin.py > $D/bench.txt 2>&1
  echo "== $v"
  python - $D <<'P'
import csv,glob,sys
k=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(k)),key=lambda r:int(r['Start_Timestamp']))
seq=[(r['Kernel_Name'].split('(')[0].replace('void orbs::','').replace('orbs::',''),(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3) for r in rows if 'copyBuffer' not in r['Kernel_Name'] and 'fillBuffer' not in r['Kernel_Name']]
i=0
while i<len(seq):
    j=i
    while j<len(seq) and seq[j][0]==seq[i][0] and abs(seq[j][1]-seq[i][1])<0.25*seq[i][1]+5: j+=1
    d=[x[1] for x in seq[i:j]]
    print(f"{seq[i][0][:44]:44s} x{j-i:3d}  min {min(d):8.1f} us")
    i=j
P
done
