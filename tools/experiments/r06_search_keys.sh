#!/bin/bash
# Round 6: the length of a query's speculative key list in k_window_search (4 / 6 / 8): parity (search tests + matcher fuzz), kernel times of the one-problem calls,
# commit rounds and rescans (probe builds r6 / r8), and the batched front-end searches.   build_variants/{k4,k6,k8,r6,r8}
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
for v in $@; do
  LP=$PWD/build_variants/$v
  echo "=================== $v"
  case $v in
  r*) LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH DROPIN_PRINT_RETURNS=1 python tools/bench_orbmatcher_dropin.py 2>&1 | grep returns | tr '\n' ' '; echo;;
  *)
    LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH ORBX_LIB=$LP/liborbx.so python -m pytest tests/test_gpu_search.py tests/test_gpu_search_kf.py tests/test_gpu_orbmatcher_dropin.py -x -q 2>&1 | tail -2
    LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH python tools/fuzz_orbmatcher.py 6000 9106 | cut -c1-160
    D=gpurun_out/search_keys/$v; rm -rf $D; mkdir -p $D
    LD_LIBRARY_PATH=$LP:$LD_LIBRARY_PATH rocprofv3 --kernel-trace --output-format csv -d $D -- python tools/bench_orbmatcher_dropin.py > $D/bench.txt 2>&1
    python - $D <<'P'
import csv,glob,sys
k=glob.glob(sys.argv[1]+'/**/*kernel_trace.csv',recursive=True)[0]
rows=sorted(csv.DictReader(open(k)),key=lambda r:int(r['Start_Timestamp']))
seq=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'k_window_search' in r['Kernel_Name']]
names=['mappoints','window','init','last_frame','SIM3','two_frames','keyframe','scw','fuse','bow','bow_kf','tri']
out={}; i=0
for n in names:
    if n=='SIM3':
        s=seq[i:i+14]; out['sim3']=min(s[0::2])+min(s[1::2]); i+=14; continue
    out[n]=min(seq[i:i+7]); i+=7
print({k:round(x,1) for k,x in out.items()})
P
    grep "ms   product" $D/bench.txt | awk '{printf "%s %s/%s  ", $1, $3, $6}'; echo
    ORBX_LIB=$LP/liborbx.so python tools/bench_search.py 2>&1 | tail -4 | cut -c1-300
    ;;
  esac
done
