#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
Q="--no-cpu-baseline --no-also --min-seconds 2.5 --parity none"
for v in c5 tree c5 tree; do
  lib=$PWD/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$PWD/orb_slam_amd/liborbx.so
  echo "$v: $(ORBX_LIB=$lib timeout 100 python bench.py $Q 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
