#!/bin/bash
# the hint written by the first frame of a launch group only (scratch build build_variants/hintf0): four-lane step against the shipped build
cd ${GRAFT_REPO_ROOT:-/root/repo}
Q="--no-cpu-baseline --no-also --min-seconds 2 --parity none"
for item in c5:1 hintf0:1 c5:1 hintf0:1 c5:3 hintf0:3; do
  v=${item%%:*}; f=${item#*:}
  echo "$v fam $f: $(ORBX_LIB=$PWD/build_variants/$v/liborbx.so timeout 60 python bench.py $Q --family $f 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
ORBX_LIB=$PWD/build_variants/hintf0/liborbx.so timeout 60 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "family or stage or batch" 2>&1 | tail -1
