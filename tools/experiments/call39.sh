#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab39 c4:0 tree:0 c4:1 tree:1 c4:1:hd1080 tree:1:hd1080 c4:4 tree:4
Q="--no-cpu-baseline --no-also --min-seconds 2"
for v in c4 tree; do
  lib=$PWD/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$PWD/orb_slam_amd/liborbx.so
  echo "$v noise 4 lanes: $(ORBX_LIB=$lib timeout 200 python bench.py $Q --family 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["parity_mismatches"])')"
done
