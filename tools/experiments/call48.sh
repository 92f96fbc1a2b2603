#!/bin/bash
# the fallback hint with a run length of six needs more than exp_ab's 2 warm-up steps to settle: 8 warm-up + 30 timed steps per run
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; D=$R/gpurun_out/ab48; mkdir -p $D
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
S="--lanes 1 --steps 30 --warmup 8 --no-cpu-baseline --region-timing --min-seconds 0 --no-also --no-parity"
for item in c5:1 tree:1 c5:3 tree:3 c5:4 tree:4 c5:1 tree:1; do
  v=${item%%:*}; f=${item#*:}
  lib=$R/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$R/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py $S --family $f > $D/${v}_f$f.json 2>/dev/null
  python -c "
import json; d=json.load(open('$D/${v}_f$f.json')); print('%-5s fam $f step %.4f fast %.4f' % ('$v', d['ms_per_step'], d['stage_ms_per_step']['fast_cells']))"
done
