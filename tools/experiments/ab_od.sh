#!/bin/bash
# Round 6 A/B: the blur per keypoint window (k_describe_od, ORBX_BLUR_ON_DEMAND=1, the default) against the blur kernels + k_describe (=0), the
# headline configuration and the 1080p one, every frame of the last step against the oracle.  usage (on the GPU box): bash tools/experiments/ab_od.sh [modes]
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
mkdir -p gpurun_out
for od in ${1:-0 1}; do
  for cfg in vga hd1080; do
    ORBX_BLUR_ON_DEMAND=$od python bench.py --config $cfg --no-also --no-cpu-baseline --live-traffic off --min-seconds 4 --parity all --detail-file gpurun_out/ab_od_${cfg}_$od.json 2>&1 | tail -1 | OD=$od CFG=$cfg python -c "
import sys, json, os
d = json.loads(sys.stdin.readline())
print('RESULT od=%s cfg=%s value' % (os.environ['OD'], os.environ['CFG']), d['value'], 'ms', d['ms_per_step'], 'stages', d.get('stage_ms_per_step'), 'parity', d['config'].get('parity_checked_frames'), d['config'].get('parity_mismatches'))
"
  done
done
