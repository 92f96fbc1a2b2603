#!/bin/bash
# Round 6: lanes per step with the on-demand description kernel (no blur side stream any more).  usage (GPU box): bash tools/experiments/r06_lanes.sh
cd ${GRAFT_REPO_ROOT:-$(dirname $0)/../..}
for cfg in vga hd1080; do
  for lanes in 2 3 4 6 8; do
    python bench.py --config $cfg --lanes $lanes --no-also --no-cpu-baseline --live-traffic off --min-seconds 3 --parity sample 2>&1 | tail -1 | CFG=$cfg LANES=$lanes python -c "
import sys, json, os
d = json.loads(sys.stdin.readline())
print('RESULT', os.environ['CFG'], 'lanes', os.environ['LANES'], d['value'], d['ms_per_step'], d['config'].get('parity_mismatches'))
"
  done
done
