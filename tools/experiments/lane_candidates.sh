#!/bin/bash
# more stream-placement candidates than the three the lane pipeline probes (ORBX_LANE_CANDIDATES=k): the probe timings of k spacer rotations, three fresh processes
mkdir -p gpurun_out/cand
for r in 1 2 3; do
ORBX_LANE_CANDIDATES=6 python bench.py --no-also --no-cpu-baseline --min-seconds 1.5 --parity none --live-traffic off --detail-file gpurun_out/cand/x.json > /dev/null 2>gpurun_out/cand/err.txt
python -c "
import json; d=json.load(open('gpurun_out/cand/x.json')); print('run $r: %.1f frames/s  step %.4f ms  probe %s chosen %s' % (d['value'], d['ms_per_step'], d['config']['lane_placement'].get('probe_ms_per_step'), d['config']['lane_placement'].get('chosen')))"
done
