#!/bin/bash
# frames per step (4 lanes of B/4 frames): does a working set that fits the 256 MiB Infinity Cache (B = 256: 4 x 61 MB of pyramid) pay?  two rounds
mkdir -p gpurun_out/bs
for r in 1 2; do for B in 256 512 1024 2048; do
python bench.py --batch $B --ring $((B*2 > 2048 ? B*2 : 2048)) --no-also --no-cpu-baseline --min-seconds 2 --parity none --live-traffic off --detail-file gpurun_out/bs/x.json > /dev/null 2>gpurun_out/bs/err.txt
python -c "
import json; d=json.load(open('gpurun_out/bs/x.json')); print('B=$B r$r: %.1f frames/s  step %.4f ms  (%.4f ms per 1024 frames)  stages %s' % (d['value'], d['ms_per_step'], d['ms_per_step']*1024/$B, {k: round(v*1024/$B,3) for k,v in d['stage_ms_per_step'].items()}))"
done; done
