mkdir -p gpurun_out/prio
for r in 1 2; do for p in "" "-1,-1,1,1" "-1,0,0,1" "-1,1,-1,1" "-1,-1,-1,-1"; do
ORBX_LANE_PRIORITIES="$p" python bench.py --no-also --no-cpu-baseline --min-seconds 2 --parity none --live-traffic off --detail-file gpurun_out/prio/x.json > /dev/null 2>gpurun_out/prio/err.txt
python -c "
import json; d=json.load(open('gpurun_out/prio/x.json')); print('prio [%s] r$r: %.1f frames/s  step %.4f ms  placement %s' % ('$p', d['value'], d['ms_per_step'], d['config']['lane_placement'].get('probe_ms_per_step')))"
done; done
