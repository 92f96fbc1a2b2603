set -x
cd /root/repo
for od in 0 1; do
  for cfg in vga hd1080; do
    ORBX_BLUR_ON_DEMAND=$od python bench.py --config $cfg --no-also --no-cpu-baseline --live-traffic off --min-seconds 4 --parity all --detail-file gpurun_out/ab_od_${cfg}_$od.json 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline())
print('RESULT od=$od cfg=$cfg value', d['value'], 'ms', d['ms_per_step'], 'stages', d.get('stage_ms_per_step'), 'parity', d['config'].get('parity_checked_frames'), d['config'].get('parity_mismatches'))
"
  done
done
