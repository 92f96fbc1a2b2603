cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c9; mkdir -p $O
for l in 2 4 6 8; do
  timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample --lanes $l > $O/lanes_$l.json 2>$O/lanes_$l.err
  python -c "
import json; d=json.load(open('$O/lanes_$l.json')); print('lanes $l', d['value'], d['ms_per_step'], d['config']['lanes'], d['config']['host_submit_ms_per_step'], d['config']['lane_placement']['probe_ms_per_step'])"
done
for l in 4 8; do
  timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample --lanes $l > $O/hdlanes_$l.json 2>$O/hdlanes_$l.err
  python -c "
import json; d=json.load(open('$O/hdlanes_$l.json')); print('hd lanes $l', d['value'], d['ms_per_step'], d['config']['lanes'])"
done
