cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c16; mkdir -p $O
for bb in "512 2048" "1024 2048" "2048 4096" "4096 8192"; do
  set -- $bb
  timeout 400 python bench.py --no-cpu-baseline --no-also --min-seconds 2 --parity sample --batch $1 --ring $2 > $O/batch_$1.json 2>$O/batch_$1.err
  python -c "
import json; d=json.load(open('$O/batch_$1.json')); print('batch $1', d['value'], d['ms_per_step'], d['config']['lanes'], d['config']['parity_mismatches'])"
done
