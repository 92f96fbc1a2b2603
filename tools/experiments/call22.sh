cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c22; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$name.json 2>$O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d['config']['parity_mismatches'])"; }
run base A=1
run nooverlap ORBX_OVERLAP=0
run noaffinity ORBX_XCD_AFFINITY=0
run base2 A=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
