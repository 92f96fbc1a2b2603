#!/bin/bash
# Everything a round ends with, in ONE call on the GPU box (≈12 GPU-minutes): the artifacts of profiles/ (collect_round_artifacts.sh,
# run_pmc_hd.sh), the whole GPU test suite, the torchrun two-rank line, the C++ lanes example, a front-end fuzz slice and the default line.
# usage: tools/round_end.sh <name>   (-> gpurun_out/<name>/, gpurun_out/<name>hd/)
N=${1:?name}
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tools/collect_round_artifacts.sh $N 2>&1 | tail -3 | cut -c1-300
tools/run_pmc_hd.sh ${N}hd 2>&1 | tail -1 | cut -c1-200
tools/run_pmc_clock.sh ${N}clk 2>&1 | tail -12 | cut -c1-200          # the clock the hot kernels actually run at
( O=$R/gpurun_out/${N}ta; mkdir -p $O; cd /tmp; export TMPDIR=/tmp ORBX_OVERLAP=0      # texture-addresser busy per kernel (one --pmc pass, kernel trace only)
  timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o c -- python $R/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity > $O/c.log 2>&1
  python $R/tools/pmc_table.py $O/c_counter_collection.csv > $O/pmc_ta.txt 2>&1; tail -12 $O/pmc_ta.txt | cut -c1-200 )
cd $R; (time python -m pytest tests -m gpu -q) > gpurun_out/$N/pytest_gpu.txt 2>&1; grep "passed\|failed" gpurun_out/$N/pytest_gpu.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-device --no-cpu-baseline --batch 512 --min-seconds 1 --also-min-seconds 0.5 2>/dev/null | grep "^{" > gpurun_out/$N/bench_torchrun_two_ranks.json
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from orb_slam_amd import synth
synth.frames(640, 480, synth.BLOCKS, 0, 2048).tofile("/tmp/frames.raw")
PY
(orb_slam_amd/cpp/example_lanes 640 480 1024 2 4 /tmp/frames.raw "" 200; orb_slam_amd/cpp/example_lanes 640 480 1024 2 1 /tmp/frames.raw "" 200) > gpurun_out/$N/cpp_example_lanes.txt 2>&1; grep "frames/s\|IDENT" gpurun_out/$N/cpp_example_lanes.txt
timeout 600 python tools/fuzz_batch.py 400 5102 > gpurun_out/$N/fuzz_batch_400.json 2>/dev/null; tail -c 200 gpurun_out/$N/fuzz_batch_400.json; timeout 600 python tools/fuzz_parity.py 4000 5101 > gpurun_out/$N/fuzz_parity_4000.json 2>/dev/null; tail -c 200 gpurun_out/$N/fuzz_parity_4000.json; timeout 600 python tools/fuzz_frontend.py 3000 5104 > gpurun_out/$N/fuzz_frontend_3000.json 2>/dev/null; tail -c 200 gpurun_out/$N/fuzz_frontend_3000.json; timeout 600 python tools/fuzz_match.py 20000 5105 > gpurun_out/$N/fuzz_match_20000.json 2>/dev/null; tail -c 300 gpurun_out/$N/fuzz_match_20000.json; timeout 600 python tools/fuzz_orbmatcher.py 10000 5106 > gpurun_out/$N/fuzz_orbmatcher_10000.json 2>/dev/null; tail -c 300 gpurun_out/$N/fuzz_orbmatcher_10000.json; timeout 600 python tools/fuzz_orbmatcher.py 10000 5107 real > gpurun_out/$N/fuzz_orbmatcher_real_access_10000.json 2>/dev/null; tail -c 200 gpurun_out/$N/fuzz_orbmatcher_real_access_10000.json
tools/run_pmc_match.sh ${N}mfma 2>&1 | tail -6 | cut -c1-400          # matrix-pipe counters of the matcher (FP4 kernels)
timeout 900 python bench.py --detail-file gpurun_out/$N/bench_final.json > gpurun_out/$N/bench_final.stdout 2> gpurun_out/$N/bench_final.err || tail -5 gpurun_out/$N/bench_final.err; python -c "
import json; d=json.load(open('gpurun_out/$N/bench_final.json')); print(d['value'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['parity_mismatches'], {k:(v['value'], v['roofline'].get('frac'), v['roofline'].get('traffic'), v['config']['parity_mismatches']) for k,v in d['also'].items()})"
