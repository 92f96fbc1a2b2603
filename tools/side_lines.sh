#!/bin/bash
# The side lines of profiles/ (everything except the default bench line, the counter files and the GPU suite, which tools/ab_then_collect.sh or
# collect_round_artifacts.sh produce), most important first, so that a call cut short by the GPU budget still leaves the front of the list.
# usage: tools/side_lines.sh <name>   (-> gpurun_out/<name>/)
N=${1:?name}; R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$N; mkdir -p $D; cd $R
Q="--no-cpu-baseline --no-also --min-seconds 2"
timeout 200 python bench.py --config hd1080 --cpu-allcores-seconds 0 --cpu-seconds 5 --detail-file $D/bench_hd1080.json > $D/bench_hd1080.line.json 2>/dev/null
timeout 100 python bench.py --lanes 1 $Q --detail-file $D/bench_one_lane.json > $D/bench_one_lane.line.json 2>/dev/null
timeout 100 python bench.py --config vga_extract $Q --detail-file $D/bench_extract_only.json > $D/bench_extract_only.line.json 2>/dev/null
timeout 100 python bench.py --region-timing $Q --detail-file $D/bench_region_timing.json > $D/bench_region_timing.line.json 2>/dev/null
timeout 100 python bench.py --family 0 $Q --detail-file $D/bench_noise.json > $D/bench_noise.line.json 2>/dev/null
tools/run_pmc_clock.sh ${N}_clock > /dev/null 2>&1; cp gpurun_out/${N}_clock/pmc_clock.txt $D/pmc_clock.txt
tools/run_pmc_match.sh ${N}_mfma > /dev/null 2>&1; cp gpurun_out/${N}_mfma/pmc_mfma.txt $D/pmc_mfma.txt
(cd /tmp; export TMPDIR=/tmp
 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats_overlap -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --min-seconds 0 --no-also --no-parity > $D/stats_overlap.log 2>&1
 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats_match -- python $R/bench.py --config match100k --steps 40 --warmup 5 --no-cpu-baseline --min-seconds 0 --no-parity > $D/stats_match.log 2>&1)
timeout 300 python bench.py --gpus 2 --backend gloo --share-device --no-cpu-baseline --batch 512 --min-seconds 2 --also-min-seconds 1 2>/dev/null | grep "^{" > $D/bench_two_ranks_one_gpu_gloo.json
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-device --no-cpu-baseline --batch 512 --min-seconds 1 --also-min-seconds 0.5 2>/dev/null | grep "^{" > $D/bench_torchrun_two_ranks.json
python - <<PY
import sys
sys.path.insert(0, "$R")
from orb_slam_amd import synth
synth.frames(640, 480, synth.BLOCKS, 0, 2048).tofile("/tmp/frames.raw")
PY
(orb_slam_amd/cpp/example_lanes 640 480 1024 2 4 /tmp/frames.raw "" 200; orb_slam_amd/cpp/example_lanes 640 480 1024 2 1 /tmp/frames.raw "" 200) > $D/cpp_example_lanes.txt 2>&1
(timeout 100 python tools/bench_single_frame.py; timeout 100 python tools/bench_single_frame.py 1920 1080 2000; timeout 100 orb_slam_amd/cpp/bench_single_frame; timeout 100 orb_slam_amd/cpp/bench_single_frame 1920 1080 2000) > $D/single_frame.txt 2>/dev/null
timeout 100 python tools/bench_pcie.py --steps 30 > $D/pcie_inclusive.json 2>/dev/null
timeout 200 python tools/bench_frontend.py --window 15 2>/dev/null | tail -1 > $D/frontend_w15.json
timeout 200 python tools/bench_frontend.py 2>/dev/null | tail -1 > $D/frontend_w100.json
ORBX_MATCH_MFMA=0 timeout 100 python bench.py --config match100k --no-cpu-baseline --min-seconds 2 --detail-file $D/bench_match100k_popcount.json > $D/bench_match100k_popcount.line.json 2>/dev/null
timeout 100 python tools/corun_probe.py > $D/corun_probe.json 2>/dev/null
timeout 100 python tools/bench_kf_search.py 2>/dev/null | tail -1 > $D/kf_search.json
ls $D | wc -l
