#!/bin/bash
# Everything profiles/ is built from, in one go on the GPU box.  usage: tools/collect_round_artifacts.sh <name> [lines]   (-> gpurun_out/<name>, gpurun_out/<name>_pmc)
# `lines` skips the counter passes (profiles/traffic.json and valu_mix.json must already carry this build's hash) and only re-runs the bench lines.
# Order: counter passes first (profiles/traffic.json of THIS build — it carries the library's source hash — is what bench.py replays for the
# VALU-issue roofline and roofline.hbm.traffic), then the bench lines.
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$1; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
B=1024      # bench.py's default frames per step = frames per launch of the serial command
if [ "$2" != "lines" ]; then
SER="--steps 10 --warmup 2 --no-cpu-baseline --lanes 1 --region-timing --min-seconds 0 --no-also --no-parity"
# serial command (one stream, one launch per kernel over all B frames): kernel-trace stats and the two HBM counter passes
ORBX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats -- python $R/bench.py $SER > $D/stats.log 2>&1
ORBX_OVERLAP=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -o fetch -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --lanes 1 --region-timing --min-seconds 0 --no-also --no-parity > $D/fetch.log 2>&1
ORBX_OVERLAP=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D -o write -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --lanes 1 --region-timing --min-seconds 0 --no-also --no-parity > $D/write.log 2>&1
$R/tools/run_pmc.sh $1_pmc
python $R/tools/pmc_traffic.py $D/fetch_counter_collection.csv $D/write_counter_collection.csv $R/profiles/traffic.json $B $R/gpurun_out/$1_pmc 256 > $D/traffic.log 2>&1
cp $R/profiles/traffic.json $D/traffic.json
(cd $R && python tools/valu_mix.py > $D/valu_mix.log 2>&1; cp profiles/valu_mix.json $D/valu_mix.json)
python $R/tools/pmc_table.py $R/gpurun_out/$1_pmc/a_counter_collection.csv $R/gpurun_out/$1_pmc/b_counter_collection.csv > $D/pmc_sq_counters.txt 2>&1
python $R/tools/pmc_table.py $D/fetch_counter_collection.csv $D/write_counter_collection.csv > $D/pmc_fetch_write.txt 2>&1
fi
# the default command (4 lanes) under the kernel trace, then the bench lines proper
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats_overlap -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --min-seconds 0 --no-also --no-parity > $D/stats_overlap.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats_match -- python $R/bench.py --config match100k --steps 40 --warmup 5 --no-cpu-baseline --min-seconds 0 --no-parity > $D/stats_match.log 2>&1
cd $R
( time timeout 900 python bench.py --detail-file $D/bench.json > $D/bench.line.json 2> $D/bench.err ) 2> $D/bench_wall.txt          # the driver's command: headline + also{hd1080, match100k}, parity legs, CPU baselines
Q="--no-cpu-baseline --no-also --min-seconds 2"
timeout 300 python bench.py --lanes 1 $Q --detail-file $D/bench_one_lane.json > $D/bench_one_lane.line.json 2>/dev/null
timeout 300 python bench.py --region-timing $Q --detail-file $D/bench_region_timing.json > $D/bench_region_timing.line.json 2>/dev/null
timeout 300 python bench.py --config vga_extract $Q --detail-file $D/bench_extract_only.json > $D/bench_extract_only.line.json 2>/dev/null
timeout 300 python bench.py --family 0 $Q --detail-file $D/bench_noise.json > $D/bench_noise.line.json 2>/dev/null
ORBX_MATCH_MFMA=0 timeout 300 python bench.py --config match100k --no-cpu-baseline --min-seconds 2 --detail-file $D/bench_match100k_popcount.json > $D/bench_match100k_popcount.line.json 2>/dev/null
ORBX_MATCH_MFMA=8 timeout 300 python bench.py --config match100k --no-cpu-baseline --min-seconds 2 --detail-file $D/bench_match100k_int8.json > $D/bench_match100k_int8.line.json 2>/dev/null
timeout 600 python bench.py --gpus 2 --backend gloo --share-device --no-cpu-baseline --batch 512 --min-seconds 2 --also-min-seconds 1 --detail-file $D/bench_two_ranks_one_gpu_gloo.json > $D/bench_two_ranks_one_gpu_gloo.line.json 2>/dev/null
timeout 200 python tools/corun_probe.py > $D/corun_probe.json 2>/dev/null
(timeout 100 python tools/bench_single_frame.py; timeout 100 python tools/bench_single_frame.py 1920 1080 2000; timeout 100 orb_slam_amd/cpp/bench_single_frame; timeout 100 orb_slam_amd/cpp/bench_single_frame 1920 1080 2000) > $D/single_frame.txt 2>/dev/null
timeout 100 tools/microbench/valu_rate2 > $D/valu_issue_rates2.txt 2>&1
timeout 100 tools/microbench/mfma_valu_mix > $D/mfma_valu_mix.txt 2>&1
timeout 100 tools/microbench/mfma_layout > $D/mfma_layout.txt 2>&1
timeout 200 python tools/bench_kf_search.py 2>/dev/null | tail -1 > $D/kf_search.json
timeout 300 python tools/bench_frontend.py --window 15 2>/dev/null | tail -1 > $D/frontend_w15.json
timeout 300 python tools/bench_frontend.py 2>/dev/null | tail -1 > $D/frontend_w100.json
timeout 300 python tools/bench_frontend.py --window 15 --family 5 2>/dev/null | tail -1 > $D/frontend_w15_warp.json      # the correlated stream: the window search finds its partners
timeout 300 python tools/bench_orbmatcher_dropin.py --out $D/orbmatcher_dropin.json > $D/orbmatcher_dropin.txt 2>/dev/null    # one-problem latency of the drop-in ORBmatcher beside the reference's ORBmatcher.cc on the host
ls $D | wc -l
python -c "import json; d=json.load(open('$D/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], d['cpu_baseline']['value'], d.get('cpu_baseline_allcores')); print({k: (v['value'], v['roofline']['frac']) for k, v in d['also'].items()})"
