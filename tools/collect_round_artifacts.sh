#!/bin/bash
# Everything profiles/ is built from, in one go on the GPU box.  usage: tools/collect_round_artifacts.sh <name>   (-> gpurun_out/<name>, gpurun_out/<name>_pmc)
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$1; mkdir -p $D; cd /tmp; export TMPDIR=/tmp
timeout 400 python $R/bench.py > $D/bench.json 2> $D/bench.err
ORBX_OVERLAP=0 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline --lanes 1 > $D/stats.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats_overlap -- python $R/bench.py --steps 10 --warmup 2 --no-cpu-baseline > $D/stats_overlap.log 2>&1
ORBX_OVERLAP=0 timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -o fetch -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --lanes 1 > $D/fetch.log 2>&1
ORBX_OVERLAP=0 timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D -o write -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --lanes 1 > $D/write.log 2>&1
$R/tools/run_pmc.sh $1_pmc
cd $R
timeout 300 python bench.py --no-match --no-cpu-baseline > $D/bench_extract_only.json 2>/dev/null
timeout 400 python bench.py --width 1920 --height 1080 --nfeatures 2000 --batch 128 --ring 256 > $D/bench_hd.json 2>/dev/null
timeout 300 python bench.py --family 0 --no-cpu-baseline > $D/bench_noise.json 2>/dev/null
timeout 200 python tools/bench_match.py > $D/match100k.txt 2>/dev/null
(timeout 100 python tools/bench_single_frame.py; timeout 100 python tools/bench_single_frame.py 1920 1080 2000) > $D/single_frame.txt 2>/dev/null
timeout 200 python tools/bench_kf_search.py 2>/dev/null | tail -1 > $D/kf_search.json
timeout 300 python tools/bench_frontend.py --window 15 2>/dev/null | tail -1 > $D/frontend_w15.json
timeout 300 python tools/bench_frontend.py 2>/dev/null | tail -1 > $D/frontend_w100.json
ls $D | wc -l
python -c "import json; d=json.load(open('$D/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['cpu_baseline']['value'])"
