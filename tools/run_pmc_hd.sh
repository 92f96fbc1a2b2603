#!/bin/bash
# usage (on the GPU box): tools/run_pmc_hd.sh <outdir-name>   — the counter passes of collect_round_artifacts.sh / run_pmc.sh for BASELINE configs[2]
# (1080p, 2000 keypoints, 256 frames per launch): FETCH_SIZE, WRITE_SIZE, two SQ passes (32 frames per launch) -> profiles/traffic_hd1080.json
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
C="--config hd1080 --no-cpu-baseline --lanes 1 --region-timing --min-seconds 0 --no-parity"
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch -- python $R/bench.py $C --steps 4 --warmup 1 > $O/fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o write -- python $R/bench.py $C --steps 4 --warmup 1 > $O/write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o a -- python $R/bench.py $C --batch 32 --ring 64 --steps 3 --warmup 2 > $O/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O -o b -- python $R/bench.py $C --batch 32 --ring 64 --steps 3 --warmup 2 > $O/b.log 2>&1
python $R/tools/pmc_traffic.py $O/fetch_counter_collection.csv $O/write_counter_collection.csv $R/profiles/traffic_hd1080.json 256 $O 32 hd_1920x1080_nf2000 > $O/traffic.log 2>&1
cp $R/profiles/traffic_hd1080.json $O/traffic_hd1080.json
(cd $R && python tools/valu_mix.py > $O/valu_mix.log 2>&1; cp profiles/valu_mix.json $O/valu_mix.json)
cd $R; timeout 400 python bench.py --config hd1080 --cpu-allcores-seconds 0 --detail-file $O/bench_hd.json > $O/bench_hd.line.json 2> $O/bench_hd.err
python -c "import json; d=json.loads(open('$O/bench_hd.json').read().strip().splitlines()[-1]); print(d['value'], d['roofline'])"
