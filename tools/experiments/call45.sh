#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/c45
timeout 120 tools/microbench/valu_exec_mask > gpurun_out/c45/valu_exec_mask.txt; grep "min3\|cndmask" gpurun_out/c45/valu_exec_mask.txt | head -30
( time timeout 900 python bench.py > gpurun_out/c45/bench.json 2> gpurun_out/c45/bench.err ) 2> gpurun_out/c45/bench_wall.txt
python -c "
import json; d=json.load(open('gpurun_out/c45/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['parity_mismatches'], {k:(v['value'], v['config']['parity_mismatches']) for k,v in d['also'].items()})"
