#!/bin/bash
# usage (on the GPU box): tools/run_pmc.sh <outdir-name>   — two rocprofv3 --pmc passes of SQ counters over the serial bench command (256 frames per launch, extract + match)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o a -- python $R/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O -o b -- python $R/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 > $O/b.log 2>&1
