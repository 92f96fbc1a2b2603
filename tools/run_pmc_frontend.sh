#!/bin/bash
# PMC passes over the whole front-end (tools/bench_frontend.py), one counter group per run (kernel-trace only, as gpurun requires).
# usage: tools/run_pmc_frontend.sh <outdir under gpurun_out>
set -u
OUT=/root/repo/gpurun_out/$1
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVES GRBM_GUI_ACTIVE"; do
  tag=$(echo $grp | cut -d' ' -f1)
  FRONTEND_CPU=0 timeout 300 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT -o pmc_$tag -- python /root/repo/tools/bench_frontend.py --steps 2 --window 15 > $OUT/log_$tag.txt 2>&1
done
ls $OUT
