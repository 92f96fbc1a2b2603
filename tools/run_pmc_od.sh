#!/bin/bash
# usage (on the GPU box): tools/run_pmc_od.sh <outdir-name>  — counter passes over the serial command for k_describe_od (round 6): issue / wait / LDS / matrix pipe / texture addresser
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
B="python $R/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity --live-traffic off"
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o a -- $B > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O -o b -- $B > $O/b.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O -o c -- $B > $O/c.log 2>&1
timeout 200 rocprofv3 --pmc TA_TA_BUSY_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum TD_TD_BUSY_sum --kernel-trace --output-format csv -d $O -o d -- $B > $O/d.log 2>&1
python $R/tools/pmc_table.py $(find $O -name "*counter_collection.csv") > $O/pmc_table.txt 2>&1
grep -E "kernel|describe|fast_cells|blur" $O/pmc_table.txt
find $O -name "*kernel_trace.csv" | head -1 | xargs -I{} python $R/tools/rocprof_summary.py {} 2>/dev/null | head -20
