#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/pmc32; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
timeout 60 rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TA_[A-Za-z0-9_]*\|TCP_[A-Za-z0-9_]*\|TD_[A-Za-z0-9_]*" $O/avail.txt | sort -u | tr '\n' ' ' | head -c 6000
echo
CMD="python $R/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity"
timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o c -- $CMD > $O/c.log 2>&1
tail -3 $O/c.log
timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $O -o d -- $CMD > $O/d.log 2>&1
tail -3 $O/d.log
ls $O
