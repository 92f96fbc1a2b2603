#!/bin/bash
# usage (on the GPU box): tools/run_pmc_clock.sh <outdir-name>  — the shader clock every hot kernel actually runs at: one rocprofv3 --pmc pass
# (GRBM_GUI_ACTIVE, SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, SQ_WAVE_CYCLES; kernel trace only) over the serial command with 1024 frames per launch
# (long dispatches: the counter start / stop around a dispatch is then a few percent of GRBM_GUI_ACTIVE), VGA and 1080p.
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
C="GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES"
S="--steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity"
timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o vga -- python $R/bench.py $S > $O/vga.log 2>&1
timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o hd -- python $R/bench.py $S --config hd1080 > $O/hd.log 2>&1
python $R/tools/pmc_clock_table.py $O > $O/pmc_clock.txt
python $R/tools/pmc_clock_json.py $O/pmc_clock.txt $R/profiles/clock.json && cp $R/profiles/clock.json $O/clock.json
cat $O/pmc_clock.txt
