#!/bin/bash
# usage (on the GPU box): tools/run_pmc_match.sh <outdir-name>  — matrix-pipe counters of the 100k x 100k matcher (BASELINE configs[4]) and of the
# per-frame batch matcher: one rocprofv3 --pmc pass each (kernel trace only), then a per-kernel table with the derived MFMA utilisation
#   MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE per XCD x 4 SIMDs x 256 CUs)   (rocprofiler-sdk's definition; tools/pmc_mfma_table.py)
R=${GRAFT_REPO_ROOT:-/root/repo}; O=$R/gpurun_out/$1; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_I8 SQ_INSTS_VALU_MFMA_F6F4 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"      # (ORBX_MATCH_MFMA=8 in the environment: the int8 kernels)
timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o m100k -- python $R/bench.py --config match100k --steps 6 --warmup 2 --no-cpu-baseline --min-seconds 0 --no-parity > $O/m100k.log 2>&1
ORBX_OVERLAP=0 timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O -o batch -- python $R/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity > $O/batch.log 2>&1
python $R/tools/pmc_mfma_table.py $O > $O/pmc_mfma.txt
cat $O/pmc_mfma.txt
