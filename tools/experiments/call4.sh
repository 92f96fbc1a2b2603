cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c4
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blur_planes" > gpurun_out/c4/pytest_blur.txt 2>&1; tail -3 gpurun_out/c4/pytest_blur.txt
tools/exp_ab.sh c4 tree:1 noblurm:1 mbw8:1 mbw2:1 tree:1:hd1080 noblurm:1:hd1080 tree:1 noblurm:1
O=$GRAFT_REPO_ROOT/gpurun_out/c4; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
B="python $GRAFT_REPO_ROOT/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity"
for v in tree; do
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o ${v}_fetch -- $B > $O/${v}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o ${v}_write -- $B > $O/${v}_write.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O -o ${v}_a -- $B > $O/${v}_a.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --kernel-trace --output-format csv -d $O -o ${v}_b -- $B > $O/${v}_b.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_table.py $O/${v}_fetch_counter_collection.csv $O/${v}_write_counter_collection.csv $O/${v}_a_counter_collection.csv $O/${v}_b_counter_collection.csv 2>&1 | grep -i "kernel\|blur" | cut -c1-300
done
