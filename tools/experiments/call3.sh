cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c3; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
B="python $GRAFT_REPO_ROOT/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity"
for v in tree noblurm; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  export ORBX_LIB=$lib
  timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o ${v}_fetch -- $B > $O/${v}_fetch.log 2>&1
  timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o ${v}_write -- $B > $O/${v}_write.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O -o ${v}_a -- $B > $O/${v}_a.log 2>&1
  timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM --kernel-trace --output-format csv -d $O -o ${v}_b -- $B > $O/${v}_b.log 2>&1
  timeout 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d $O -o ${v}_c -- $B > $O/${v}_c.log 2>&1
  timeout 200 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum --kernel-trace --output-format csv -d $O -o ${v}_d -- $B > $O/${v}_d.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_table.py $O/${v}_fetch_counter_collection.csv $O/${v}_write_counter_collection.csv $O/${v}_a_counter_collection.csv $O/${v}_b_counter_collection.csv $O/${v}_c_counter_collection.csv $O/${v}_d_counter_collection.csv 2>&1 | grep -i "kernel\|blur" | cut -c1-400
done
grep -i "blur" $O/tree_a_kernel_trace.csv | head -2 | cut -c1-400; head -1 $O/tree_a_kernel_trace.csv
tail -3 $O/tree_c.log; tail -3 $O/tree_d.log
