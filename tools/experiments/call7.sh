cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c7; mkdir -p $O
for v in tree noblurm; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
  ORBX_LIB=$lib timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/benchhd_$v.json 2>$O/benchhd_$v.err
  python -c "
import json; d=json.load(open('$O/benchhd_$v.json')); print('$v hd', d['value'], d['ms_per_step'], d['stage_ms_per_step'])"
done
cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
B="python $GRAFT_REPO_ROOT/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity"
v=tree
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o ${v}_fetch -- $B > $O/${v}_fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o ${v}_write -- $B > $O/${v}_write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O -o ${v}_a -- $B > $O/${v}_a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o ${v}_b -- $B > $O/${v}_b.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_table.py $O/${v}_fetch_counter_collection.csv $O/${v}_write_counter_collection.csv $O/${v}_a_counter_collection.csv $O/${v}_b_counter_collection.csv 2>&1 | grep -i "kernel\|blur" | cut -c1-400
grep "k_blur_mfma" $O/tree_a_kernel_trace.csv | head -3 | awk -F, '{print $10, $11, $11-$10}'
