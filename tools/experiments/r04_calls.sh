#!/bin/bash
# The GPU calls of round 4, as they were run — ONE indexed script instead of 51 files (ADVICE r04 / VERDICT r04 #9).
# usage: tools/experiments/r04_calls.sh <n>      (on the GPU box: gpurun -- tools/experiments/r04_calls.sh <n>)
# The library variants they compare were scratch builds (build_variants/ is not tracked); what each call measured is in README.md / NOTES.md 9.
R=${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}; cd "$R"
case "$1" in
1)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/c1/pytest_gpu.txt 2>&1; tail -5 gpurun_out/c1/pytest_gpu.txt
ORBX_LIB=$GRAFT_REPO_ROOT/build_variants/pf/liborbx.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or end_to_end" > gpurun_out/c1/pytest_pf.txt 2>&1; tail -2 gpurun_out/c1/pytest_pf.txt
tools/exp_ab.sh c1 base:1 tree:1 onepass:1 e_win:1 e_patch:1 e_both:1 pf:1 pf_win:1 base:4 tree:4 base:0 tree:0 base:3 tree:3 base:1:hd1080 tree:1:hd1080 tree:1 base:1
timeout 900 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; tail -3 gpurun_out/c1/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/c1/bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["parity_checked_frames"], d["config"]["parity_mismatches"], d["config"]["parity_note"][-40:])
for k,v in d["also"].items(): print(k, v["value"], v["ms_per_step"], v["config"].get("parity_checked_frames"), v["config"]["parity_mismatches"], v.get("stage_ms_per_step"))
print(d["cpu_baseline"]); print(d.get("cpu_baseline_reference_source"))
PY
;;
2)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c2
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blur_planes" > gpurun_out/c2/pytest_blur.txt 2>&1; tail -15 gpurun_out/c2/pytest_blur.txt
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/c2/pytest_gpu.txt 2>&1; tail -5 gpurun_out/c2/pytest_gpu.txt
tools/exp_ab.sh c2 base:1 tree:1 noblurm:1 mbw8:1 mbw2:1 tree:1:hd1080 noblurm:1:hd1080 mbw8:1:hd1080 tree:0 selc2:0 tree:1 noblurm:1
cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c2 -o noise -- python $GRAFT_REPO_ROOT/bench.py --family 0 --lanes 1 --steps 5 --warmup 2 --no-cpu-baseline --min-seconds 0 --no-also --no-parity > $GRAFT_REPO_ROOT/gpurun_out/c2/noise.log 2>&1
head -12 $GRAFT_REPO_ROOT/gpurun_out/c2/noise_kernel_stats.csv | cut -c1-150
cd $GRAFT_REPO_ROOT
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/c2/bench.json 2> gpurun_out/c2/bench.err; tail -3 gpurun_out/c2/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/c2/bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["parity_checked_frames"], d["config"]["parity_mismatches"], d["stage_ms_per_step"])
for k,v in d["also"].items(): print(k, v["value"], v["ms_per_step"], v["config"].get("parity_checked_frames"), v["config"]["parity_mismatches"], v.get("stage_ms_per_step"))
PY
;;
3)
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
;;
4)
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
;;
6)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blur_planes" > gpurun_out/c6/pytest_blur.txt 2>&1; tail -12 gpurun_out/c6/pytest_blur.txt
tools/exp_ab.sh c6 tree:1 noblurm:1 tree:1:hd1080 noblurm:1:hd1080 tree:1 noblurm:1
;;
7)
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
;;
8)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c8; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blur_planes" > $O/pytest_blur.txt 2>&1; tail -3 $O/pytest_blur.txt
for v in tree bs6 bs12 bs99 noblurm; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['blur'], d['config']['parity_mismatches'])"
  ORBX_LIB=$lib timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/benchhd_$v.json 2>$O/benchhd_$v.err
  python -c "
import json; d=json.load(open('$O/benchhd_$v.json')); print('$v hd', d['value'], d['ms_per_step'], d['stage_ms_per_step']['blur'], d['config']['parity_mismatches'])"
done
;;
9)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c9; mkdir -p $O
for l in 2 4 6 8; do
  timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample --lanes $l > $O/lanes_$l.json 2>$O/lanes_$l.err
  python -c "
import json; d=json.load(open('$O/lanes_$l.json')); print('lanes $l', d['value'], d['ms_per_step'], d['config']['lanes'], d['config']['host_submit_ms_per_step'], d['config']['lane_placement']['probe_ms_per_step'])"
done
for l in 4 8; do
  timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample --lanes $l > $O/hdlanes_$l.json 2>$O/hdlanes_$l.err
  python -c "
import json; d=json.load(open('$O/hdlanes_$l.json')); print('hd lanes $l', d['value'], d['ms_per_step'], d['config']['lanes'])"
done
;;
10)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c10; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_pipeline.py -x -q > $O/pytest.txt 2>&1; tail -6 $O/pytest.txt
for v in tree nopipe qpw2 qpw8 dw2 dw2q8; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['describe'], d['config']['parity_mismatches'])"
done
for v in tree nopipe; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/benchhd_$v.json 2>$O/benchhd_$v.err
  python -c "
import json; d=json.load(open('$O/benchhd_$v.json')); print('$v hd', d['value'], d['ms_per_step'], d['stage_ms_per_step']['describe'], d['config']['parity_mismatches'])"
done
;;
11)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c11; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -k upload > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/bench_pcie.py --ingest kernel > $O/pcie_kernel.json 2>$O/pcie_kernel.err; cat $O/pcie_kernel.json; tail -2 $O/pcie_kernel.err
timeout 300 python tools/bench_pcie.py --ingest copy > $O/pcie_copy.json 2>$O/pcie_copy.err; cat $O/pcie_copy.json
HSA_ENABLE_SDMA=0 timeout 300 python tools/bench_pcie.py --ingest copy > $O/pcie_copy_nosdma.json 2>$O/pcie_copy_nosdma.err; cat $O/pcie_copy_nosdma.json
timeout 300 python tools/bench_pcie.py --ingest kernel --outputs counts > $O/pcie_kernel_counts.json 2>/dev/null; cat $O/pcie_kernel_counts.json
;;
12)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c12; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_golden.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in tree nopatch16 tree nopatch16; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['describe'], d['config']['parity_mismatches'])"
done
for v in tree nopatch16; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/benchhd_$v.json 2>$O/benchhd_$v.err
  python -c "
import json; d=json.load(open('$O/benchhd_$v.json')); print('$v hd', d['value'], d['ms_per_step'], d['stage_ms_per_step']['describe'], d['config']['parity_mismatches'])"
done
;;
13)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c13; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_golden.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
for v in tree nohyb hyb35 hybt32 tree nohyb; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['pyramid'], d['config']['parity_mismatches'])"
done
for v in tree nohyb; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/benchhd_$v.json 2>$O/benchhd_$v.err
  python -c "
import json; d=json.load(open('$O/benchhd_$v.json')); print('$v hd', d['value'], d['ms_per_step'], d['stage_ms_per_step']['pyramid'], d['config']['parity_mismatches'])"
done
;;
14)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c14; mkdir -p $O
ORBX_LIB=$GRAFT_REPO_ROOT/build_variants/ilp2hd/liborbx.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blur_planes" > $O/pytest_blur.txt 2>&1; tail -3 $O/pytest_blur.txt
for v in tree ilp2 tree ilp2; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['blur'], d['config']['parity_mismatches'])"
done
for v in tree mfmahd ilp2hd; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --config hd1080 --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/benchhd_$v.json 2>$O/benchhd_$v.err
  python -c "
import json; d=json.load(open('$O/benchhd_$v.json')); print('$v hd', d['value'], d['ms_per_step'], d['stage_ms_per_step']['blur'], d['config']['parity_mismatches'])"
done
;;
15)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c15; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python tools/fuzz_parity.py 4000 4001 > $O/fuzz_parity_4000.json 2>$O/fuzz_parity.err; tail -c 300 $O/fuzz_parity_4000.json
timeout 600 python tools/fuzz_batch.py 400 4002 > $O/fuzz_batch_400.json 2>$O/fuzz_batch.err; tail -c 300 $O/fuzz_batch_400.json
;;
16)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c16; mkdir -p $O
for bb in "512 2048" "1024 2048" "2048 4096" "4096 8192"; do
  set -- $bb
  timeout 400 python bench.py --no-cpu-baseline --no-also --min-seconds 2 --parity sample --batch $1 --ring $2 > $O/batch_$1.json 2>$O/batch_$1.err
  python -c "
import json; d=json.load(open('$O/batch_$1.json')); print('batch $1', d['value'], d['ms_per_step'], d['config']['lanes'], d['config']['parity_mismatches'])"
done
;;
17)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c17; mkdir -p $O
for v in tree qt2 tree qt2; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['match'], d['config']['parity_mismatches'])"
done
timeout 600 python -m pytest tests/test_gpu_bench.py -x -q -k "default_line" 2>&1 | tail -3
;;
18)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c18; mkdir -p $O
for v in tree pf22a pf22b pf22c pf33 pf12; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); print('$v', d['value'], d['ms_per_step'], d['stage_ms_per_step']['pyramid'], d['config']['parity_mismatches'])" || tail -2 $O/bench_$v.err
done
;;
19)
cd $GRAFT_REPO_ROOT
tools/exp_ab.sh c19 tree:1 fs9216:1 fs12288:1 fs6144:1 fs192:1 fs3:1 fs1:1 tree:4 fs9216:4 fs6144:4 tree:1:hd1080 fl8192:1:hd1080 fl6144:1:hd1080 fl10240:1:hd1080 fl512:1:hd1080 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$10,$11}'
;;
20)
cd $GRAFT_REPO_ROOT
tools/exp_ab.sh c20 tree:1 fs1:1 fs1b:1 fs1c:1 fs1d:1 fs1e:1 fs1f:1 tree:1 fs1:1 fs1:4 tree:4 fs1:0 tree:0 tree:1:hd1080 fl1:1:hd1080 fl1b:1:hd1080 fl1c:1:hd1080 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$10,$11}'
;;
21)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_golden.py -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for v in tree dw2 dw8 rz32 rz64 mbw2 mbw8 tree; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); s=d['stage_ms_per_step']; print('$v', d['value'], d['ms_per_step'], s['pyramid'], s['fast_cells'], s['blur'], s['describe'], d['config']['parity_mismatches'])"
done
;;
22)
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c22; mkdir -p $O
run() { name=$1; shift; env "$@" timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$name.json 2>$O/bench_$name.err
  python -c "
import json; d=json.load(open('$O/bench_$name.json')); print('$name', d['value'], d['ms_per_step'], d['config']['parity_mismatches'])"; }
run base A=1
run nooverlap ORBX_OVERLAP=0
run noaffinity ORBX_XCD_AFFINITY=0
run base2 A=1
run hwq8 GPU_MAX_HW_QUEUES=8
run hwq2 GPU_MAX_HW_QUEUES=2
;;
23)
cd $GRAFT_REPO_ROOT
tools/exp_ab.sh c23 tree:1 bs5:1 bs6:1 bs12:1 tree:1 2>&1 | awk '{print $1,$2,$3,$4,$5,$6,$7,$18,$19}'
;;
24)
cd ${GRAFT_REPO_ROOT:-$R}
tools/exp_ab.sh ab24 head:1 tree:1 head:4 tree:4 head:1 tree:1 head:1:hd1080 tree:1:hd1080
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
;;
25)
cd ${GRAFT_REPO_ROOT:-$R}
ORBX_DBG_GEOM=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --min-seconds 0 --batch 64 2>&1 | grep -m2 "orbx geometry"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
tools/exp_ab.sh ab25 head:1 tree:1 head:4 tree:4 head:0 tree:0 head:1:hd1080 tree:1:hd1080
;;
26)
cd ${GRAFT_REPO_ROOT:-$R}
tools/exp_ab.sh ab26 head:1 tree:1 poolrot:1 head:4 tree:4 poolrot:4 head:1:hd1080 tree:1:hd1080 poolrot:1:hd1080 head:1 tree:1 poolrot:1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
;;
27)
cd ${GRAFT_REPO_ROOT:-$R}
export ORBX_LIB=$PWD/build_variants/prof/liborbx.so
mkdir -p gpurun_out/prof27
timeout 120 python tools/fast_prof.py 1 > gpurun_out/prof27/f1.txt 2>&1
timeout 120 python tools/fast_prof.py 4 > gpurun_out/prof27/f4.txt 2>&1
timeout 120 python tools/fast_prof.py 1 1920 1080 64 > gpurun_out/prof27/hd.txt 2>&1
;;
28)
cd ${GRAFT_REPO_ROOT:-$R}
tools/exp_ab.sh ab28 head:1 tree:1 head:4 tree:4 head:1:hd1080 tree:1:hd1080 head:1 tree:1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
timeout 100 python tools/corun_probe.py 2>&1 | tail -5
;;
29)
cd ${GRAFT_REPO_ROOT:-$R}
export ORBX_LIB=$PWD/build_variants/profd/liborbx.so PROF_KERNEL=describe
mkdir -p gpurun_out/prof29
timeout 120 python tools/fast_prof.py 1 > gpurun_out/prof29/f1.txt 2>&1
timeout 120 python tools/fast_prof.py 1 1920 1080 64 > gpurun_out/prof29/hd.txt 2>&1
unset ORBX_LIB
timeout 100 python tools/corun_probe.py 2>&1 | tail -3
;;
30)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab30 head:1 tree:1 qpw1:1 qpw2:1 qpw8:1 head:1:hd1080 tree:1:hd1080 qpw8:1:hd1080 head:4 tree:4
;;
31)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab31 tree:1 fs128:1 fs64:1 fs128p2:1 tree:4 fs128:4 fs64:4 tree:1
;;
32)
R=${GRAFT_REPO_ROOT:-$R}; O=$R/gpurun_out/pmc32; mkdir -p $O; cd /tmp; export TMPDIR=/tmp; export ORBX_OVERLAP=0
timeout 60 rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TA_[A-Za-z0-9_]*\|TCP_[A-Za-z0-9_]*\|TD_[A-Za-z0-9_]*" $O/avail.txt | sort -u | tr '\n' ' ' | head -c 6000
echo
CMD="python $R/bench.py --batch 256 --ring 512 --steps 3 --warmup 2 --lanes 1 --region-timing --no-cpu-baseline --min-seconds 0 --no-also --no-parity"
timeout 200 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TD_TD_BUSY_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O -o c -- $CMD > $O/c.log 2>&1
tail -3 $O/c.log
timeout 200 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum --kernel-trace --output-format csv -d $O -o d -- $CMD > $O/d.log 2>&1
tail -3 $O/d.log
ls $O
;;
33)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab33 x4off:1 tree:1 x4w3:1 x4w2:1 x4off:1:hd1080 tree:1:hd1080 x4w3:1:hd1080 x4off:4 tree:4 x4w3:4
;;
34)
cd ${GRAFT_REPO_ROOT:-$R}
tools/exp_ab.sh ab34 tree:1 abl1:1 abl2:1 abl4:1 abl8:1 abl16:1 tree:1
;;
35)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -3
tools/exp_ab.sh ab35 c2:1 tree:1 c2:1:hd1080 tree:1:hd1080 c2:4 tree:4 c2:1 tree:1
;;
36)
cd ${GRAFT_REPO_ROOT:-$R}
Q="--no-cpu-baseline --no-also --min-seconds 2 --parity none"
for f in 0 20000 23000 27000 32000; do
  echo "floor $f: $(ORBX_FAST_LDS_FLOOR=$f timeout 200 python bench.py $Q 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
echo "floor 0 again: $(timeout 200 python bench.py $Q 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
;;
37)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab37 c3:1 tree:1 c3:4 tree:4 c3:2 tree:2 c3:1:hd1080 tree:1:hd1080 c3:1 tree:1
;;
38)
cd ${GRAFT_REPO_ROOT:-$R}
Q="--no-cpu-baseline --no-also --min-seconds 2 --parity none"
for v in tree dw2 dw3 mw2 tree; do
  lib=$PWD/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$PWD/orb_slam_amd/liborbx.so
  echo "$v: $(ORBX_LIB=$lib timeout 200 python bench.py $Q 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
;;
39)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab39 c4:0 tree:0 c4:1 tree:1 c4:1:hd1080 tree:1:hd1080 c4:4 tree:4
Q="--no-cpu-baseline --no-also --min-seconds 2"
for v in c4 tree; do
  lib=$PWD/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$PWD/orb_slam_amd/liborbx.so
  echo "$v noise 4 lanes: $(ORBX_LIB=$lib timeout 200 python bench.py $Q --family 0 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"], d["config"]["parity_mismatches"])')"
done
;;
40)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -3
tools/exp_ab.sh ab40 c5:1 tree:1 c5:1:hd1080 tree:1:hd1080 c5:1 tree:1
timeout 300 python tools/fuzz_parity.py 400 977 2>/dev/null | tail -c 300
;;
41)
cd ${GRAFT_REPO_ROOT:-$R}
tools/exp_ab.sh ab41 c5:1 rz6:1 rz5:1 tree:1 c5:1:hd1080 rz6:1:hd1080 rz5:1:hd1080
;;
42)
cd ${GRAFT_REPO_ROOT:-$R}
tools/exp_ab.sh ab42 c5:1 tree:1 rt5:1 c5:1:hd1080 tree:1:hd1080 rt5:1:hd1080 c5:1 tree:1 rt5:1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -2
;;
43)
cd ${GRAFT_REPO_ROOT:-$R}; mkdir -p gpurun_out/fz43
timeout 900 python tools/fuzz_parity.py 8000 5201 > gpurun_out/fz43/fuzz_parity_8000.json 2>/dev/null; tail -c 250 gpurun_out/fz43/fuzz_parity_8000.json; echo
timeout 900 python tools/fuzz_batch.py 700 5202 > gpurun_out/fz43/fuzz_batch_700.json 2>/dev/null; tail -c 250 gpurun_out/fz43/fuzz_batch_700.json; echo
timeout 600 python tools/fuzz_frontend.py 8000 5203 > gpurun_out/fz43/fuzz_frontend_8000.json 2>/dev/null; tail -c 250 gpurun_out/fz43/fuzz_frontend_8000.json; echo
;;
44)
cd ${GRAFT_REPO_ROOT:-$R}
tools/exp_ab.sh ab44 c5:1 tree:1 c5:4 tree:4 c5:1:hd1080 tree:1:hd1080 c5:1 tree:1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
;;
45)
cd ${GRAFT_REPO_ROOT:-$R}; mkdir -p gpurun_out/c45
timeout 120 tools/microbench/valu_exec_mask > gpurun_out/c45/valu_exec_mask.txt; grep "min3\|cndmask" gpurun_out/c45/valu_exec_mask.txt | head -30
( time timeout 900 python bench.py > gpurun_out/c45/bench.json 2> gpurun_out/c45/bench.err ) 2> gpurun_out/c45/bench_wall.txt
python -c "
import json; d=json.load(open('gpurun_out/c45/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['config']['parity_mismatches'], {k:(v['value'], v['config']['parity_mismatches']) for k,v in d['also'].items()})"
;;
46)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab46 c5:2 tree:2 c5:1 tree:1 c5:4 tree:4 2>&1 | sed 's/quota.*//'
;;
47)
cd ${GRAFT_REPO_ROOT:-$R}
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
tools/exp_ab.sh ab47 c5:2 tree:2 c5:1 tree:1 c5:4 tree:4 c5:3 tree:3 2>&1 | sed 's/quota.*//'
;;
48)
# the fallback hint with a run length of six needs more than exp_ab's 2 warm-up steps to settle: 8 warm-up + 30 timed steps per run
R=${GRAFT_REPO_ROOT:-$R}; cd $R; D=$R/gpurun_out/ab48; mkdir -p $D
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
S="--lanes 1 --steps 30 --warmup 8 --no-cpu-baseline --region-timing --min-seconds 0 --no-also --no-parity"
for item in c5:1 tree:1 c5:3 tree:3 c5:4 tree:4 c5:1 tree:1; do
  v=${item%%:*}; f=${item#*:}
  lib=$R/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$R/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py $S --family $f > $D/${v}_f$f.json 2>/dev/null
  python -c "
import json; d=json.load(open('$D/${v}_f$f.json')); print('%-5s fam $f step %.4f fast %.4f' % ('$v', d['ms_per_step'], d['stage_ms_per_step']['fast_cells']))"
done
;;
49)
# lean re-collection for the last kernel change of the round (the hash-gated counter files and the bench lines; fuzz and the full GPU suite ran on the build before)
R=${GRAFT_REPO_ROOT:-$R}; cd $R
tools/collect_round_artifacts.sh r4d 2>&1 | tail -3 | cut -c1-400
tools/run_pmc_hd.sh r4dhd 2>&1 | tail -1 | cut -c1-200
cd $R; timeout 120 python -m pytest tests/test_golden.py tests/test_gpu_bench_shapes.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
;;
50)
cd ${GRAFT_REPO_ROOT:-$R}
Q="--no-cpu-baseline --no-also --min-seconds 2.5 --parity none"
for v in c5 tree c5 tree; do
  lib=$PWD/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$PWD/orb_slam_amd/liborbx.so
  echo "$v: $(ORBX_LIB=$lib timeout 100 python bench.py $Q 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
;;
51)
# the hint written by the first frame of a launch group only (scratch build build_variants/hintf0): four-lane step against the shipped build
cd ${GRAFT_REPO_ROOT:-$R}
Q="--no-cpu-baseline --no-also --min-seconds 2 --parity none"
for item in c5:1 hintf0:1 c5:1 hintf0:1 c5:3 hintf0:3; do
  v=${item%%:*}; f=${item#*:}
  echo "$v fam $f: $(ORBX_LIB=$PWD/build_variants/$v/liborbx.so timeout 60 python bench.py $Q --family $f 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
ORBX_LIB=$PWD/build_variants/hintf0/liborbx.so timeout 60 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "family or stage or batch" 2>&1 | tail -1
;;
*) echo "usage: $0 <call number>  (1..51, see README.md)"; exit 2 ;;
esac
