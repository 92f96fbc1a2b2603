#!/bin/bash
# One GPU call for a kernel change when GPU-minutes are short: (1) the extractor parity tests on the new library, (2) the serial per-kernel
# timings of the VGA and 1080p streams for build_variants/base/liborbx.so (the library before the change) and the new one ON THE SAME BOX,
# (3) only if the tests pass and k_fast_cells got faster by more than box noise: the minimum that the hash-gated files of profiles/ need
# (kernel stats, FETCH / WRITE / two SQ passes for both streams -> traffic.json, traffic_hd1080.json), the default bench line, the GPU suite.
# usage: tools/ab_then_collect.sh <name> [min gain, default 0.02]   (-> gpurun_out/<name>/)
N=${1:?name}; GAIN=${2:-0.02}
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$N; mkdir -p $D; cd $R
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_golden.py tests/test_gpu_select.py tests/test_gpu_pipeline.py -x -q > $D/pytest_quick.txt 2>&1
tail -2 $D/pytest_quick.txt
grep -q " passed" $D/pytest_quick.txt && ! grep -q "failed\|error" $D/pytest_quick.txt || { echo "GATE: parity tests failed"; exit 1; }
S="--lanes 1 --steps 20 --warmup 3 --no-cpu-baseline --region-timing --min-seconds 0 --no-also --no-parity"
for rep in 1 2; do
  ORBX_LIB=$R/build_variants/base/liborbx.so timeout 200 python bench.py $S --detail-file $D/base_vga_$rep.json > $D/base_vga_$rep.line.json 2>/dev/null
  timeout 200 python bench.py $S --detail-file $D/new_vga_$rep.json > $D/new_vga_$rep.line.json 2>/dev/null
done
ORBX_LIB=$R/build_variants/base/liborbx.so timeout 200 python bench.py $S --config hd1080 --detail-file $D/base_hd.json > $D/base_hd.line.json 2>/dev/null
timeout 200 python bench.py $S --config hd1080 --detail-file $D/new_hd.json > $D/new_hd.line.json 2>/dev/null
python - <<PY
import json, sys
def fast(n):
    d = json.load(open("$D/%s.json" % n)); return d["stage_ms_per_step"]["fast_cells"], d["ms_per_step"]
b = min(fast("base_vga_1")[0], fast("base_vga_2")[0]); n = min(fast("new_vga_1")[0], fast("new_vga_2")[0])
print("VGA  k_fast_cells base %.4f new %.4f ms (%.1f %%); step base %.4f new %.4f" % (b, n, 100 * (n / b - 1), min(fast("base_vga_1")[1], fast("base_vga_2")[1]), min(fast("new_vga_1")[1], fast("new_vga_2")[1])))
bh, nh = fast("base_hd"), fast("new_hd")
print("1080 k_fast_cells base %.4f new %.4f ms (%.1f %%); step base %.4f new %.4f" % (bh[0], nh[0], 100 * (nh[0] / bh[0] - 1), bh[1], nh[1]))
open("$D/gate.txt", "w").write("go" if n < b * (1 - $GAIN) else "stop")
PY
[ "$(cat $D/gate.txt)" = "go" ] || { echo "GATE: gain below $GAIN, nothing collected"; exit 0; }
cd /tmp; export TMPDIR=/tmp
Q="--no-cpu-baseline --lanes 1 --region-timing --min-seconds 0 --no-also --no-parity"
ORBX_OVERLAP=0 timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $D -o stats -- python $R/bench.py --steps 10 --warmup 2 $Q > $D/stats.log 2>&1
ORBX_OVERLAP=0 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D -o fetch -- python $R/bench.py --steps 4 --warmup 1 $Q > $D/fetch.log 2>&1
ORBX_OVERLAP=0 timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D -o write -- python $R/bench.py --steps 4 --warmup 1 $Q > $D/write.log 2>&1
$R/tools/run_pmc.sh ${N}_pmc
python $R/tools/pmc_traffic.py $D/fetch_counter_collection.csv $D/write_counter_collection.csv $R/profiles/traffic.json 1024 $R/gpurun_out/${N}_pmc 256 > $D/traffic.log 2>&1
cp $R/profiles/traffic.json $D/traffic.json
python $R/tools/pmc_table.py $R/gpurun_out/${N}_pmc/a_counter_collection.csv $R/gpurun_out/${N}_pmc/b_counter_collection.csv > $D/pmc_sq_counters.txt 2>&1
python $R/tools/pmc_table.py $D/fetch_counter_collection.csv $D/write_counter_collection.csv > $D/pmc_fetch_write.txt 2>&1
O=$R/gpurun_out/${N}hd; mkdir -p $O; export ORBX_OVERLAP=0
C="--config hd1080 --no-cpu-baseline --lanes 1 --region-timing --min-seconds 0 --no-parity"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O -o fetch -- python $R/bench.py $C --steps 4 --warmup 1 > $O/fetch.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O -o write -- python $R/bench.py $C --steps 4 --warmup 1 > $O/write.log 2>&1
timeout 200 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o a -- python $R/bench.py $C --batch 32 --ring 64 --steps 3 --warmup 2 > $O/a.log 2>&1
timeout 200 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace --output-format csv -d $O -o b -- python $R/bench.py $C --batch 32 --ring 64 --steps 3 --warmup 2 > $O/b.log 2>&1
python $R/tools/pmc_traffic.py $O/fetch_counter_collection.csv $O/write_counter_collection.csv $R/profiles/traffic_hd1080.json 256 $O 32 hd_1920x1080_nf2000 > $O/traffic.log 2>&1
cp $R/profiles/traffic_hd1080.json $O/traffic_hd1080.json
unset ORBX_OVERLAP
(cd $R && python tools/valu_mix.py > $D/valu_mix.log 2>&1; cp profiles/valu_mix.json $D/valu_mix.json)
cd $R
timeout 400 python bench.py --detail-file $D/bench.json > $D/bench.line.json 2> $D/bench.err
python -c "import json; d=json.load(open('$D/bench.json')); print(d['value'], d['ms_per_step'], d['roofline'], d.get('roofline_valu')); print({k: (v['value'], v['roofline'].get('frac'), v['roofline'].get('traffic')) for k, v in d['also'].items()})"
(time timeout 600 python -m pytest tests -m gpu -q) > $D/pytest_gpu.txt 2>&1; tail -4 $D/pytest_gpu.txt | head -2
