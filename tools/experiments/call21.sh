cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c21; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_golden.py -x -q > $O/pytest.txt 2>&1; tail -2 $O/pytest.txt
for v in tree dw2 dw8 rz32 rz64 mbw2 mbw8 tree; do
  lib=$GRAFT_REPO_ROOT/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$GRAFT_REPO_ROOT/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --parity sample > $O/bench_$v.json 2>$O/bench_$v.err
  python -c "
import json; d=json.load(open('$O/bench_$v.json')); s=d['stage_ms_per_step']; print('$v', d['value'], d['ms_per_step'], s['pyramid'], s['fast_cells'], s['blur'], s['describe'], d['config']['parity_mismatches'])"
done
