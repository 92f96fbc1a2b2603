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
