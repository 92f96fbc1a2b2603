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
