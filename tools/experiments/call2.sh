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
