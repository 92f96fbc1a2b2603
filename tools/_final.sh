R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tools/collect_round_artifacts.sh r3g 2>&1 | tail -3 | cut -c1-400
tools/run_pmc_hd.sh r3ghd 2>&1 | tail -1 | cut -c1-300
cd $R; (time python -m pytest tests -m gpu -q) > gpurun_out/r3g/pytest_gpu.txt 2>&1; grep "passed\|failed" gpurun_out/r3g/pytest_gpu.txt
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --backend gloo --share-device --no-cpu-baseline --batch 512 --min-seconds 1 --also-min-seconds 0.5 2>/dev/null | grep "^{" > gpurun_out/r3g/bench_torchrun_two_ranks.json
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from orb_slam_amd import synth
synth.frames(640, 480, synth.BLOCKS, 0, 2048).tofile("/tmp/frames.raw")
PY
(orb_slam_amd/cpp/example_lanes 640 480 1024 2 4 /tmp/frames.raw "" 200; orb_slam_amd/cpp/example_lanes 640 480 1024 2 1 /tmp/frames.raw "" 200) > gpurun_out/r3g/cpp_example_lanes.txt 2>&1; grep "frames/s\|IDENT" gpurun_out/r3g/cpp_example_lanes.txt
timeout 900 python tools/fuzz_parity.py 5000 419 > gpurun_out/r3g/fuzz_parity_5000.json 2>/dev/null; cut -c1-260 gpurun_out/r3g/fuzz_parity_5000.json
