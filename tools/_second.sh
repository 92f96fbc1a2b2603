R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/r3b; mkdir -p $D; cd $R
(time python -m pytest tests -m gpu -q) > $D/pytest_gpu.txt 2>&1; tail -1 $D/pytest_gpu.txt | head -1; grep "passed\|failed" $D/pytest_gpu.txt
tools/run_pmc_hd.sh r3hd 2>&1 | tail -2
timeout 900 python tools/fuzz_parity.py 6000 311 > $D/fuzz_parity_6000.json 2>$D/fuzz_parity.err; cat $D/fuzz_parity_6000.json | cut -c1-300
timeout 900 python tools/fuzz_frontend.py 3000 77 > $D/fuzz_frontend_3000.json 2>$D/fuzz_frontend.err; tail -c 400 $D/fuzz_frontend_3000.json
python - <<PY
import numpy as np, sys
sys.path.insert(0, "$R")
from orb_slam_amd import synth
synth.frames(640, 480, synth.BLOCKS, 0, 2048).tofile("/tmp/frames.raw")
PY
(orb_slam_amd/cpp/example_lanes 640 480 1024 2 4 /tmp/frames.raw "" 200; orb_slam_amd/cpp/example_lanes 640 480 1024 2 1 /tmp/frames.raw "" 200) > $D/cpp_example_lanes.txt 2>&1; cat $D/cpp_example_lanes.txt | tail -12
for e in ORBX_OVERLAP=0 ORBX_XCD_AFFINITY=0 ORBX_PYR_PER_LEVEL=1 ORBX_ZERO_COPY=0; do env $e timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -x -q -k "not randomised" 2>&1 | tail -1 | sed "s/^/$e: /"; done
