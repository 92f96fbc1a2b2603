#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -3
tools/exp_ab.sh ab40 c5:1 tree:1 c5:1:hd1080 tree:1:hd1080 c5:1 tree:1
timeout 300 python tools/fuzz_parity.py 400 977 2>/dev/null | tail -c 300
