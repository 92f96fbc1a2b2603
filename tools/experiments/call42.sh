#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/exp_ab.sh ab42 c5:1 tree:1 rt5:1 c5:1:hd1080 tree:1:hd1080 rt5:1:hd1080 c5:1 tree:1 rt5:1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py tests/test_gpu_bench_shapes.py -x -q -m gpu 2>&1 | tail -2
