#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/exp_ab.sh ab44 c5:1 tree:1 c5:4 tree:4 c5:1:hd1080 tree:1:hd1080 c5:1 tree:1
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
