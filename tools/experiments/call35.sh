#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -3
tools/exp_ab.sh ab35 c2:1 tree:1 c2:1:hd1080 tree:1:hd1080 c2:4 tree:4 c2:1 tree:1
