#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab46 c5:2 tree:2 c5:1 tree:1 c5:4 tree:4 2>&1 | sed 's/quota.*//'
