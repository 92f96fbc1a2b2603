#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -1
tools/exp_ab.sh ab47 c5:2 tree:2 c5:1 tree:1 c5:4 tree:4 c5:3 tree:3 2>&1 | sed 's/quota.*//'
