#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab37 c3:1 tree:1 c3:4 tree:4 c3:2 tree:2 c3:1:hd1080 tree:1:hd1080 c3:1 tree:1
