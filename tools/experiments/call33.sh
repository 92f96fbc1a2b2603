#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab33 x4off:1 tree:1 x4w3:1 x4w2:1 x4off:1:hd1080 tree:1:hd1080 x4w3:1:hd1080 x4off:4 tree:4 x4w3:4
