#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab30 head:1 tree:1 qpw1:1 qpw2:1 qpw8:1 head:1:hd1080 tree:1:hd1080 qpw8:1:hd1080 head:4 tree:4
