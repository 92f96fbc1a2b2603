#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
tools/exp_ab.sh ab31 tree:1 fs128:1 fs64:1 fs128p2:1 tree:4 fs128:4 fs64:4 tree:1
