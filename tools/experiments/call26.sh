#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
tools/exp_ab.sh ab26 head:1 tree:1 poolrot:1 head:4 tree:4 poolrot:4 head:1:hd1080 tree:1:hd1080 poolrot:1:hd1080 head:1 tree:1 poolrot:1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
