#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
ORBX_DBG_GEOM=1 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-also --min-seconds 0 --batch 64 2>&1 | grep -m2 "orbx geometry"
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
tools/exp_ab.sh ab25 head:1 tree:1 head:4 tree:4 head:0 tree:0 head:1:hd1080 tree:1:hd1080
