#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ORBX_LIB=$PWD/build_variants/prof/liborbx.so
mkdir -p gpurun_out/prof27
timeout 120 python tools/fast_prof.py 1 > gpurun_out/prof27/f1.txt 2>&1
timeout 120 python tools/fast_prof.py 4 > gpurun_out/prof27/f4.txt 2>&1
timeout 120 python tools/fast_prof.py 1 1920 1080 64 > gpurun_out/prof27/hd.txt 2>&1
