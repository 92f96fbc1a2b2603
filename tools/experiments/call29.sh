#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
export ORBX_LIB=$PWD/build_variants/profd/liborbx.so PROF_KERNEL=describe
mkdir -p gpurun_out/prof29
timeout 120 python tools/fast_prof.py 1 > gpurun_out/prof29/f1.txt 2>&1
timeout 120 python tools/fast_prof.py 1 1920 1080 64 > gpurun_out/prof29/hd.txt 2>&1
unset ORBX_LIB
timeout 100 python tools/corun_probe.py 2>&1 | tail -3
