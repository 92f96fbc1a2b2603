#!/bin/bash
# lean re-collection for the last kernel change of the round (the hash-gated counter files and the bench lines; fuzz and the full GPU suite ran on the build before)
R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R
tools/collect_round_artifacts.sh r4d 2>&1 | tail -3 | cut -c1-400
tools/run_pmc_hd.sh r4dhd 2>&1 | tail -1 | cut -c1-200
cd $R; timeout 120 python -m pytest tests/test_golden.py tests/test_gpu_bench_shapes.py tests/test_gpu_pipeline.py -x -q -m gpu 2>&1 | tail -1
