#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/fz43
timeout 900 python tools/fuzz_parity.py 8000 5201 > gpurun_out/fz43/fuzz_parity_8000.json 2>/dev/null; tail -c 250 gpurun_out/fz43/fuzz_parity_8000.json; echo
timeout 900 python tools/fuzz_batch.py 700 5202 > gpurun_out/fz43/fuzz_batch_700.json 2>/dev/null; tail -c 250 gpurun_out/fz43/fuzz_batch_700.json; echo
timeout 600 python tools/fuzz_frontend.py 8000 5203 > gpurun_out/fz43/fuzz_frontend_8000.json 2>/dev/null; tail -c 250 gpurun_out/fz43/fuzz_frontend_8000.json; echo
