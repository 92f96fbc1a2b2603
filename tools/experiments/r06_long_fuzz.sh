#!/bin/bash
# Round 6, final build: longer fuzz slices on other seeds (one GPU call; ~12 GPU-minutes).  Results -> gpurun_out/r06_fuzz/
cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/r06_fuzz; mkdir -p $O
timeout 900 python tools/fuzz_orbmatcher.py 60000 12106 > $O/fuzz_orbmatcher_60000.json 2> $O/orbmatcher.err; echo "orbmatcher rc=$?"; tail -c 400 $O/fuzz_orbmatcher_60000.json
timeout 900 python tools/fuzz_orbmatcher.py 40000 12107 real > $O/fuzz_orbmatcher_real_access_40000.json 2>> $O/orbmatcher.err; echo "orbmatcher (ORBmatcherAccess.h) rc=$?"; tail -c 300 $O/fuzz_orbmatcher_real_access_40000.json
timeout 900 python tools/fuzz_batch.py 1000 12102 > $O/fuzz_batch_1000.json 2>/dev/null; echo "batch rc=$?"; tail -c 300 $O/fuzz_batch_1000.json
timeout 900 python tools/fuzz_parity.py 12000 12101 > $O/fuzz_parity_12000.json 2>/dev/null; echo "parity rc=$?"; tail -c 300 $O/fuzz_parity_12000.json
timeout 600 python tools/fuzz_frontend.py 8000 12104 > $O/fuzz_frontend_8000.json 2>/dev/null; echo "frontend rc=$?"; tail -c 300 $O/fuzz_frontend_8000.json
timeout 600 python tools/fuzz_match.py 40000 12105 > $O/fuzz_match_40000.json 2>/dev/null; echo "match rc=$?"; tail -c 300 $O/fuzz_match_40000.json
