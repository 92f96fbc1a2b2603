#!/bin/bash
# a second long fuzz slice on the round's final build (seeds 7101 / 7102 / 7105); -> gpurun_out/r05bfz2
R=${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}; cd $R; D=gpurun_out/r05bfz2; mkdir -p $D
timeout 420 python tools/fuzz_parity.py 18000 7101 > $D/fuzz_parity_18000.json 2>/dev/null; tail -c 220 $D/fuzz_parity_18000.json; echo
timeout 200 python tools/fuzz_batch.py 1100 7102 > $D/fuzz_batch_1100.json 2>/dev/null; tail -c 260 $D/fuzz_batch_1100.json; echo
timeout 150 python tools/fuzz_match.py 30000 7105 > $D/fuzz_match_30000.json 2>/dev/null; tail -c 330 $D/fuzz_match_30000.json; echo
