#!/bin/bash
# the long fuzz slice on the round's final build (other seeds than tools/round_end.sh) + the default command once more; -> gpurun_out/r05bfz
R=${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}; cd $R; D=gpurun_out/r05bfz; mkdir -p $D
timeout 900 python tools/fuzz_match.py 80000 6105 > $D/fuzz_match_80000.json 2>/dev/null; tail -c 330 $D/fuzz_match_80000.json; echo
timeout 600 python tools/fuzz_parity.py 10000 6101 > $D/fuzz_parity_10000.json 2>/dev/null; tail -c 220 $D/fuzz_parity_10000.json; echo
timeout 600 python tools/fuzz_batch.py 900 6102 > $D/fuzz_batch_900.json 2>/dev/null; tail -c 260 $D/fuzz_batch_900.json; echo
timeout 300 python tools/fuzz_frontend.py 8000 6104 > $D/fuzz_frontend_8000.json 2>/dev/null; tail -c 220 $D/fuzz_frontend_8000.json; echo
( time timeout 600 python bench.py --detail-file $D/bench.json > $D/bench.stdout 2> $D/bench.err ) 2> $D/bench_wall.txt; tail -n 1 $D/bench.stdout | cut -c1-400; tail -3 $D/bench_wall.txt
