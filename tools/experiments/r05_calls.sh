#!/bin/bash
# The GPU calls of round 5, one case per call (same conventions as r04_calls.sh).  usage: tools/experiments/r05_calls.sh <n>
R=${GRAFT_REPO_ROOT:-$(git -C "$(dirname "$0")" rev-parse --show-toplevel)}; cd "$R"
case "$1" in
1)  # the new bench line, the band hint + workgroup-form long selection against round 4
(timeout 600 python -m pytest tests/test_gpu_select.py tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -x -q 2>&1 | tail -15) > gpurun_out/c1/pytest.txt 2>&1
tail -4 gpurun_out/c1/pytest.txt
tools/exp_ab.sh c1ab r04:1 tree:1 r04:0 tree:0 r04:3 tree:3 r04:4 tree:4 tree:1:hd1080 r04:1:hd1080 2>&1 | tail -12
(time timeout 600 python bench.py --detail-file gpurun_out/c1/bench.json > gpurun_out/c1/bench.stdout 2> gpurun_out/c1/bench.err) 2>&1 | grep real
tail -c 3700 gpurun_out/c1/bench.stdout | head -c 1200; echo; wc -c gpurun_out/c1/bench.stdout; tail -1 gpurun_out/c1/bench.stdout | wc -c
;;
2)  # class-partitioned k_cell_select_long (chunks of 8 / 16 / 32 cells) against round 4, every kernel alone on the chip
(timeout 600 python -m pytest tests/test_gpu_select.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5) > gpurun_out/c2/pytest.txt 2>&1; tail -2 gpurun_out/c2/pytest.txt
timeout 300 python tools/fuzz_batch.py 40 501 > gpurun_out/c2/fuzz_batch.json 2>/dev/null; tail -c 300 gpurun_out/c2/fuzz_batch.json; echo
export ORBX_OVERLAP=0
tools/exp_ab.sh c2ab r04:0 tree:0 chunk8:0 chunk32:0 r04:3 tree:3 chunk8:3 chunk32:3 r04:1 tree:1 2>&1 | tail -12
;;
*) echo "usage: $0 <call number>"; exit 2 ;;
esac
