cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c15; mkdir -p $O
(time timeout 1500 python -m pytest tests -m gpu -x -q) > $O/pytest_gpu.txt 2>&1; tail -5 $O/pytest_gpu.txt
timeout 600 python tools/fuzz_parity.py 4000 4001 > $O/fuzz_parity_4000.json 2>$O/fuzz_parity.err; tail -c 300 $O/fuzz_parity_4000.json
timeout 600 python tools/fuzz_batch.py 400 4002 > $O/fuzz_batch_400.json 2>$O/fuzz_batch.err; tail -c 300 $O/fuzz_batch_400.json
