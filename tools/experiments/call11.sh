cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/c11; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_pipeline.py -x -q -k upload > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
timeout 300 python tools/bench_pcie.py --ingest kernel > $O/pcie_kernel.json 2>$O/pcie_kernel.err; cat $O/pcie_kernel.json; tail -2 $O/pcie_kernel.err
timeout 300 python tools/bench_pcie.py --ingest copy > $O/pcie_copy.json 2>$O/pcie_copy.err; cat $O/pcie_copy.json
HSA_ENABLE_SDMA=0 timeout 300 python tools/bench_pcie.py --ingest copy > $O/pcie_copy_nosdma.json 2>$O/pcie_copy_nosdma.err; cat $O/pcie_copy_nosdma.json
timeout 300 python tools/bench_pcie.py --ingest kernel --outputs counts > $O/pcie_kernel_counts.json 2>/dev/null; cat $O/pcie_kernel_counts.json
