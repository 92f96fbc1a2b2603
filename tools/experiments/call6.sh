cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c6
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k "blur_planes" > gpurun_out/c6/pytest_blur.txt 2>&1; tail -12 gpurun_out/c6/pytest_blur.txt
tools/exp_ab.sh c6 tree:1 noblurm:1 tree:1:hd1080 noblurm:1:hd1080 tree:1 noblurm:1
