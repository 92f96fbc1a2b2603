#!/bin/bash
# Fast inner loop on the GPU box: extractor parity tests, then the per-kernel serial timings of the two stream shapes.
# usage: tools/quick_check.sh <name>   (-> gpurun_out/<name>/)
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$1; mkdir -p $D; cd $R
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_golden.py tests/test_gpu_select.py -x -q > $D/pytest.txt 2>&1; tail -3 $D/pytest.txt
S="--lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --region-timing --min-seconds 0 --no-also --no-parity"
timeout 300 python bench.py $S --detail-file $D/serial_vga.json > $D/serial_vga.line.json 2>$D/serial_vga.err
timeout 300 python bench.py $S --config hd1080 --detail-file $D/serial_hd.json > $D/serial_hd.line.json 2>$D/serial_hd.err
timeout 300 python bench.py --no-cpu-baseline --no-also --min-seconds 1.5 --detail-file $D/lanes_vga.json > $D/lanes_vga.line.json 2>$D/lanes_vga.err
python - <<PY
import json
for n in ("serial_vga", "serial_hd", "lanes_vga"):
    try:
        d = json.load(open("$D/%s.json" % n))
        print(n, d["value"], d["ms_per_step"], d["stage_ms_per_step"])
    except Exception as e:
        print(n, "failed", e)
PY
