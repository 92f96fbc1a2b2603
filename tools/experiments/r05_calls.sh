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
3)  # register-resident selection passes (ranges <= 64) + the fused stopper sweep against the build before them (head1) and round 4
mkdir -p gpurun_out/c3
(timeout 600 python -m pytest tests/test_gpu_select.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -5) > gpurun_out/c3/pytest.txt 2>&1; tail -2 gpurun_out/c3/pytest.txt
timeout 300 python tools/fuzz_batch.py 40 502 > gpurun_out/c3/fuzz_batch.json 2>/dev/null; tail -c 300 gpurun_out/c3/fuzz_batch.json; echo
timeout 300 python tools/fuzz_parity.py 400 503 > gpurun_out/c3/fuzz_parity.json 2>/dev/null; tail -c 300 gpurun_out/c3/fuzz_parity.json; echo
export ORBX_OVERLAP=0
tools/exp_ab.sh c3ab head1:1 tree:1 head1:0 tree:0 head1:3 tree:3 head1:4 tree:4 head1:1:hd1080 tree:1:hd1080 2>&1 | tail -12
;;
4)  # what crashed in call 3's test run
mkdir -p gpurun_out/c4
for t in tests/test_gpu_select.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_golden.py; do
  timeout 300 python -X faulthandler -m pytest $t -x -q -v > gpurun_out/c4/$(basename $t).txt 2>&1; echo "$t rc=$?"; grep -n "PASSED\|FAILED\|Fatal\|fault\|Memory access" gpurun_out/c4/$(basename $t).txt | tail -4
done
;;
5)  # which list aborts k_debug_nth (call 4: test_structured_lists)
mkdir -p gpurun_out/c5
timeout 200 python - > gpurun_out/c5/nth.txt 2>&1 <<'PY'
import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, oracle_lib as orc
from orb_slam_amd import capi
for n in [2, 3, 4, 5, 7, 63, 64, 65, 128, 1000, 2049, 5000]:
    base = np.arange(n, dtype=np.float32)
    pats = [base, base[::-1].copy(), np.full(n, 20, np.float32), np.minimum(base, base[::-1]),
            np.where(np.arange(n) % 2 == 0, 50, 10).astype(np.float32), np.concatenate([base[: n // 2], base[: n - n // 2]]),
            (np.arange(n) % 3).astype(np.float32)]
    for pi, r in enumerate(pats):
        for nth in sorted({1, n // 3, n // 2, n - 1}):
            if 0 < nth < n:
                print("n", n, "pat", pi, "nth", nth, flush=True)
                got = capi.nth_element_perm(r, nth); ref = orc.nth_element_perm(r, nth)
                if not np.array_equal(got, ref): print("  MISMATCH first at", int(np.argmax(got != ref)), flush=True)
print("done")
PY
tail -5 gpurun_out/c5/nth.txt
;;
6)  # the selection inlined again (call 3's build called wave_nth_element as a function: FLAT accesses, aperture violation in k_debug_nth)
mkdir -p gpurun_out/c6
(timeout 600 python -m pytest tests/test_gpu_select.py tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py tests/test_golden.py -x -q 2>&1 | tail -5) > gpurun_out/c6/pytest.txt 2>&1; tail -2 gpurun_out/c6/pytest.txt
timeout 300 python tools/fuzz_batch.py 60 602 > gpurun_out/c6/fuzz_batch.json 2>/dev/null; tail -c 300 gpurun_out/c6/fuzz_batch.json; echo
timeout 300 python tools/fuzz_parity.py 600 603 > gpurun_out/c6/fuzz_parity.json 2>/dev/null; tail -c 300 gpurun_out/c6/fuzz_parity.json; echo
export ORBX_OVERLAP=0
tools/exp_ab.sh c6ab head1:1 tree:1 head1:0 tree:0 head1:3 tree:3 head1:4 tree:4 head1:1:hd1080 tree:1:hd1080 2>&1 | tail -12
;;
7)  # k_blur_mfma with the waves of a workgroup staging together (2 / 4 adjacent strips), also on 1920-px rows (ORBX_BLUR_MFMA=2)
mkdir -p gpurun_out/c7
(timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_golden.py -x -q 2>&1 | tail -5) > gpurun_out/c7/pytest.txt 2>&1; tail -2 gpurun_out/c7/pytest.txt
for v in mb4 hdg2 hdg4; do ORBX_LIB=$R/build_variants/$v/liborbx.so timeout 300 python -m pytest tests/test_gpu_parity.py -x -q -k "blur_planes or large_batch" 2>&1 | tail -1; done
export ORBX_OVERLAP=0
tools/exp_ab.sh c7ab head2:1 tree:1 mb4:1 head2:1:hd1080 hdg2:1:hd1080 hdg4:1:hd1080 2>&1 | tail -12
;;
8)  # the masked dense match (t_valid) + where the default line stands
mkdir -p gpurun_out/c8
(timeout 600 python -m pytest tests/test_gpu_matcher.py -x -q 2>&1 | tail -5) > gpurun_out/c8/pytest.txt 2>&1; tail -2 gpurun_out/c8/pytest.txt
(time timeout 600 python bench.py --detail-file gpurun_out/c8/bench.json > gpurun_out/c8/bench.stdout 2> gpurun_out/c8/bench.err) 2>&1 | grep real
tail -1 gpurun_out/c8/bench.stdout | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms_per_step']); print({k:(v['value'], v['ms_per_step']) for k,v in d['also_summary'].items()})"
;;
9)  # a thirds class in k_cell_select_long (S-noise lists of 520-600 corners sat just above the quarter)
mkdir -p gpurun_out/c9
(timeout 600 python -m pytest tests/test_gpu_bench_shapes.py tests/test_gpu_parity.py -x -q 2>&1 | tail -5) > gpurun_out/c9/pytest.txt 2>&1; tail -2 gpurun_out/c9/pytest.txt
timeout 300 python tools/fuzz_batch.py 40 902 > gpurun_out/c9/fuzz_batch.json 2>/dev/null; tail -c 200 gpurun_out/c9/fuzz_batch.json; echo
ORBX_OVERLAP=0 tools/exp_ab.sh c9ab head3:0 tree:0 head3:3 tree:3 2>&1 | tail -4
for v in head3 tree; do lib=$R/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$R/orb_slam_amd/liborbx.so
ORBX_LIB=$lib timeout 200 python bench.py --family 0 --no-also --no-cpu-baseline --min-seconds 4 --parity sample | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v noise 4 lanes', d['value'], d['ms_per_step'])"; done
;;
10) # PCIe-inclusive line: NUMA-local pinned buffers, chunked uploads on two upload streams per lane
mkdir -p gpurun_out/c10
numactl -H 2>/dev/null | head -4; cat /sys/bus/pci/devices/*/numa_node 2>/dev/null | sort | uniq -c | head -4
for v in "" "--numa-bind" "--chunks 4 --streams-per-lane 2" "--numa-bind --chunks 4 --streams-per-lane 2" "--numa-bind --chunks 8 --streams-per-lane 2" "--numa-bind --outputs counts"; do
  timeout 120 python tools/bench_pcie.py --steps 30 $v 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%-50s %9.1f frames/s  %.3f ms  h2d %.1f GB/s d2h %.1f  bad %d  %s' % ('$v', d['value'], d['ms_per_step'], d['h2d_GBps'], d['d2h_GBps'], d['mismatching_frames'], d.get('numa_binding')))"
done | tee gpurun_out/c10/pcie.txt
;;
11) # k_fast_cells workgroups walking 2 / 4 / 16 bands (grid-stride); k_describe's table barrier behind the window DMA issue
mkdir -p gpurun_out/c11
for v in fp4 dlb; do ORBX_LIB=$R/build_variants/$v/liborbx.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py -x -q 2>&1 | tail -1; done
ORBX_OVERLAP=0 tools/exp_ab.sh c11ab tree:1 fp2:1 fp4:1 fp16:1 dlb:1 tree:1 tree:4 fp4:4 tree:1:hd1080 fp4:1:hd1080 dlb:1:hd1080 2>&1 | tail -12
for v in tree fp4 dlb; do lib=$R/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$R/orb_slam_amd/liborbx.so
ORBX_LIB=$lib timeout 200 python bench.py --no-also --no-cpu-baseline --min-seconds 4 --parity sample | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v blocks 4 lanes', d['value'], d['ms_per_step'])"; done
;;
12) # k_describe's prologue: pattern word requested first (dep), barrier behind the patch loads (dlb2 / dep2), keypoints + counts + status by scalar loads (dsc1, dsc = + dep)
mkdir -p gpurun_out/c12
for v in dep2 dsc; do ORBX_LIB=$R/build_variants/$v/liborbx.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_golden.py -x -q 2>&1 | tail -1; done
ORBX_OVERLAP=0 tools/exp_ab.sh c12ab tree:1 dep:1 dlb2:1 dep2:1 dsc1:1 dsc:1 tree:1 tree:1:hd1080 dep2:1:hd1080 dsc:1:hd1080 2>&1 | tail -12
for v in tree dep2 dsc; do lib=$R/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$R/orb_slam_amd/liborbx.so
ORBX_LIB=$lib timeout 200 python bench.py --no-also --no-cpu-baseline --min-seconds 4 --parity sample | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v blocks 4 lanes', d['value'], d['ms_per_step'])"; done
;;
13) # wave-life profiles of the round-5 k_describe / k_fast_cells (instrumented builds: tools/build_prof_variant.py describe | fast)
mkdir -p gpurun_out/c13
PROF_KERNEL=describe ORBX_LIB=$R/build_variants/profd/liborbx.so timeout 200 python tools/fast_prof.py 1 > gpurun_out/c13/describe_wave_phases.txt 2>&1; head -12 gpurun_out/c13/describe_wave_phases.txt
ORBX_LIB=$R/build_variants/prof/liborbx.so timeout 200 python tools/fast_prof.py 1 > gpurun_out/c13/fast_wave_phases.txt 2>&1; head -12 gpurun_out/c13/fast_wave_phases.txt
;;
14) # the bench tests after the change that keeps the JSON line last when RCCL writes to the C-level stdout
mkdir -p gpurun_out/c14
(time timeout 900 python -m pytest tests/test_gpu_bench.py -x -q) > gpurun_out/c14/pytest.txt 2>&1; tail -4 gpurun_out/c14/pytest.txt
timeout 300 python bench.py --gpus 1 --backend nccl --force-dist --steps 2 --warmup 1 --batch 128 --ring 256 --min-seconds 0.2 --no-also --no-cpu-baseline > gpurun_out/c14/rccl.stdout 2> gpurun_out/c14/rccl.err; tail -c 300 gpurun_out/c14/rccl.stdout; echo; grep -c . gpurun_out/c14/rccl.stdout; grep -v "^{\|^#detail" gpurun_out/c14/rccl.stdout | head -3
;;
15) # k_describe: the patch of levels >= 1 from dword-aligned addresses (36 bytes per row, three lanes x 12 bytes; masks and weights shifted instead of the data)
mkdir -p gpurun_out/c15
ORBX_LIB=$R/build_variants/dap/liborbx.so timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_bench_shapes.py tests/test_golden.py -x -q 2>&1 | tail -2
ORBX_LIB=$R/build_variants/dap/liborbx.so timeout 300 python tools/fuzz_batch.py 40 1502 2>/dev/null | tail -c 250; echo
ORBX_OVERLAP=0 tools/exp_ab.sh c15ab tree:1 dap:1 tree:1 dap:1 tree:1:hd1080 dap:1:hd1080 tree:4 dap:4 2>&1 | tail -8
for v in tree dap; do lib=$R/build_variants/$v/liborbx.so; [ $v = tree ] && lib=$R/orb_slam_amd/liborbx.so
ORBX_LIB=$lib timeout 200 python bench.py --no-also --no-cpu-baseline --min-seconds 4 --parity sample | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v blocks 4 lanes', d['value'], d['ms_per_step'])"; done
;;
16) # the final state: smoke(), the whole GPU suite, the default line (final bench.py: clock-priced rooflines, the line printed last)
mkdir -p gpurun_out/c16
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
(time python -m pytest tests -m gpu -q) > gpurun_out/c16/pytest_gpu.txt 2>&1; grep "passed\|failed" gpurun_out/c16/pytest_gpu.txt
(time timeout 600 python bench.py --detail-file gpurun_out/c16/bench.json > gpurun_out/c16/bench.stdout 2> gpurun_out/c16/bench.err) 2>&1 | grep real
tail -1 gpurun_out/c16/bench.stdout > gpurun_out/c16/bench.line.json; wc -c gpurun_out/c16/bench.line.json
python -c "import json; d=json.load(open('gpurun_out/c16/bench.line.json')); print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['valu_issue'], d['roofline_valu']); print({k:(v['value'], v['parity_mismatches']) for k,v in d['also_summary'].items()})"
;;
17) # a second, larger fuzz round on the final build (other seeds)
mkdir -p gpurun_out/c17
timeout 900 python tools/fuzz_parity.py 8000 6201 > gpurun_out/c17/fuzz_parity_8000.json 2>/dev/null; tail -c 250 gpurun_out/c17/fuzz_parity_8000.json; echo
timeout 900 python tools/fuzz_batch.py 700 6202 > gpurun_out/c17/fuzz_batch_700.json 2>/dev/null; tail -c 250 gpurun_out/c17/fuzz_batch_700.json; echo
timeout 900 python tools/fuzz_frontend.py 6000 6203 > gpurun_out/c17/fuzz_frontend_6000.json 2>/dev/null; tail -c 250 gpurun_out/c17/fuzz_frontend_6000.json; echo
timeout 900 python tools/fuzz_match.py 3000 6204 > gpurun_out/c17/fuzz_match_3000.json 2>/dev/null; tail -c 250 gpurun_out/c17/fuzz_match_3000.json; echo
;;
*) echo "usage: $0 <call number>"; exit 2 ;;
esac
