cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c1
(time timeout 1500 python -m pytest tests -m gpu -x -q) > gpurun_out/c1/pytest_gpu.txt 2>&1; tail -5 gpurun_out/c1/pytest_gpu.txt
ORBX_LIB=$GRAFT_REPO_ROOT/build_variants/pf/liborbx.so timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "stage_parity or end_to_end" > gpurun_out/c1/pytest_pf.txt 2>&1; tail -2 gpurun_out/c1/pytest_pf.txt
tools/exp_ab.sh c1 base:1 tree:1 onepass:1 e_win:1 e_patch:1 e_both:1 pf:1 pf_win:1 base:4 tree:4 base:0 tree:0 base:3 tree:3 base:1:hd1080 tree:1:hd1080 tree:1 base:1
timeout 900 python bench.py > gpurun_out/c1/bench.json 2> gpurun_out/c1/bench.err; tail -3 gpurun_out/c1/bench.err
python - <<PY
import json
d=json.load(open("gpurun_out/c1/bench.json"))
print(d["value"], d["ms_per_step"], d["config"]["parity_checked_frames"], d["config"]["parity_mismatches"], d["config"]["parity_note"][-40:])
for k,v in d["also"].items(): print(k, v["value"], v["ms_per_step"], v["config"].get("parity_checked_frames"), v["config"]["parity_mismatches"], v.get("stage_ms_per_step"))
print(d["cpu_baseline"]); print(d.get("cpu_baseline_reference_source"))
PY
