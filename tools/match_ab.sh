#!/bin/bash
# A/B of library variants on the dense top-2 kernels in ONE GPU call, variants interleaved twice (order effects: clocks, thermals).
# usage: tools/match_ab.sh <name> "<variant> ..." [paths]      (variant "tree" = orb_slam_amd/liborbx.so, else build_variants/<v>/liborbx.so)
N=${1:?name}; V=${2:?variants}; P=${3:-2}
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$N; mkdir -p $D; cd $R
for round in 1 2; do
  for v in $V; do
    lib=$R/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$R/orb_slam_amd/liborbx.so
    ORBX_LIB=$lib timeout 200 python tools/match_paths_bench.py --paths $P > $D/${v}_$round.json 2> $D/${v}_$round.err
    python - <<PY
import json
try:
    d = json.load(open("$D/${v}_$round.json"))
    print("%-8s r$round " % "$v" + "  ".join("%s med %.4f sus %.4f %s" % (k, x["median_ms"], x["sustained_ms"], "" if x["equal_to_first"] else "MISMATCH") for k, x in d.items()))
except Exception as e:
    print("$v FAILED", e)
PY
  done
done
