#!/bin/bash
# A/B of library variants in ONE GPU call: serial per-kernel timings (every kernel alone on the chip) of build_variants/<v>/liborbx.so for
# a list of (variant, family) pairs.   usage: tools/exp_ab.sh <name> "<variant>:<family>[:config] ..."   (variant "tree" = orb_slam_amd/liborbx.so)
N=${1:?name}; shift
R=${GRAFT_REPO_ROOT:-/root/repo}; D=$R/gpurun_out/$N; mkdir -p $D; cd $R
S="--lanes 1 --steps 10 --warmup 2 --no-cpu-baseline --region-timing --min-seconds 0 --no-also --no-parity"
for item in $*; do
  v=${item%%:*}; rest=${item#*:}; f=${rest%%:*}; c=vga; [ "$rest" != "$f" ] && c=${rest#*:}
  lib=$R/build_variants/$v/liborbx.so; [ "$v" = tree ] && lib=$R/orb_slam_amd/liborbx.so
  ORBX_LIB=$lib timeout 300 python bench.py $S --family $f --config $c --detail-file $D/${v}_f${f}_$c.json > $D/${v}_f${f}_$c.line.json 2> $D/${v}_f${f}_$c.err
  python - <<PY
import json
try:
    d = json.load(open("$D/${v}_f${f}_$c.json"))
    print("%-10s fam $f %-7s step %.4f ms  %s" % ("$v", "$c", d["ms_per_step"], " ".join("%s %.4f" % (k[:6], x) for k, x in d["stage_ms_per_step"].items())))
except Exception as e:
    print("$v $f $c FAILED", e)
PY
done
