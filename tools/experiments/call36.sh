#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
Q="--no-cpu-baseline --no-also --min-seconds 2 --parity none"
for f in 0 20000 23000 27000 32000; do
  echo "floor $f: $(ORBX_FAST_LDS_FLOOR=$f timeout 200 python bench.py $Q 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
done
echo "floor 0 again: $(timeout 200 python bench.py $Q 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(d["value"], d["ms_per_step"])')"
