#!/bin/bash
# usage: tools/build_variants.sh name1 "flags1" name2 "flags2" ...   -> build_variants/<name>/liborbx.so (A/B runs: LD_LIBRARY_PATH / ORBX_LIB)
cd "$(dirname "$0")/.."
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -Iinclude -Iorb_slam_amd/csrc"
while [ $# -ge 2 ]; do
  n=$1; f=$2; shift 2
  mkdir -p build_variants/$n
  ( /opt/rocm/bin/hipcc $HIPFLAGS $f -DORBX_SRC_HASH="\"variant-$n\"" -shared orb_slam_amd/csrc/*.hip -o build_variants/$n/liborbx.so 2>&1 | grep -E "error" ; echo "$n done" ) &
done
wait
