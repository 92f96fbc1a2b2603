#!/bin/bash
# usage: tools/build_at.sh <commit> <name> [flags]  -> build_variants/<name>/liborbx.so from the kernel sources of <commit> (A/B baseline of a round)
set -e
cd "$(dirname "$0")/.."
C=${1:?commit}; N=${2:?name}; F=$3
T=$(mktemp -d)
git archive "$C" orb_slam_amd/csrc include | tar -x -C "$T"
mkdir -p build_variants/$N
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form -Wall -Wno-unused-function -I$T/include -I$T/orb_slam_amd/csrc $F \
  -DORBX_SRC_HASH="\"at-$C\"" -shared $T/orb_slam_amd/csrc/*.hip -o build_variants/$N/liborbx.so
rm -rf "$T"; echo "$N done"
