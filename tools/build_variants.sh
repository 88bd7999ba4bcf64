#!/bin/bash
# Builds experiment variants of librt_hip.so into variants/ (scratch; for timing experiments only).
# usage: tools/build_variants.sh name1="-DFLAG ..." name2="..."
cd "$(dirname "$0")/../raytracer_amd/csrc" || exit 1
mkdir -p ../../variants
for spec in "$@"; do
  name="${spec%%=*}"; flags="${spec#*=}"
  ( hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -Xclang -target-feature -Xclang -packed-fp32-ops $flags -shared rt_device.hip -o ../../variants/librt_hip_$name.so && echo built $name ) &
done
wait
