#!/bin/bash
# On the GPU box: for each variants/librt_hip_<name>.so, swap it in and run the given pytest selection.
# usage: tools/test_variants.sh "<pytest args>" name1 name2 ...
args="$1"; shift
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for name in "$@"; do
  cp variants/librt_hip_$name.so raytracer_amd/librt_hip.so
  echo "== $name" >> gpurun_out/test_variants.log
  timeout -k 10 200 python3 -m pytest $args -x -q --timeout 60 --timeout-method thread 2>&1 | tail -2 >> gpurun_out/test_variants.log
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
