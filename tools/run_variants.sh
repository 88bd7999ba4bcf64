#!/bin/bash
# On the GPU box: for each variants/librt_hip_<name>.so, swap it in and run the bench once.
# usage: tools/run_variants.sh "<bench args>" name1 name2 ...
args="$1"; shift
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for name in "$@"; do
  cp variants/librt_hip_$name.so raytracer_amd/librt_hip.so
  echo "== $name" >> gpurun_out/variants.log
  timeout -k 10 200 python3 bench.py $args --no-cpu-baseline --no-host-frames 2>/dev/null | python3 tools/bench_line.py >> gpurun_out/variants.log || { echo "FAILED $name" >> gpurun_out/variants.log; break; }
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
