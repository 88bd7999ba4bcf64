#!/bin/bash
# On the GPU box: run one single-stream frame with each -DRT_PHASE_TIMERS variant; the per-phase clock sums go to stderr.
# usage: tools/run_phase_variants.sh "<bench args>" name1 name2 ...
args="$1"; shift
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for name in "$@"; do
  cp variants/librt_hip_$name.so raytracer_amd/librt_hip.so
  echo "== $name" >> gpurun_out/phases.log
  timeout -k 10 200 python3 bench.py $args --no-cpu-baseline --streams 1 2>> gpurun_out/phases.log | python3 tools/bench_line.py >> gpurun_out/phases.log || { echo "FAILED $name" >> gpurun_out/phases.log; break; }
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
