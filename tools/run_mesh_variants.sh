#!/bin/bash
# usage: tools/run_mesh_variants.sh name...
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for name in "$@"; do
  cp variants/librt_hip_$name.so raytracer_amd/librt_hip.so
  echo "== $name" >> gpurun_out/variants.log
  for a in "--workload bounding-box-1000x400" "--workload bounding-box-1000x400 --scale 3" "--workload chapter15-dragon-1000x1000" "--workload chapter15-dragon-1000x1000 --scale 3"; do
    timeout -k 10 200 python3 bench.py $a --no-cpu-baseline --no-host-frames 2>/dev/null | python3 tools/bench_line.py >> gpurun_out/variants.log || { echo "FAILED $name" >> gpurun_out/variants.log; exit 1; }
  done
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
