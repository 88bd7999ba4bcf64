#!/bin/bash
# On the GPU box: for each variants/librt_hip_<name>.so, swap it in and run the bench once.
# usage: tools/run_variants.sh "<bench args>" name1 name2 ...
args="$1"; shift
cp raytracer_amd/librt_hip.so /tmp/librt_hip_orig.so
for name in "$@"; do
  cp variants/librt_hip_$name.so raytracer_amd/librt_hip.so
  echo "== $name" >> gpurun_out/variants.log
  timeout -k 10 200 python3 bench.py $args --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
for line in sys.stdin:
    if line.startswith('{'):
        d = json.loads(line); r = d['roofline']; o = r['other_kernels']
        print('frame %.1f ms  trace %.1f  shadow %.1f  shade %.1f  Mrays/s %.0f' % (d['ms_per_step'], r['kernel_ms_per_frame'], o['k_shadow']['ms_per_frame'], o['k_shade']['ms_per_frame'], d['value']))
" >> gpurun_out/variants.log || { echo "FAILED $name" >> gpurun_out/variants.log; break; }
done
cp /tmp/librt_hip_orig.so raytracer_amd/librt_hip.so
