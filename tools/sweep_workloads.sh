#!/bin/bash
# On the GPU box: one bench.py line per workload (with the CPU oracle beside it) into gpurun_out/workloads.jsonl.
out=gpurun_out/workloads.jsonl
: > $out
for w in chapter6-1000x1000 spheres-3000x2400-depth20 chapter15-dragon-1000x1000 bounding-box-1000x400 chapter16-1000x1000 area-light-800x320; do
  timeout -k 10 400 python3 bench.py --workload $w 2>/dev/null >> $out || exit 1
done
timeout -k 10 400 python3 bench.py --workload bounding-box-1000x400 --scale 3 --cpu-rows 96 2>/dev/null >> $out || exit 1
timeout -k 10 400 python3 bench.py --workload chapter15-dragon-1000x1000 --scale 3 --cpu-rows 256 2>/dev/null >> $out || exit 1
