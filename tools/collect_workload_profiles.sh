#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of bench.py for the other workloads, one run each,
# under gpurun_out/profiles_workloads/<workload>/.  Copy the kt_kernel_stats.csv files into profiles/.
export TMPDIR=/tmp
out=gpurun_out/profiles_workloads
for w in spheres-3000x2400-depth20 chapter15-dragon-1000x1000 bounding-box-1000x400 chapter16-1000x1000 area-light-800x320; do
  mkdir -p $out/$w
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -o kt -- python3 bench.py --workload $w --no-cpu-baseline --streams 1 > $out/$w/log.txt 2>&1 || exit 1
done
echo done
