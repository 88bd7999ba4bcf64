#!/bin/bash
# On the GPU box: rocprofv3 --kernel-trace --stats of bench.py for the other workloads, one run each,
# under gpurun_out/profiles_workloads/<workload>/.  Copy the kt_kernel_stats.csv files into profiles/.
export TMPDIR=/tmp
out=gpurun_out/profiles_workloads
for w in spheres-3000x2400-depth20 chapter15-dragon-1000x1000 bounding-box-1000x400 chapter16-1000x1000 area-light-800x320; do
  mkdir -p $out/$w
  timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/$w -o kt -- python3 bench.py --workload $w --no-cpu-baseline --no-host-frames --streams 1 > $out/$w/log.txt 2>&1 || exit 1
done

# the two larger mesh frames
mkdir -p $out/bounding-box-3000x1200 $out/chapter15-dragon-3000x3000
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/bounding-box-3000x1200 -o kt -- python3 bench.py --workload bounding-box-1000x400 --scale 3 --no-cpu-baseline --no-host-frames --streams 1 > $out/bounding-box-3000x1200/log.txt 2>&1 || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/chapter15-dragon-3000x3000 -o kt -- python3 bench.py --workload chapter15-dragon-1000x1000 --scale 3 --no-cpu-baseline --no-host-frames --streams 1 > $out/chapter15-dragon-3000x3000/log.txt 2>&1 || exit 1
echo done
